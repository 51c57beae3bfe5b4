"""Module names of the reference's predict_model/ package, so `from predict_model.v3plus import
Xception_DeepLabV3_Plus` (predict.py:5-9) keeps working with `building_detection_amd.predict_model`."""

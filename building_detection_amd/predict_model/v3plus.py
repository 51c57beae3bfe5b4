from ..zoo.deeplab import Xception_DeepLabV3_Plus  # predict.py:7

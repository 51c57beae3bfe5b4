"""building_detection_amd — MI355X (gfx950) native engine for the segmentation hot path of
A511-1103/building-detection: DeepLabv3+/ASPP, DeepLab-BAM, SCSE-UNet, Res34-UNet and HRNet forward+backward
behind the reference's own tf.keras Model-build / predict.py / model_fuse.py surface.

Importing the package needs no GPU (graph construction, shape inference, LR schedules and weight I/O are host
logic); anything that computes goes through libsegengine.so (hand-written HIP, C ABI in include/segengine.h)
and raises if the library or a gfx950 device is missing — there is no CPU fallback.
"""
__version__ = "0.1.0"

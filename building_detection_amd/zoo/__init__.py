"""The five segmentation graphs of the reference (predict_model/*.py), built on the engine's layers."""
from .deeplab import Xception_DeepLabV3_Plus, Xception_DeepLabV3_Plus_bam
from .unets import UNet, ResNetFamily, HRNet

BUILDERS = {
    "v3plus": Xception_DeepLabV3_Plus,
    "bam": Xception_DeepLabV3_Plus_bam,
    "scse": lambda shape=(512, 512, 3), num_classes=2: UNet(num_classes, shape),
    "res34": lambda shape=(512, 512, 3), num_classes=2: ResNetFamily(shape).run_model("res34"),
    "hrnet": HRNet,
}

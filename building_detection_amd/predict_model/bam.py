from ..zoo.deeplab import Xception_DeepLabV3_Plus_bam  # predict.py:9

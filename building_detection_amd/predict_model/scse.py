from ..zoo.unets import UNet  # predict.py:8

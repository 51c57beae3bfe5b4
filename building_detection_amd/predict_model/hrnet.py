from ..zoo.unets import HRNet  # predict.py:6

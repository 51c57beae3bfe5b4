from ..zoo.unets import ResNetFamily  # predict.py:5

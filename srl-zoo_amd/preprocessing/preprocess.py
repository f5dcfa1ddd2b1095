"""Geometry of the observations (reference: preprocessing/preprocess.py:7-17).

Every frame is resized to 224 x 224; the channel count is a MODULE GLOBAL because train.py overwrites it for --multi-view
(two stacked cameras -> 6, with triplets 9: reference train.py:116-123) before any network is built, and the networks read
it at construction time through getNChannels().
"""
IMAGE_HEIGHT = IMAGE_WIDTH = 224
N_CHANNELS = 3  # overwritten by train.py / tests: always read it through getNChannels()


def getNChannels():
    """Current number of input channels."""
    return N_CHANNELS


def getInputDim():
    """Flattened size of one observation with the current channel count."""
    return getNChannels() * IMAGE_HEIGHT * IMAGE_WIDTH


INPUT_DIM = getInputDim()  # value at import time (3 channels), kept for code that imports the constant

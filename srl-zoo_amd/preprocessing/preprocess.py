"""Image constants shared by the loader and the networks (reference: preprocessing/preprocess.py:7-17).

N_CHANNELS is a module global that train.py overwrites for --multi-view (reference train.py:116-123); the networks
read it at construction time through getNChannels().
"""
IMAGE_WIDTH = 224
IMAGE_HEIGHT = 224
N_CHANNELS = 3
INPUT_DIM = IMAGE_WIDTH * IMAGE_HEIGHT * N_CHANNELS


def getNChannels():
    return N_CHANNELS


def getInputDim():
    return IMAGE_WIDTH * IMAGE_HEIGHT * N_CHANNELS

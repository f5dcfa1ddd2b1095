from .data_loader import preprocessImage  # noqa: F401
from .preprocess import getNChannels, getInputDim, N_CHANNELS  # noqa: F401

"""Pixel normalisation (reference: preprocessing/utils.py:6-66)."""
import numpy as np

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def preprocessInput(x, mode="image_net"):
    """In-place normalisation of an RGB float image in [0,255] -> ImageNet-normalised ("image_net") or [-1,1] ("tf")."""
    assert x.shape[-1] == 3, "Color channel must be at the end of the tensor {}".format(x.shape)
    x /= 255.
    if mode == "tf":
        x -= 0.5
        x *= 2.
    elif mode == "image_net":
        for c in range(3):
            x[..., c] -= _MEAN[c]
        for c in range(3):
            x[..., c] /= _STD[c]
    else:
        raise ValueError("Unknown mode for preprocessing")
    return x


def deNormalize(x, mode="image_net"):
    """Inverse of preprocessInput; a single (3, W, H) tensor is first transposed back to (H, W, 3)."""
    if x.shape[0] == 3 and len(x.shape) == 3:
        x = np.transpose(x, (2, 1, 0))
    assert x.shape[-1] == 3, "Color channel must be at the end of the tensor {}".format(x.shape)
    if mode == "tf":
        x /= 2.
        x += 0.5
    elif mode == "image_net":
        for c in range(3):
            x[..., c] *= _STD[c]
        for c in range(3):
            x[..., c] += _MEAN[c]
    else:
        raise ValueError("Unknown mode for deNormalize")
    return np.clip(x, 0, 1)

import numpy as np


def pad_center(data, size=None, axis=-1, **kw):
    n = data.shape[axis]
    lp = (size - n) // 2
    pads = [(0, 0)] * data.ndim
    pads[axis] = (lp, size - n - lp)
    return np.pad(data, pads)


def tiny(x):
    return np.finfo(np.float32).tiny


def normalize(x, **kw):
    return x

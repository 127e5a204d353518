"""Camera-convention helper of the path (reference utils/accio2ngp.py:133-139)."""
import numpy as np


def converter(T_accio_list):
    """OpenCV -> NGP/OpenGL camera convention: negate the y and z axis columns of the
    rotation block of every 4x4 in the batch; returns a copy, dtype preserved."""
    out = np.array(T_accio_list, copy=True)
    out[..., :3, 1:3] *= -1
    return out

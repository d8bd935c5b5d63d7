"""JPEG -> DCT coefficients without the inverse transform: the `jpeg2dct` package surface the reference's
generators import (`from jpeg2dct.numpy import load, loads`), served by the in-tree C++ reader (csrc/dj_jpeg.cpp,
C ABI include/dj_jpeg.h)."""
from . import numpy  # noqa: F401

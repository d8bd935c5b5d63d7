"""`jpeg2dct.numpy` look-alike: load / loads -> (dct_y, dct_cb, dct_cr), each (blocks_h, blocks_w, 64) int16 in natural
coefficient order, de-quantised when normalized=True (jpeg2dct's default) -- what
localisation_part/data_generator/object_detection_2d_data_generator_dct_j2d.py:1180 and
classification_part/vgg_jpeg_keras/generators/generators.py:120-130 consume.  `decode_batch` is the batched,
multi-threaded form that fills the float32 batch tensors directly."""
from __future__ import absolute_import

import ctypes
import os

import numpy as _np

_LIB = None


class JpegInfo(ctypes.Structure):   # dj_jpeg_info
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("n_components", ctypes.c_int),
                ("h_samp", ctypes.c_int * 4), ("v_samp", ctypes.c_int * 4), ("blocks_w", ctypes.c_int * 4),
                ("blocks_h", ctypes.c_int * 4), ("quant", (ctypes.c_int * 64) * 4), ("sof", ctypes.c_int)]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "libdj_jpeg.so")
        if not os.path.exists(path):
            raise ImportError("libdj_jpeg.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        lib.dj_jpeg_last_error.restype = ctypes.c_char_p
        lib.dj_jpeg_read_info.restype = ctypes.c_int
        lib.dj_jpeg_read_info.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.POINTER(JpegInfo)]
        lib.dj_jpeg_read_coefficients.restype = ctypes.c_int
        lib.dj_jpeg_read_coefficients.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int,
                                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long),
                                                  ctypes.POINTER(JpegInfo)]
        lib.dj_jpeg_decode_batch_f32.restype = ctypes.c_int
        lib.dj_jpeg_decode_batch_f32.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_long),
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p] + [ctypes.c_int] * 5
        _LIB = lib
    return _LIB


def _check(rc):
    if rc != 0:
        raise ValueError("JPEG coefficient reader: " + _lib().dj_jpeg_last_error().decode())


def info(buf):
    """Frame geometry, sampling factors and quantisation tables (natural order) without decoding."""
    buf = bytes(buf)
    inf = JpegInfo()
    _check(_lib().dj_jpeg_read_info(buf, len(buf), ctypes.byref(inf)))
    return inf


def loads(buf, normalized=True, channels=3):
    buf = bytes(buf)
    inf = info(buf)
    if inf.sof not in (0, 1):
        raise ValueError("JPEG coefficient reader: unsupported JPEG process SOF%d (only baseline / extended "
                         "sequential Huffman)" % inf.sof)
    n = inf.n_components
    planes = [_np.empty((inf.blocks_h[c], inf.blocks_w[c], 64), dtype=_np.int16) for c in range(n)]
    ptrs = (ctypes.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - n))
    caps = (ctypes.c_long * 4)(*[p.size for p in planes] + [0] * (4 - n))
    _check(_lib().dj_jpeg_read_coefficients(buf, len(buf), int(bool(normalized)), ptrs, caps, None))
    if channels == 1 or n == 1:
        empty = _np.zeros((0, 0, 64), dtype=_np.int16)
        return (planes[0], empty, empty) if channels == 3 else (planes[0],)
    return tuple(planes[:3])


def load(filename, normalized=True, channels=3):
    with open(filename, "rb") as f:
        return loads(f.read(), normalized=normalized, channels=channels)


def decode_batch(buffers, y_blocks, c_blocks, normalized=True, n_threads=None, out=None):
    """n JPEG byte strings -> float32 (n, yh, yw, 64), (n, ch, cw, 64), (n, ch, cw, 64), decoded by `n_threads` host
    threads (the GIL is released for the whole call).  `out` may hold three preallocated (e.g. pinned) arrays."""
    bufs = [bytes(b) for b in buffers]
    n = len(bufs)
    (yh, yw), (ch, cw) = y_blocks, c_blocks
    if out is None:
        out = (_np.empty((n, yh, yw, 64), _np.float32), _np.empty((n, ch, cw, 64), _np.float32),
               _np.empty((n, ch, cw, 64), _np.float32))
    y, cb, cr = out
    for a, shp in ((y, (n, yh, yw, 64)), (cb, (n, ch, cw, 64)), (cr, (n, ch, cw, 64))):
        assert a.dtype == _np.float32 and a.flags["C_CONTIGUOUS"] and tuple(a.shape) == shp
    arr = (ctypes.c_char_p * n)(*bufs)
    sizes = (ctypes.c_long * n)(*[len(b) for b in bufs])
    if n_threads is None:
        try:
            n_threads = len(os.sched_getaffinity(0))
        except AttributeError:
            n_threads = os.cpu_count() or 1
    _check(_lib().dj_jpeg_decode_batch_f32(arr, sizes, n, int(bool(normalized)), y.ctypes.data, cb.ctypes.data,
                                           cr.ctypes.data, yh, yw, ch, cw, int(n_threads)))
    return y, cb, cr

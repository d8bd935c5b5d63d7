"""Subset of `keras.backend` the reference's model code touches (names, uid counters, layout)."""
import collections

import torch

_UIDS = collections.defaultdict(int)
_SEED = [0]
_GEN = [None]


def backend():
    # the reference's AnchorBoxes layer insists on 'tensorflow'
    # (localisation_part/keras_layers/keras_layer_AnchorBoxes.py:98-99); the layout contract is the same
    return "tensorflow"


def image_dim_ordering():
    return "tf"


def image_data_format():
    return "channels_last"


_FLOATX = {0: "float32", 1: "float16", 2: "bfloat16", 3: "float32x3", 4: "float32x6", 5: "float32_mfma"}


def floatx():
    """'float32' (default), or the reduced-precision MFMA mode set by `set_floatx`."""
    from .. import _lib
    return _FLOATX[_lib.load().dj_get_compute_mode()]


def set_floatx(value):
    """`K.set_floatx('float16')` selects the mixed-precision convolution arithmetic of BASELINE config 5: forward GEMMs
    on fp16 MFMA, gradient GEMMs on bf16 MFMA, fp32 accumulation; weights, activations, gradients and optimizer state
    stay fp32 tensors (unlike Keras, which would also store float16 variables).  'bfloat16': bf16 in every GEMM.
    'float32x3' (no Keras counterpart): fp32 tensors and fp32-grade results (~1e-5 relative per product), every product
    as three bf16 MFMAs on operands split into a high and a low bf16 half when they go to LDS.
    'float32x6': three bf16 pieces per operand (all 24 significant bits) and six MFMAs per product -- what is dropped is
    2^-24 of a product, the size of one fp32 rounding: fp32 results, on the bf16 matrix pipe.
    'float32' (default): fp32 tensors and fp32 results -- per geometry the fp32 MFMA kernel or the float32x6 kernel,
    whichever the tuning table measured faster (equally far from the fp64 oracle); what every 1e-3 parity claim refers to.
    'float32_mfma': fp32 MFMA instructions only (the default before the split kernels existed)."""
    from .. import _lib
    modes = {v: k for k, v in _FLOATX.items()}
    if value not in modes:
        raise ValueError("Unknown floatx type: " + str(value))
    lib = _lib.load()
    lib.dj_set_compute_mode(modes[value])        # process default: what models lowered from now on compute in
    lib.dj_set_thread_compute_mode(-1)           # this thread follows it again (a plan run may have pinned its own mode)


def get_uid(prefix=""):
    _UIDS[prefix] += 1
    return _UIDS[prefix]


def clear_session():
    _UIDS.clear()


def set_random_seed(seed):
    """Seed of the weight initialisers (Keras takes it from numpy / TF global seeds)."""
    _SEED[0] = int(seed)
    _GEN[0] = None


def generator():
    if _GEN[0] is None:
        _GEN[0] = torch.Generator().manual_seed(_SEED[0])
    return _GEN[0]


def int_shape(x):
    return x.shape


def shape(x):
    return x.shape

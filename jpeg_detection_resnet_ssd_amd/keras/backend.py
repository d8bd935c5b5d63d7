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


def floatx():
    return "float32"


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

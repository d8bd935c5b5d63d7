"""Keras 2.2.4 initialisers used by the reference's builders: he_normal (every Conv2D),
glorot_uniform (Conv2DTranspose / Dense defaults), zeros / ones / constant."""
import math

import torch

from . import backend as K


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    receptive = 1
    for s in shape[:-2]:
        receptive *= s
    return shape[-2] * receptive, shape[-1] * receptive


def he_normal(shape):
    fan_in, _ = _fans(shape)
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=K.generator())
    return t


def glorot_uniform(shape):
    fan_in, fan_out = _fans(shape)
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=K.generator()) * 2 - 1) * limit


def zeros(shape):
    return torch.zeros(shape)


def ones(shape):
    return torch.ones(shape)


def constant(value):
    def init(shape):
        return torch.full(shape, float(value))
    return init


def get(identifier):
    if callable(identifier):
        return identifier
    table = {"he_normal": he_normal, "glorot_uniform": glorot_uniform, "zeros": zeros, "ones": ones}
    if identifier not in table:
        raise ValueError("unknown initializer %r" % (identifier,))
    return table[identifier]

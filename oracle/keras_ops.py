"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain PyTorch functional ops, of the Keras 2.2.4 / TensorFlow 1.8
layer arithmetic the reference's hot path runs.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this package.

PARITY UNPINNED for everything in this file: the arithmetic lives in third-party
Keras 2.2.4 + tensorflow-gpu 1.8.0 (pins: localisation_part/Pipfile:7-8,
classification_part/requirements.txt:28,44,99), which are neither vendored in the
reference nor installable here (no network), and the reference's only test module
(classification_part/vgg_jpeg_keras/tests/generators/tests_generators.py) holds no
vector for these ops.  The semantics below restate the published behaviour of those
libraries, anchored on the reference's call sites (cited per function, paths relative to
the reference root, L/ = localisation_part/, C/ = classification_part/).

All tensors are NHWC (`bn_axis = 3`, L/models/keras_ssd300_dct_j2d_resnet.py:61); conv
kernels are HWIO, Conv2DTranspose kernels (kh, kw, out, in) -- the Keras layouts.
Functions are dtype-agnostic (float32 or float64) and differentiable by torch.autograd.
"""
import math

import torch
import torch.nn.functional as F

BN_EPSILON = 1e-3   # keras.layers.BatchNormalization default (Keras 2.2.4)
BN_MOMENTUM = 0.99  # keras.layers.BatchNormalization default (Keras 2.2.4)


def same_padding(in_size, kernel, stride, dilation=1):
    """TensorFlow 'SAME' padding: (before, after, out_size).  Even kernels pad (0, 1)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + (kernel - 1) * dilation + 1 - in_size, 0)
    before = total // 2
    return before, total - before, out


def valid_out_size(in_size, kernel, stride, dilation=1):
    return (in_size - (kernel - 1) * dilation - 1) // stride + 1


def conv2d(x, kernel, bias=None, strides=(1, 1), padding="valid", dilation_rate=(1, 1)):
    """keras.layers.Conv2D.call: cross-correlation + bias.

    Call sites: L/models/keras_ssd300_dct_j2d_resnet.py:77-96,128-160 (ResNet blocks,
    `valid` 1x1 and `same` kxk), :483-545 (SSD extra layers, dilation 6, stride 2 valid),
    :562-675 (predictors, 3x3 same)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    xn = x.permute(0, 3, 1, 2)
    if padding == "same":
        pt, pb, _ = same_padding(x.shape[1], kh, strides[0], dilation_rate[0])
        pl, pr, _ = same_padding(x.shape[2], kw, strides[1], dilation_rate[1])
        xn = F.pad(xn, (pl, pr, pt, pb))
    elif padding != "valid":
        raise ValueError("padding must be 'valid' or 'same'")
    w = kernel.permute(3, 2, 0, 1)
    y = F.conv2d(xn, w, bias, stride=strides, padding=0, dilation=dilation_rate)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose(x, kernel, bias=None, strides=(2, 2)):
    """keras.layers.Conv2DTranspose(filters, k, strides=s), padding 'valid'
    (L/models/keras_ssd300_dct_j2d_resnet.py:1709-1711; C/vgg_jpeg_keras/networks/resnet_dct.py:614-616).
    Kernel layout (kh, kw, out, in); out[n, s*i+a, s*j+b, co] += x[n,i,j,ci] * K[a,b,co,ci]."""
    xn = x.permute(0, 3, 1, 2)
    w = kernel.permute(3, 2, 0, 1)  # torch conv_transpose2d weight: (in, out, kh, kw)
    y = F.conv_transpose2d(xn, w, bias, stride=strides)
    return y.permute(0, 2, 3, 1)


def batch_norm_train(x, gamma, beta, epsilon=BN_EPSILON):
    """keras.layers.BatchNormalization(axis=3) in training mode: batch statistics over
    (N,H,W), biased variance for the normalisation.  Returns (y, mean, biased_var).
    Call sites: L/models/keras_ssd300_dct_j2d_resnet.py:80,90,96,135,145,151,160,446,458,1716."""
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) * torch.rsqrt(var + epsilon) * gamma + beta
    return y, mean, var


def batch_norm_moving_update(moving_mean, moving_var, mean, biased_var, count, momentum=BN_MOMENTUM):
    """Moving-average update Keras 2.2.4 performs after a training step with the TF fused
    batch norm: moving = moving*momentum + batch*(1-momentum), the batch variance being
    Bessel-corrected (count/(count-1)) as tf.nn.fused_batch_norm returns it."""
    unbiased = biased_var * (count / max(count - 1, 1))
    return (moving_mean * momentum + mean * (1 - momentum),
            moving_var * momentum + unbiased * (1 - momentum))


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var, epsilon=BN_EPSILON):
    return (x - moving_mean) * torch.rsqrt(moving_var + epsilon) * gamma + beta


def relu(x):
    return torch.clamp_min(x, 0)


def max_pool_3x3_s1_same(x):
    """MaxPooling2D((3,3), strides=(1,1), padding='same') -- `pool5_ssd`
    (L/models/keras_ssd300_dct_j2d_resnet.py:481,1110).  TF pads with -inf (padding never wins)."""
    xn = x.permute(0, 3, 1, 2)
    y = F.max_pool2d(xn, kernel_size=3, stride=1, padding=1)
    return y.permute(0, 2, 3, 1)


def max_pool(x, pool, strides, padding="valid"):
    """General MaxPooling2D (ResNet50RGB stem: 3x3 stride 2 after ZeroPadding2D(1),
    C/vgg_jpeg_keras/networks/resnet_dct.py:165-314)."""
    xn = x.permute(0, 3, 1, 2)
    if padding == "same":
        pt, pb, _ = same_padding(x.shape[1], pool[0], strides[0])
        pl, pr, _ = same_padding(x.shape[2], pool[1], strides[1])
        xn = F.pad(xn, (pl, pr, pt, pb), value=float("-inf"))
    y = F.max_pool2d(xn, kernel_size=pool, stride=strides)
    return y.permute(0, 2, 3, 1)


def zero_padding(x, pad):
    """ZeroPadding2D(((t,b),(l,r))) (L/models/keras_ssd300_dct_j2d_resnet.py:514,1143,1166)."""
    (t, b), (l, r) = pad
    return F.pad(x, (0, 0, l, r, t, b))


def upsampling_nearest_2x(x):
    """UpSampling2D() (L/models/keras_ssd300_dct_j2d_resnet.py:1669): each pixel repeated 2x2."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def l2_normalization(x, gamma):
    """L/keras_layers/keras_layer_L2Normalization.py:61-63: K.l2_normalize(x, axis=3) * gamma;
    TF l2_normalize = x * rsqrt(max(sum(x^2), 1e-12))."""
    ss = (x * x).sum(dim=-1, keepdim=True)
    return x * torch.rsqrt(torch.clamp_min(ss, 1e-12)) * gamma


def softmax(x):
    """Activation('softmax') on the last axis (L/models/keras_ssd300_dct_j2d_resnet.py:873)."""
    return torch.softmax(x, dim=-1)


def global_average_pooling(x):
    return x.mean(dim=(1, 2))


def dense(x, kernel, bias):
    return x @ kernel + bias


def categorical_crossentropy(y_true, probs):
    """keras.losses.categorical_crossentropy on softmax outputs (TF backend, Keras 2.2.4):
    renormalise, clip to [1e-7, 1-1e-7], -sum(y*log p); Keras then means over the batch.
    Call site: C/config/resnet/config_file.py:64."""
    p = probs / probs.sum(dim=-1, keepdim=True)
    p = torch.clamp(p, 1e-7, 1 - 1e-7)
    return -(y_true * torch.log(p)).sum(dim=-1)


# ------------------------------------------------------------------------------------
# SSD loss  (L/keras_loss_function/keras_ssd_loss.py:53-211)
# ------------------------------------------------------------------------------------
def smooth_l1(y_true, y_pred):
    """keras_ssd_loss.py:72-75."""
    d = y_true - y_pred
    a = d.abs()
    return torch.where(a < 1.0, 0.5 * d * d, a - 0.5).sum(dim=-1)


def log_loss(y_true, y_pred):
    """keras_ssd_loss.py:93-95."""
    return -(y_true * torch.log(torch.clamp_min(y_pred, 1e-15))).sum(dim=-1)


def ssd_loss(y_true, y_pred, neg_pos_ratio=3, n_neg_min=0, alpha=1.0, return_parts=False):
    """SSDLoss.compute_loss (keras_ssd_loss.py:98-211): returns the (batch,) vector Keras
    receives; Keras's own mean over the batch then yields sum(...)/n_positive."""
    batch, n_boxes = y_pred.shape[0], y_pred.shape[1]
    cls = log_loss(y_true[:, :, :-12], y_pred[:, :, :-12])
    loc = smooth_l1(y_true[:, :, -12:-8], y_pred[:, :, -12:-8])
    negatives = y_true[:, :, 0]
    positives = y_true[:, :, 1:-12].max(dim=-1).values
    n_positive = positives.sum()
    pos_cls = (cls * positives).sum(dim=-1)
    neg_all = cls * negatives
    n_neg_losses = int((neg_all != 0).sum())
    n_keep = min(max(neg_pos_ratio * int(n_positive.item()), n_neg_min), n_neg_losses)
    if n_neg_losses == 0:
        neg_cls = torch.zeros(batch, dtype=y_pred.dtype)
        keep = torch.zeros(batch, n_boxes, dtype=y_pred.dtype)
    else:
        flat = neg_all.reshape(-1)
        _, idx = torch.topk(flat.detach(), k=n_keep, sorted=False)
        keep = torch.zeros_like(flat)
        keep[idx] = 1.0
        keep = keep.reshape(batch, n_boxes)
        neg_cls = (cls * keep).sum(dim=-1)
    loc_loss = (loc * positives).sum(dim=-1)
    total = (pos_cls + neg_cls + alpha * loc_loss) / torch.clamp_min(n_positive, 1.0)
    total = total * float(batch)
    if return_parts:
        return total, dict(cls=cls, loc=loc, positives=positives, negatives=negatives,
                           keep=keep, n_positive=n_positive, n_keep=n_keep)
    return total


# ------------------------------------------------------------------------------------
# Optimizer / regulariser  (Keras 2.2.4)
# ------------------------------------------------------------------------------------
def l2_penalty(kernel, l2):
    """keras.regularizers.l2(l): l * sum(w^2) (no 1/2); used on SSD-head kernels only
    (L/models/keras_ssd300_dct_j2d_resnet.py:490...673)."""
    return l2 * (kernel * kernel).sum()


def sgd_keras_step(param, grad, velocity, lr, momentum, decay, iterations, nesterov):
    """keras.optimizers.SGD.get_updates: lr_t = lr/(1+decay*iterations);
    v = momentum*v - lr_t*g; p += momentum*v - lr_t*g (Nesterov) or p += v.
    Call sites: L/training_dct_pascal_j2d_resnet.py:152; C/config/resnet/config_file.py:58-63."""
    lr_t = lr / (1.0 + decay * iterations)
    v = momentum * velocity - lr_t * grad
    if nesterov:
        new_p = param + momentum * v - lr_t * grad
    else:
        new_p = param + v
    return new_p, v


# ------------------------------------------------------------------------------------
# Initialisers (Keras 2.2.4) -- only used to create seeded synthetic weights
# ------------------------------------------------------------------------------------
def he_normal(shape, gen):
    """VarianceScaling(scale=2, mode='fan_in', distribution='normal'): truncated normal,
    stddev = sqrt(2/fan_in)/0.87962566103423978."""
    receptive = 1
    for s in shape[:-2]:
        receptive *= s
    fan_in = shape[-2] * receptive
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    t = torch.empty(shape, dtype=torch.float64)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t


def glorot_uniform(shape, gen, fan_in=None, fan_out=None):
    receptive = 1
    for s in shape[:-2]:
        receptive *= s
    if fan_in is None:
        fan_in = shape[-2] * receptive
    if fan_out is None:
        fan_out = shape[-1] * receptive
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, dtype=torch.float64, generator=gen) * 2 - 1) * limit

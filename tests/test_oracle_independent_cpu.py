"""The oracle (torch-functional restatement) against INDEPENDENT restatements written as explicit numpy loops that
follow the Keras / TensorFlow definitions and the reference's SSDLoss line by line, plus autograd self-consistency
(gradcheck) of every differentiable op.  These are the checks that stand in for the unavailable Keras 2.2.4 / TF 1.8
outputs ("parity unpinned")."""
import numpy as np
import pytest
import torch

from oracle import keras_ops as ko


def loop_conv2d(x, k, bias, strides, padding, dilation):
    """TensorFlow conv2d semantics: SAME pads total = max((ceil(H/s)-1)*s + (k-1)*d + 1 - H, 0), before = total // 2."""
    b, h, w, ci = x.shape
    kh, kw, _, co = k.shape
    (sh, sw), (dh, dw) = strides, dilation
    if padding == "same":
        oh, ow = -(-h // sh), -(-w // sw)
        th = max((oh - 1) * sh + (kh - 1) * dh + 1 - h, 0)
        tw = max((ow - 1) * sw + (kw - 1) * dw + 1 - w, 0)
        pt, pl = th // 2, tw // 2
    else:
        oh, ow = (h - (kh - 1) * dh - 1) // sh + 1, (w - (kw - 1) * dw - 1) // sw + 1
        pt = pl = 0
    y = np.zeros((b, oh, ow, co))
    for n in range(b):
        for i in range(oh):
            for j in range(ow):
                for a in range(kh):
                    for c in range(kw):
                        hh, ww = i * sh + a * dh - pt, j * sw + c * dw - pl
                        if 0 <= hh < h and 0 <= ww < w:
                            y[n, i, j] += x[n, hh, ww] @ k[a, c]
    return y + (bias if bias is not None else 0.0)


@pytest.mark.parametrize("kh,stride,padding,dil", [(3, 1, "same", 1), (2, 1, "same", 1), (1, 2, "valid", 1), (3, 2, "valid", 1),
                                                   (3, 1, "same", 6), (7, 2, "valid", 1), (3, 1, "valid", 1)])
def test_conv2d_against_loops(kh, stride, padding, dil):
    rng = np.random.default_rng(kh * 10 + stride)
    x = rng.normal(size=(2, 9, 8, 3))
    k = rng.normal(size=(kh, kh, 3, 4))
    bias = rng.normal(size=4)
    got = ko.conv2d(torch.from_numpy(x), torch.from_numpy(k), torch.from_numpy(bias), (stride, stride), padding, (dil, dil))
    np.testing.assert_allclose(got.numpy(), loop_conv2d(x, k, bias, (stride, stride), padding, (dil, dil)), atol=1e-12)


def test_conv2d_transpose_against_loops():
    """Keras Conv2DTranspose(k=2, s=2, 'valid'): out[n, 2i+a, 2j+b, co] = bias + sum_ci x[n,i,j,ci] * W[a,b,co,ci]."""
    rng = np.random.default_rng(1)
    x, k, bias = rng.normal(size=(2, 3, 4, 5)), rng.normal(size=(2, 2, 6, 5)), rng.normal(size=6)
    want = np.zeros((2, 6, 8, 6))
    for i in range(3):
        for j in range(4):
            for a in range(2):
                for b in range(2):
                    want[:, 2 * i + a, 2 * j + b] = x[:, i, j] @ k[a, b].T + bias
    got = ko.conv2d_transpose(torch.from_numpy(x), torch.from_numpy(k), torch.from_numpy(bias), (2, 2))
    np.testing.assert_allclose(got.numpy(), want, atol=1e-12)


def test_max_pool_and_upsampling_against_loops():
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, 5, 5, 3))
    want = np.full((2, 5, 5, 3), -np.inf)
    for i in range(5):
        for j in range(5):
            for a in (-1, 0, 1):
                for b in (-1, 0, 1):
                    if 0 <= i + a < 5 and 0 <= j + b < 5:
                        want[:, i, j] = np.maximum(want[:, i, j], x[:, i + a, j + b])
    np.testing.assert_array_equal(ko.max_pool_3x3_s1_same(torch.from_numpy(x)).numpy(), want)
    up = ko.upsampling_nearest_2x(torch.from_numpy(x)).numpy()
    assert up.shape == (2, 10, 10, 3)
    for i in range(10):
        for j in range(10):
            np.testing.assert_array_equal(up[:, i, j], x[:, i // 2, j // 2])


def numpy_ssd_loss(y_true, y_pred, neg_pos_ratio=3, n_neg_min=0, alpha=1.0):
    """keras_ssd_loss.py:98-211 step by step in numpy (explicit sort instead of tf.nn.top_k)."""
    batch, n_boxes = y_pred.shape[:2]
    cls_loss = -np.sum(y_true[:, :, :-12] * np.log(np.maximum(y_pred[:, :, :-12], 1e-15)), axis=-1)      # :77-96
    d = y_true[:, :, -12:-8] - y_pred[:, :, -12:-8]
    loc_loss = np.sum(np.where(np.abs(d) < 1.0, 0.5 * d ** 2, np.abs(d) - 0.5), axis=-1)                 # :53-75
    negatives = y_true[:, :, 0]
    positives = np.max(y_true[:, :, 1:-12], axis=-1)
    n_positive = positives.sum()
    pos_class_loss = np.sum(cls_loss * positives, axis=-1)
    neg_class_loss_all = cls_loss * negatives
    n_neg_losses = np.count_nonzero(neg_class_loss_all)
    n_negative_keep = int(min(max(neg_pos_ratio * int(n_positive), n_neg_min), n_neg_losses))
    if n_neg_losses == 0:
        neg_class_loss = np.zeros(batch)
    else:
        flat = neg_class_loss_all.reshape(-1)
        keep = np.zeros_like(flat)
        keep[np.argsort(-flat, kind="stable")[:n_negative_keep]] = 1.0
        neg_class_loss = np.sum(cls_loss * keep.reshape(batch, n_boxes), axis=-1)
    class_loss = pos_class_loss + neg_class_loss
    loc = np.sum(loc_loss * positives, axis=-1)
    return (class_loss + alpha * loc) / max(1.0, n_positive) * batch


@pytest.mark.parametrize("seed,ratio,n_min", [(0, 3, 0), (1, 3, 0), (2, 1, 5), (3, 3, 0)])
def test_ssd_loss_against_numpy_restatement(seed, ratio, n_min):
    rng = np.random.default_rng(seed)
    b, n, c = 3, 40, 5
    logits = rng.normal(size=(b, n, c))
    probs = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    y_pred = np.concatenate([probs, rng.normal(size=(b, n, 4)), np.zeros((b, n, 8))], axis=-1)
    y_true = np.zeros((b, n, c + 12))
    y_true[..., 0] = 1.0
    pos = rng.random((b, n)) < (0.15 if seed != 3 else 0.0)           # seed 3: no positives at all
    cls = rng.integers(1, c, size=(b, n))
    for i in range(b):
        for j in range(n):
            if pos[i, j]:
                y_true[i, j, 0] = 0.0
                y_true[i, j, cls[i, j]] = 1.0
                y_true[i, j, c:c + 4] = rng.normal(size=4) * 1.5
    neutral = (rng.random((b, n)) < 0.1) & ~pos
    y_true[neutral, 0] = 0.0
    got = ko.ssd_loss(torch.from_numpy(y_true), torch.from_numpy(y_pred), neg_pos_ratio=ratio, n_neg_min=n_min)
    np.testing.assert_allclose(got.numpy(), numpy_ssd_loss(y_true, y_pred, ratio, n_min), rtol=1e-12, atol=1e-12)


def test_batch_norm_and_losses_gradcheck():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 2, 2, 4, generator=g, dtype=torch.float64, requires_grad=True)
    gam = torch.rand(4, generator=g, dtype=torch.float64, requires_grad=True)
    bet = torch.randn(4, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b, c: ko.batch_norm_train(a, b, c)[0], (x, gam, bet))
    logits = torch.randn(6, 5, generator=g, dtype=torch.float64, requires_grad=True)
    onehot = torch.eye(5, dtype=torch.float64)[torch.tensor([0, 1, 2, 3, 4, 0])]
    assert torch.autograd.gradcheck(lambda z: ko.categorical_crossentropy(onehot, ko.softmax(z)), (logits,))
    k = torch.randn(2, 2, 3, 4, generator=g, dtype=torch.float64, requires_grad=True)
    xs = torch.randn(1, 2, 3, 4, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: ko.conv2d_transpose(a, b, None, (2, 2)), (xs, k))


def test_ssd_loss_gradient_is_p_minus_y_on_kept_boxes():
    """d loss / d logits through the softmax = (p - y) / n_positive * batch... on positives and mined negatives, 0 elsewhere
    (SURVEY 8(a) row 17)."""
    rng = np.random.default_rng(7)
    b, n, c = 2, 12, 4
    logits = torch.tensor(rng.normal(size=(b, n, c)), requires_grad=True)
    y_true = np.zeros((b, n, c + 12))
    y_true[..., 0] = 1.0
    y_true[0, 3, 0], y_true[0, 3, 2] = 0.0, 1.0
    y_true[1, 5, 0], y_true[1, 5, 1] = 0.0, 1.0
    yt = torch.from_numpy(y_true)
    probs = ko.softmax(logits)
    y_pred = torch.cat([probs, torch.zeros(b, n, 12, dtype=torch.float64)], dim=-1)
    loss = ko.ssd_loss(yt, y_pred).mean()
    loss.backward()
    p = probs.detach().numpy()
    neg = -np.log(p[..., 0]) * y_true[..., 0]
    kept = np.zeros(b * n)
    kept[np.argsort(-neg.reshape(-1))[:6]] = 1.0                       # 3 * n_positive
    sel = kept.reshape(b, n) + y_true[..., 1:c].max(-1)
    want = (p - y_true[..., :c]) * sel[..., None] / 2.0                # / n_positive (2); the batch factor cancels in the mean
    np.testing.assert_allclose(logits.grad.numpy(), want, atol=1e-12)

"""CPU: sanity pins of the oracle itself (hand-computed small cases and autograd self-consistency)."""
import numpy as np
import torch

from oracle import keras_ops as ko


def test_same_padding_even_kernel_is_asymmetric():
    x = torch.arange(16.0).reshape(1, 4, 4, 1)
    k = torch.ones(2, 2, 1, 1)
    y = ko.conv2d(x, k, None, (1, 1), "same")
    assert y.shape == (1, 4, 4, 1)
    assert float(y[0, 0, 0, 0]) == 0 + 1 + 4 + 5           # window starts AT the pixel: no padding before
    assert float(y[0, 3, 3, 0]) == 15                      # one row/column of zeros after
    y3 = ko.conv2d(x, torch.ones(3, 3, 1, 1), None, (1, 1), "same", (1, 1))
    assert float(y3[0, 0, 0, 0]) == 0 + 1 + 4 + 5
    ys = ko.conv2d(x, torch.ones(1, 1, 1, 1), None, (2, 2), "valid")
    assert ys.reshape(-1).tolist() == [0, 2, 8, 10]        # valid 1x1 stride 2 samples pixels 0, 2


def test_conv_transpose_layout():
    x = torch.zeros(1, 2, 2, 1)
    x[0, 1, 0, 0] = 3.0
    k = torch.arange(8.0).reshape(2, 2, 2, 1)              # (kh, kw, out, in)
    y = ko.conv2d_transpose(x, k, None, (2, 2))
    assert y.shape == (1, 4, 4, 2)
    assert y[0, 2:4, 0:2, :].reshape(-1).tolist() == (3 * k[..., 0]).reshape(-1).tolist()


def test_batch_norm_and_moving_update():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3, 3, 5, generator=g, dtype=torch.float64)
    y, m, v = ko.batch_norm_train(x, torch.ones(5, dtype=torch.float64), torch.zeros(5, dtype=torch.float64))
    assert torch.allclose(y.mean(dim=(0, 1, 2)), torch.zeros(5, dtype=torch.float64), atol=1e-12)
    assert torch.allclose((y ** 2).mean(dim=(0, 1, 2)), v / (v + 1e-3), atol=1e-12)
    z, o = torch.zeros(5, dtype=torch.float64), torch.ones(5, dtype=torch.float64)
    nm, nv = ko.batch_norm_moving_update(z, o, m, v, 36)
    assert torch.allclose(nm, 0.01 * m) and torch.allclose(nv, 0.99 + 0.01 * v * 36 / 35)


def test_ssd_loss_hand_case():
    """2 images x 4 boxes, 3 classes (+12): 1 positive, mining keeps 3 of the 5 non-zero negatives."""
    yt = torch.zeros(2, 4, 15)
    yp = torch.zeros(2, 4, 15)
    probs = torch.tensor([[0.7, 0.2, 0.1], [0.5, 0.25, 0.25], [0.9, 0.05, 0.05], [0.2, 0.4, 0.4],
                          [0.6, 0.3, 0.1], [0.3, 0.3, 0.4], [1.0, 0.0, 0.0], [0.8, 0.1, 0.1]])
    yp[..., :3] = probs.reshape(2, 4, 3)
    yt[..., 0] = 1.0
    yt[0, 1, 0], yt[0, 1, 2] = 0.0, 1.0                    # positive of class 2 at (0,1)
    yt[1, 3, 0] = 0.0                                       # neutral box
    yt[0, 1, 3:7] = torch.tensor([0.5, -2.0, 0.1, 0.0])    # offsets (pred 0): 0.125 + 1.5 + 0.005 + 0
    vec, parts = ko.ssd_loss(yt, yp, return_parts=True)
    neg = -torch.log(torch.tensor([0.7, 0.9, 0.2, 0.6, 0.3]))   # (1,2) has p0 = 1 -> zero loss, not counted
    kept = torch.sort(neg, descending=True).values[:3].sum()
    expect = (-np.log(0.25) + float(kept) + (0.125 + 1.5 + 0.005)) / 1.0
    assert parts["n_keep"] == 3 and float(parts["n_positive"]) == 1
    assert abs(float(vec.mean()) - expect) < 1e-6
    yt0 = yt.clone()
    yt0[0, 1, 2], yt0[0, 1, 0] = 0.0, 1.0                  # no positives -> n_keep = max(0, n_neg_min) = 0
    assert float(ko.ssd_loss(yt0, yp).sum()) == 0.0


def test_sgd_keras_vs_manual():
    p, g, v = torch.tensor([1.0]), torch.tensor([0.5]), torch.tensor([0.2])
    np_, nv = ko.sgd_keras_step(p, g, v, lr=0.1, momentum=0.9, decay=0.0, iterations=0, nesterov=False)
    assert abs(float(nv) - (0.18 - 0.05)) < 1e-7 and abs(float(np_) - 1.13) < 1e-7
    np2, _ = ko.sgd_keras_step(p, g, v, lr=0.1, momentum=0.9, decay=0.0, iterations=0, nesterov=True)
    assert abs(float(np2) - (1.0 + 0.9 * 0.13 - 0.05)) < 1e-7
    _, nv3 = ko.sgd_keras_step(p, g, v, lr=0.1, momentum=0.9, decay=0.5, iterations=2, nesterov=False)
    assert abs(float(nv3) - (0.18 - 0.05 * 0.5)) < 1e-7


def test_l2norm_and_gradcheck_small():
    x = torch.randn(2, 3, 3, 4, dtype=torch.float64, requires_grad=True)
    gam = torch.rand(4, dtype=torch.float64, requires_grad=True)
    y = ko.l2_normalization(x, gam)
    assert torch.allclose((y / gam).pow(2).sum(-1), torch.ones(2, 3, 3, dtype=torch.float64))
    assert torch.autograd.gradcheck(lambda a, b: ko.l2_normalization(a, b), (x, gam))
    k = torch.randn(2, 2, 4, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: ko.conv2d(a, b, None, (1, 1), "same"), (x, k))

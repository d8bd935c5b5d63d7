"""GPU parity of the HBM-bound kernels (BatchNormalization, ReLU/Add, L2Normalization, pooling,
softmax, SSD loss with hard-negative mining, categorical cross-entropy, SGD) against the CPU oracle.
Tolerance 1e-3 relative (north_star) unless a test states a tighter one."""
import numpy as np
import pytest
import torch

from oracle import keras_ops as ko

pytestmark = pytest.mark.gpu


def close(a, ref, rel=1e-3, abs_=1e-5):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    return float((a - ref).abs().max()) <= rel * float(ref.abs().max()) + abs_


@pytest.fixture()
def E(cuda):
    from jpeg_detection_resnet_ssd_amd import engine
    return engine


@pytest.mark.parametrize("shape", [(3, 19, 19, 128), (2, 38, 38, 64), (5, 3, 3, 30), (4, 1, 1, 256)])
def test_batchnorm_train_forward_backward(shape, cuda, E):
    g = torch.Generator().manual_seed(11)
    c = shape[-1]
    rows = int(np.prod(shape[:-1]))
    x = torch.randn(shape, generator=g) * 3 + torch.randn(c, generator=g) * 5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    dy = torch.randn(shape, generator=g)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr, mean, var = ko.batch_norm_train(xr, gr, br)
    out = ko.relu(yr)
    out.backward(dy.double())
    mm0, mv0 = torch.zeros(c), torch.ones(c)
    nm, nv = ko.batch_norm_moving_update(mm0.double(), mv0.double(), mean.detach(), var.detach(), rows)

    xd, gd, bd, dyd = x.to(cuda), gamma.to(cuda), beta.to(cuda), dy.to(cuda)
    nr = E.query("dj_reduce_rows", rows)
    part = torch.empty(nr, 2, c, device=cuda)
    scale, shift, smean, sinv = (torch.empty(c, device=cuda) for _ in range(4))
    mm, mv = mm0.to(cuda), mv0.to(cuda)
    E.call("dj_colstats_partial", xd, rows, c, c, part)
    E.call("dj_bn_train_finalize", part, nr, rows, None, gd, bd, 1e-3, 0.99, mm, mv, scale, shift, smean, sinv, c)
    y = torch.empty_like(xd)
    E.call("dj_affine_act", xd, c, scale, shift, None, 0, None, None, y, c, rows, c, 1)
    torch.cuda.synchronize()
    assert close(y, out)
    assert close(mm, nm) and close(mv, nv)
    # backward with the ReLU mask recomputed from the affine (mask_mode 2)
    k0, k1, k2, dg, db = (torch.empty(c, device=cuda) for _ in range(5))
    dz = torch.empty_like(xd)
    E.call("dj_bn_bwd_reduce", dyd, c, xd, c, None, 0, smean, sinv, scale, shift, 2, rows, c, part)
    E.call("dj_bn_bwd_finalize", part, nr, rows, gd, smean, sinv, dg, db, k0, k1, k2, c)
    E.call("dj_bn_bwd_apply", dyd, c, xd, c, None, 0, scale, shift, 2, k0, k1, k2, dz, c, rows, c, None, 0, 0)
    torch.cuda.synchronize()
    assert close(dg, gr.grad) and close(db, br.grad)
    assert close(dz, xr.grad, rel=2e-3)
    # mask_mode 1 (mask from the materialised output) gives the same result
    dz1 = torch.empty_like(xd)
    E.call("dj_bn_bwd_reduce", dyd, c, xd, c, y, c, smean, sinv, None, None, 1, rows, c, part)
    E.call("dj_bn_bwd_finalize", part, nr, rows, gd, smean, sinv, dg, db, k0, k1, k2, c)
    dm = torch.full((rows, c), 2.0, device=cuda)
    E.call("dj_bn_bwd_apply", dyd, c, xd, c, y, c, None, None, 1, k0, k1, k2, dz1, c, rows, c, dm, c, 1)
    torch.cuda.synchronize()
    assert close(dz1, xr.grad, rel=2e-3)
    # ... and the second output accumulated the ReLU-masked upstream gradient (the identity shortcut's gradient)
    assert torch.equal(dm.cpu(), (2.0 + dyd * (y > 0)).reshape(rows, c).cpu())


def test_add_relu_and_relu_bwd_and_copy(cuda, E):
    g = torch.Generator().manual_seed(5)
    rows, c = 700, 96
    a, b = torch.randn(rows, c, generator=g), torch.randn(rows, c, generator=g)
    sa, ta, sb, tb = (torch.randn(c, generator=g) for _ in range(4))
    ref = ko.relu(a * sa + ta + b * sb + tb)
    y = torch.empty(rows, c, device=cuda)
    E.call("dj_affine_act", a.to(cuda), c, sa.to(cuda), ta.to(cuda), b.to(cuda), c, sb.to(cuda), tb.to(cuda), y, c,
           rows, c, 1)
    torch.cuda.synchronize()
    assert close(y, ref, rel=1e-5)
    dy = torch.randn(rows, c, generator=g)
    dx = torch.ones(rows, c, device=cuda)
    E.call("dj_relu_bwd", dy.to(cuda), c, y, c, dx, c, rows, c, 1)
    torch.cuda.synchronize()
    assert close(dx, 1.0 + dy * (ref > 0), rel=1e-6)
    # strided copy into a channel slice, then accumulate
    dst = torch.zeros(rows, 200, device=cuda)
    E.call("dj_copy2d", a.to(cuda), c, dst[:, 50:50 + c], 200, rows, c, 0)
    E.call("dj_copy2d", a.to(cuda), c, dst[:, 50:50 + c], 200, rows, c, 1)
    torch.cuda.synchronize()
    assert torch.equal(dst[:, 50:50 + c].cpu(), 2 * a) and float(dst[:, :50].abs().max()) == 0


def test_upsample_and_gap(cuda, E):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 19, 19, 128, generator=g)
    y = torch.zeros(2, 38, 38, 192, device=cuda)
    E.call("dj_upsample2x", x.to(cuda), 128, y[..., 64:], 192, 2, 19, 19, 128)
    torch.cuda.synchronize()
    assert torch.equal(y[..., 64:].cpu(), ko.upsampling_nearest_2x(x))
    p = torch.empty(2, 128, device=cuda)
    E.call("dj_global_avg_pool_fwd", x.to(cuda), p, 2, 361, 128)
    torch.cuda.synchronize()
    assert close(p, ko.global_average_pooling(x), rel=1e-5)


@pytest.mark.parametrize("shape", [(2, 38, 38, 384), (3, 10, 10, 1024), (2, 38, 38, 64)])
def test_l2norm(shape, cuda, E):
    g = torch.Generator().manual_seed(8)
    c = shape[-1]
    rows = int(np.prod(shape[:-1]))
    x = torch.randn(shape, generator=g) * 4
    x[0, 0, 0] = 0.0  # a zero pixel hits the 1e-12 clamp
    gamma = torch.rand(c, generator=g) * 5 + 18
    dy = torch.randn(shape, generator=g)
    xr, gr = x.double().requires_grad_(True), gamma.double().requires_grad_(True)
    yr = ko.l2_normalization(xr, gr)
    yr.backward(dy.double())
    xd, gd, dyd = x.to(cuda), gamma.to(cuda), dy.to(cuda)
    y, rn = torch.empty_like(xd), torch.empty(rows, device=cuda)
    E.call("dj_l2norm_fwd", xd, c, gd, y, c, rn, rows, c)
    nr = E.query("dj_reduce_rows", rows)
    part = torch.empty(nr, 2, c, device=cuda)
    dx, dg = torch.empty_like(xd), torch.empty(c, device=cuda)
    E.call("dj_l2norm_bwd", dyd, c, xd, c, gd, rn, dx, c, part, rows, c, 0)
    E.call("dj_colreduce_finalize", part, nr, c, 0, dg, 0)
    torch.cuda.synchronize()
    assert close(y, yr)
    mask = torch.ones(shape, dtype=torch.bool)
    mask[0, 0, 0] = False  # autograd of rsqrt(clamp) at the clamp differs from the constant-norm gradient
    assert close(dx.cpu()[mask], xr.grad[mask])
    assert close(dg, gr.grad)


@pytest.mark.parametrize("saved", [False, True])
@pytest.mark.parametrize("hw,ch", [(5, 256), (10, 256), (7, 30)])   # 30 channels: the scalar kernels (C % 4 != 0)
def test_maxpool(hw, ch, saved, cuda, E):
    g = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn(3, hw, hw, ch, generator=g))     # exact ties (zeros): first maximum must win
    dy = torch.randn(3, hw, hw, ch, generator=g)
    xr = x.double().requires_grad_(True)
    yr = ko.max_pool_3x3_s1_same(xr)
    yr.backward(dy.double())
    y, dx = torch.empty(x.shape, device=cuda), torch.empty(x.shape, device=cuda)
    am = torch.empty(x.numel(), dtype=torch.uint8, device=cuda) if saved else None
    E.call("dj_maxpool2d_fwd", x.to(cuda), y, 3, hw, hw, ch, hw, hw, 3, 3, 1, 1, 1, 1, 0, am)
    E.call("dj_maxpool2d_bwd", None if saved else x.to(cuda), dy.to(cuda), dx, 3, hw, hw, ch, hw, hw, 3, 3, 1, 1, 1, 1,
           0, 0, am)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().double(), yr.detach())
    # ties: torch's CPU max_pool2d also routes the gradient to the first maximum in window order
    assert close(dx, xr.grad, rel=1e-5)


@pytest.mark.parametrize("saved", [False, True])
def test_maxpool_stem_zero_padded_stride2(saved, cuda, E):
    """ZeroPadding2D(1) + MaxPooling2D((3,3), strides 2): the ResNet50RGB stem (zeros take part in the max)."""
    g = torch.Generator().manual_seed(19)
    b, hw, c = 2, 12, 64
    x = torch.randn(b, hw, hw, c, generator=g)        # negative values: a zero pad can win
    xr = x.double().requires_grad_(True)
    yr = ko.max_pool(ko.zero_padding(xr, ((1, 1), (1, 1))), (3, 3), (2, 2), "valid")
    oh = yr.shape[1]
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    y, dx = torch.empty(yr.shape, device=cuda), torch.ones(x.shape, device=cuda)
    am = torch.empty(y.numel(), dtype=torch.uint8, device=cuda) if saved else None
    E.call("dj_maxpool2d_fwd", x.to(cuda), y, b, hw, hw, c, oh, oh, 3, 3, 2, 2, 1, 1, 1, am)
    E.call("dj_maxpool2d_bwd", x.to(cuda), dy.to(cuda), dx, b, hw, hw, c, oh, oh, 3, 3, 2, 2, 1, 1, 1, 1, am)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().double(), yr.detach())
    assert close(dx - 1.0, xr.grad, rel=1e-5)


@pytest.mark.parametrize("rows,c", [(5000, 21), (64, 1000)])
def test_softmax(rows, c, cuda, E):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(rows, c, generator=g) * 4
    dp = torch.randn(rows, c, generator=g)
    xr = x.double().requires_grad_(True)
    pr = ko.softmax(xr)
    pr.backward(dp.double())
    p, dx = torch.empty(rows, c, device=cuda), torch.empty(rows, c, device=cuda)
    E.call("dj_softmax_fwd", x.to(cuda), p, rows, c)
    E.call("dj_softmax_bwd", p, dp.to(cuda), c, dx, rows, c, 0)
    torch.cuda.synchronize()
    assert close(p, pr) and close(dx, xr.grad)


def _ssd_case(batch, nbox, seed, pos_frac):
    g = torch.Generator().manual_seed(seed)
    n_cls = 21
    logits = torch.randn(batch, nbox, n_cls, generator=g) * 2
    loc = torch.randn(batch, nbox, 4, generator=g) * 1.5
    anchors = torch.rand(batch, nbox, 8, generator=g)
    y_true = torch.zeros(batch, nbox, 33)
    u = torch.rand(batch, nbox, generator=g)
    cls = torch.randint(1, n_cls, (batch, nbox), generator=g)
    pos = u < pos_frac
    neutral = (u >= pos_frac) & (u < pos_frac + 0.02)
    neg = ~pos & ~neutral
    y_true[..., 0][neg] = 1.0
    y_true.view(-1, 33)[pos.view(-1).nonzero().squeeze(1), cls.view(-1)[pos.view(-1)]] = 1.0
    y_true[..., 21:25] = torch.randn(batch, nbox, 4, generator=g)
    y_true[..., 25:] = anchors
    return logits, loc, anchors, y_true


@pytest.mark.parametrize("batch,nbox,pos_frac", [(4, 8732, 0.004), (2, 500, 0.0), (3, 700, 0.4), (32, 8732, 0.003)])
def test_ssd_loss(batch, nbox, pos_frac, cuda, E):
    logits, loc, anchors, y_true = _ssd_case(batch, nbox, 21, pos_frac)
    lr = logits.double().requires_grad_(True)
    locr = loc.double().requires_grad_(True)
    ypr = torch.cat([ko.softmax(lr), locr, anchors.double()], dim=2)
    ypr.retain_grad()
    vec, parts = ko.ssd_loss(y_true.double(), ypr, return_parts=True)
    loss = vec.mean()
    loss.backward()

    yp = torch.cat([torch.softmax(logits, -1), loc, anchors], dim=2).to(cuda).contiguous()
    yt = y_true.to(cuda)
    n = batch * nbox
    ws = torch.empty(E.query("dj_ssd_loss_workspace_floats", n), device=cuda)
    out = torch.zeros(8, device=cuda)
    d = torch.full((batch, nbox, 33), float("nan"), device=cuda)
    E.call("dj_ssd_loss_fwd", yt, yp, n, 21, 3, 0, 1.0, ws, out)
    E.call("dj_ssd_loss_bwd", yt, yp, n, 21, 1.0, 1.0, ws, out, d)
    torch.cuda.synchronize()
    o = out.cpu()
    assert abs(float(o[0]) - float(loss)) <= 1e-3 * abs(float(loss)) + 1e-6
    assert int(o[1]) == int(parts["n_positive"]) and int(o[2]) == parts["n_keep"]
    keep = ws[4 * n:5 * n].cpu().view(batch, nbox)
    assert int(keep.sum()) == parts["n_keep"]
    # the kept set may differ from torch.topk only among exactly tied losses
    kr = parts["keep"]
    diff = (keep.double() != kr)
    if diff.any():
        negl = (parts["cls"] * parts["negatives"]).detach()
        assert negl[diff].unique().numel() == 1
    assert close(d, ypr.grad, rel=1e-3, abs_=1e-7)


def test_categorical_crossentropy(cuda, E):
    g = torch.Generator().manual_seed(12)
    rows, c = 64, 1000
    logits = torch.randn(rows, c, generator=g) * 3
    y = torch.zeros(rows, c)
    y[torch.arange(rows), torch.randint(0, c, (rows,), generator=g)] = 1.0
    lr = logits.double().requires_grad_(True)
    pr = ko.softmax(lr)
    pr.retain_grad()
    loss = ko.categorical_crossentropy(y.double(), pr).mean()
    loss.backward()
    p = torch.softmax(logits, -1).to(cuda)
    lrows, dp, out = torch.empty(rows, device=cuda), torch.empty(rows, c, device=cuda), torch.zeros(8, device=cuda)
    E.call("dj_categorical_crossentropy", y.to(cuda), p, rows, c, 1.0, lrows, dp, out)
    dx = torch.empty(rows, c, device=cuda)
    E.call("dj_softmax_bwd", p, dp, c, dx, rows, c, 0)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(loss)) <= 1e-4 * float(loss)
    assert close(dx, lr.grad)


@pytest.mark.parametrize("nesterov", [False, True])
def test_sgd_keras_formula(nesterov, cuda, E):
    g = torch.Generator().manual_seed(13)
    n = 100003
    p, gr, v = (torch.randn(n, generator=g) for _ in range(3))
    lr_t = 0.1 / (1 + 1e-4 * 7)
    g_eff = gr.double() * 0.125 + 2 * 5e-4 * p.double()
    pn, vn = ko.sgd_keras_step(p.double(), g_eff, v.double(), 0.1, 0.9, 1e-4, 7, nesterov)
    pd, gd, vd = p.to(cuda), gr.to(cuda), v.to(cuda)
    ss = torch.zeros(1, device=cuda)
    E.call("dj_sgd_momentum_update", pd, gd, vd, n, lr_t, 0.9, int(nesterov), 5e-4, 0.125, ss)
    torch.cuda.synchronize()
    assert close(pd, pn, rel=1e-6) and close(vd, vn, rel=1e-6)
    assert abs(float(ss) - float((p.double() ** 2).sum())) <= 1e-4 * float((p.double() ** 2).sum())


@pytest.mark.parametrize("rows,c,ld", [(3200, 126, 126), (800, 24, 24), (37, 5, 9), (8192, 512, 512)])
def test_colsum_direct(rows, c, ld, cuda, E):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, ld, generator=g)
    out = torch.full((c,), 1.5, device=cuda)
    E.call("dj_colsum_direct", x.to(cuda), rows, c, ld, out, 1)
    torch.cuda.synchronize()
    assert close(out.cpu(), 1.5 + x[:, :c].double().sum(0), rel=1e-6)


def test_colsum_multi(cuda):
    """dj_colsum_multi: 35 tensors of different shapes (two launches of <= 32 parts) == dj_colsum_direct one by one, bit
    for bit (same kernel body)."""
    from jpeg_detection_resnet_ssd_amd import engine
    g = torch.Generator().manual_seed(4)
    shapes = [(3200, 126, 126), (800, 24, 24), (37, 5, 9), (8192, 512, 512), (32, 84, 84), (288, 16, 16), (1, 3, 7)] * 5
    xs = [torch.randn(r, ld, generator=g).to(cuda) for r, _, ld in shapes]
    outs = [torch.empty(c, device=cuda) for _, c, _ in shapes]
    refs = [torch.empty(c, device=cuda) for _, c, _ in shapes]
    engine.colsum_multi([(x, r, c, ld, o) for x, o, (r, c, ld) in zip(xs, outs, shapes)])()
    for x, o, (r, c, ld) in zip(xs, refs, shapes):
        engine.call("dj_colsum_direct", x, r, c, ld, o, 0)
    torch.cuda.synchronize()
    for o, ref in zip(outs, refs):
        assert torch.equal(o, ref)


def test_two_threads_two_arithmetic_modes_through_the_c_abi(cuda):
    """two host threads launch convolutions at the same time on their own streams, one in exact fp32
    and one in fp16-MFMA mode (per-thread override of the C library), while a third keeps rewriting the tuner's override
    table.  Each thread's results must equal what the same calls give alone: the fp32 thread bit for bit."""
    import threading
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    desc = Kn.make_conv_desc(4, 19, 19, 128, 128, (3, 3), (1, 1), "same", (1, 1))
    x = torch.randn(4, 19, 19, 128, generator=g).to(cuda)
    w = (torch.randn(3, 3, 128, 128, generator=g) * 0.05).to(cuda)
    bias = torch.randn(128, generator=g).to(cuda)

    def run(mode, n, stream, out):
        lib.dj_set_thread_compute_mode(mode)
        lib.dj_conv2d_tune_set(0, desc, 2, 1)      # 64x64 tiles, no split-K: no atomics, so bit-reproducible
        with torch.cuda.stream(stream):
            for i in range(n):
                y = torch.empty(4, 19, 19, 128, device=cuda)
                Kn.conv2d_fwd(desc, x, w, bias, y)
                out.append(y)
        stream.synchronize()
        lib.dj_conv2d_tune_set(0, desc, -1, 1)
        lib.dj_set_thread_compute_mode(-1)

    alone32, alone16 = [], []
    run(0, 1, torch.cuda.Stream(), alone32)
    run(1, 1, torch.cuda.Stream(), alone16)
    assert float((alone32[0] - alone16[0]).abs().max()) > 1e-5      # the two modes really differ
    stop = threading.Event()

    def retune():
        other = Kn.make_conv_desc(4, 19, 19, 128, 128, (1, 1), (1, 1), "valid", (1, 1))
        i = 0
        while not stop.is_set():
            lib.dj_conv2d_tune_set(0, other, i % lib.dj_conv2d_tune_configs(), 1)
            i += 1
        lib.dj_conv2d_tune_set(0, other, -1, 1)

    out32, out16, errs = [], [], []

    def guarded(*a):
        try:
            run(*a)
        except Exception as e:     # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=guarded, args=(0, 40, torch.cuda.Stream(), out32)),
          threading.Thread(target=guarded, args=(1, 40, torch.cuda.Stream(), out16)), threading.Thread(target=retune)]
    for t in ts:
        t.start()
    ts[0].join()
    ts[1].join()
    stop.set()
    ts[2].join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert len(out32) == 40 and len(out16) == 40
    for y in out32:
        assert torch.equal(y, alone32[0])
    for y in out16:
        assert torch.equal(y, alone16[0])
    assert lib.dj_get_compute_mode() == 0

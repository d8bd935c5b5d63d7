"""Arithmetic modes 3 and 4: fp32 tensors, each product of a convolution GEMM on the bf16 matrix pipe (dj_igemm_h16.h).
K.set_floatx('float32x3') (PREC 3): operands split into a high and a low bf16 half, three MFMAs.  The split leaves 2^-18
of an operand behind and the lo*lo term is dropped: <= ~1e-5 per product, a few 1e-6 rel-L2 after random-sign accumulation.
Bound: 3e-5 rel-L2 per GEMM (50x tighter than the fp16 mode's, 30x inside the 1e-3 parity bar of the exact-fp32 mode).
K.set_floatx('float32x6') (PREC 4): three bf16 pieces (all 24 significant bits), six MFMAs; dropped terms <= 2^-24 of a
product.  Bound: 2e-6 rel-L2 per GEMM -- the bound tests/test_lowp_gpu.py holds the exact-fp32 MFMA kernels to -- and its
error against the fp64 oracle is compared with the exact kernels' on the same operands.
The whole training step: the 1e-3 bar of the exact mode for both."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {"float32x3": 3e-5, "float32x6": 2e-6}
MODES = ["float32x3", "float32x6"]


@pytest.fixture()
def floatx():
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    yield K
    K.set_floatx("float32")


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("geom", [(8, 19, 19, 256, 256, 3, 1), (4, 38, 38, 128, 512, 1, 1), (8, 10, 10, 512, 512, 3, 1),
                                  (4, 38, 38, 256, 128, 1, 2)])
def test_conv_directions_in_split_bf16_arithmetic(geom, mode, floatx):
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    b, h, w, ci, co, k, s = geom
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    bias = torch.randn(co, generator=g)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = ko.conv2d(xr, wr, bias.double(), (s, s), "same")
    dy = torch.randn(*yr.shape, generator=g) * 1e-4
    yr.backward(dy.double())
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), "same", (1, 1))
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    y, dx, dw = torch.empty_like(dyd), torch.empty_like(xd), torch.zeros_like(wd)
    y32, dx32, dw32 = torch.empty_like(dyd), torch.empty_like(xd), torch.zeros_like(wd)
    floatx.set_floatx("float32_mfma")           # fp32 MFMA instructions only (the default mode may name a split kernel)
    Kn.conv2d_fwd(desc, xd, wd, bias.cuda(), y32)
    Kn.conv2d_dgrad(desc, dyd, wd, dx32)
    Kn.conv2d_wgrad(desc, xd, dyd, dw32)
    floatx.set_floatx(mode)
    assert floatx.floatx() == mode
    Kn.conv2d_fwd(desc, xd, wd, bias.cuda(), y)
    Kn.conv2d_dgrad(desc, dyd, wd, dx)
    Kn.conv2d_wgrad(desc, xd, dyd, dw)
    torch.cuda.synchronize()
    e = (rel_l2(y.cpu(), yr.detach()), rel_l2(dx.cpu(), xr.grad), rel_l2(dw.cpu(), wr.grad))
    e32 = (rel_l2(y32.cpu(), yr.detach()), rel_l2(dx32.cpu(), xr.grad), rel_l2(dw32.cpu(), wr.grad))
    print("%s %s: fwd %.2e dgrad %.2e wgrad %.2e  (exact-fp32 MFMA kernels: %.2e %.2e %.2e)" % ((mode, geom) + e + e32))
    assert max(e) <= TOL[mode], e
    if mode == "float32x6":                     # fp32-grade: within 2x of the fp32 MFMA kernels' own distance to fp64
        assert all(a <= 2.0 * b + 1e-7 for a, b in zip(e, e32)), (e, e32)
    assert not torch.equal(y, y32)              # the split-operand kernels did run (the fp32 MFMA rounds differently)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sx,sw,sd", [(1e15, 1e-15, 1e-12), (1e-18, 1e3, 1e10), (3e-30, 1.0, 1e20), (1e25, 1e-10, 1e-20)])
def test_split_modes_keep_the_fp32_exponent_range(mode, sx, sw, sd, floatx):
    """bf16 pieces have the exponent range of fp32: operands far from 1 (activations of 1e15 against weights of 1e-15,
    activations of 3e-30 against gradients of 1e20, ...; every result inside the fp32 range) keep the mode's accuracy -- what an fp16-piece split could not do.  (Documented limits:
    an operand above the largest bf16, 3.39e38, rounds to infinity where fp32 would still be finite; below ~1e-33 the low
    pieces are denormal and the relative accuracy of a product falls back towards bf16x2's.)"""
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    b, h, w, ci, co, k = 4, 19, 19, 256, 128, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, h, w, ci, generator=g) * sx
    wt = torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5 * sw
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = ko.conv2d(xr, wr, None, (1, 1), "same")
    dy = torch.randn(*yr.shape, generator=g) * sd
    yr.backward(dy.double())
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), "same", (1, 1))
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    y, dx, dw = torch.empty_like(dyd), torch.empty_like(xd), torch.zeros_like(wd)
    floatx.set_floatx(mode)
    Kn.conv2d_fwd(desc, xd, wd, None, y)
    Kn.conv2d_dgrad(desc, dyd, wd, dx)
    Kn.conv2d_wgrad(desc, xd, dyd, dw)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()
    e = (rel_l2(y.cpu(), yr.detach()), rel_l2(dx.cpu(), xr.grad), rel_l2(dw.cpu(), wr.grad))
    assert max(e) <= TOL[mode], (e, sx, sw, sd)


@pytest.mark.parametrize("mode", MODES)
def test_split_modes_refuse_16_bit_tensors(mode, floatx):
    """Modes 3 / 4 are arithmetics of fp32 tensors: 16-bit storage belongs to mode 1 and is refused here."""
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    desc = Kn.make_conv_desc(4, 19, 19, 256, 256, (1, 1), (1, 1), "same", (1, 1))
    x = torch.randn(4, 19, 19, 256, device="cuda").half()
    w = torch.randn(1, 1, 256, 256, device="cuda")
    y = torch.empty(4, 19, 19, 256, device="cuda")
    floatx.set_floatx(mode)
    with pytest.raises(Exception):
        Kn.conv2d_fwd(desc, x, w, None, y)
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("geom", [(4, 19, 19, 256, 192, 3, 1, "same"), (3, 10, 10, 512, 256, 1, 1, "valid"),
                                  (5, 10, 10, 128, 320, 3, 2, "same")])
def test_split_modes_every_tile_and_k_depth(geom, mode, floatx):
    """Every configuration index of the tuner under modes 3 / 4 = every PREC-3 / PREC-4 variant of the 16-bit-tile kernel (tiles, K-step
    depths, prefetch sets, the no-bounds 1x1 variant), in each direction, with the BN prologue / statistics / residual-add
    forms and with split-K, against the fp64 oracle at the mode's bound."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    lib = _lib.load()
    b, h, w, ci, co, k, s, pad = geom
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    sc, sh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.5
    res = torch.randn(b, h, w, ci, generator=g)
    xa = torch.relu(x * sc + sh)
    xr, wr = xa.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = ko.conv2d(xr, wr, None, (s, s), pad)
    dy = torch.randn(*yr.shape, generator=g) * 1e-3
    yr.backward(dy.double())
    w_plain = wt.double().requires_grad_(True)
    ko.conv2d(x.double(), w_plain, None, (s, s), pad).backward(dy.double())
    xs = torch.relu(x * sc + sh + res)
    ys = ko.conv2d(xs.double(), wt.double(), None, (s, s), pad) if (k == 1 and s == 1) else None
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (1, 1))
    rows = Kn.conv2d_stats_rows(desc)
    xd, wd, dyd, scd, shd, resd = [t.cuda() for t in (x, wt, dy, sc, sh, res)]
    floatx.set_floatx(mode)
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for splits in (1, 2):
                tag = "cfg %d splits %d" % (cfg, splits)
                y = torch.empty(yr.shape, device="cuda")
                stats = torch.zeros(rows, 2, co, device="cuda")
                _lib.check(lib.dj_conv2d_tune_set(4, desc, cfg, 1), "tune_set")
                Kn.conv2d_fwd(desc, xd, wd, None, y, scd, shd, True, False, stats)
                _lib.check(lib.dj_conv2d_tune_set(0, desc, cfg, splits), "tune_set")
                y0 = torch.zeros(yr.shape, device="cuda")
                Kn.conv2d_fwd(desc, xd, wd, None, y0, scd, shd, True, False, None, y_zeroed=True)
                dx = torch.empty(x.shape, device="cuda")
                _lib.check(lib.dj_conv2d_tune_set(1, desc, cfg, splits), "tune_set")
                Kn.conv2d_dgrad(desc, dyd, wd, dx)
                _lib.check(lib.dj_conv2d_tune_set(2, desc, cfg, splits), "tune_set")
                dw_pro, dw_plain = torch.zeros(wt.shape, device="cuda"), torch.full(wt.shape, 7.0, device="cuda")
                Kn.conv2d_wgrad(desc, xd, dyd, dw_pro, scd, shd, True, dw_zeroed=True)
                Kn.conv2d_wgrad(desc, xd, dyd, dw_plain)
                outs = [y, y0, dx, dw_pro, dw_plain]
                if ys is not None:
                    y3, sm = torch.empty(yr.shape, device="cuda"), torch.empty(x.shape, device="cuda")
                    Kn.conv2d_fwd_addrelu(desc, xd, wd, None, y3, scd, shd, resd, None, None, sm)
                    outs += [y3, sm]
                torch.cuda.synchronize()
                outs = [o.cpu().double() for o in outs]
                refs = [yr.detach(), yr.detach(), xr.grad, wr.grad, w_plain.grad] + ([ys, xs.double()] if ys is not None else [])
                errs = [rel_l2(o, r) for o, r in zip(outs, refs)]
                assert max(errs) <= TOL[mode], (tag, errs)
                st = stats.cpu().double()
                assert (st[:, 0].sum(0) - outs[0].reshape(-1, co).sum(0)).abs().max() <= 1e-4 * float(yr.detach().abs().max()) * b * h * w, tag
    finally:
        for direction in (0, 4, 1, 2):
            _lib.check(lib.dj_conv2d_tune_set(direction, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("archi", ["deconv", "ssd_custom"])
def test_training_step_split_modes_meet_the_fp32_bar(archi, floatx):
    """SSD300 training step under float32x3 against the fp64 oracle at the bar of the exact-fp32 mode -- predictions 1e-3
    (max-norm), loss 1e-3 -- and, for the weight update (whose gradient at batch 2 is ill-conditioned in ANY fp32
    arithmetic: tests/test_ssd_gpu.py holds the exact-fp32 mode to 1e-2 per well-conditioned tensor), 5e-3 rel-L2 over all
    tensors together; the exact-fp32 mode's own figure is printed beside it (measured: 3.6e-4 against 1.6e-3)."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from oracle import ssd_resnet_dct as oracle
    floatx.clear_session()
    model, sizes = workloads.build_ssd(archi)
    x, y_true = workloads.synthetic_batch(archi, sizes, 2)
    w0 = model.get_weights_dict()
    wt = {k: torch.from_numpy(v).double() for k, v in w0.items()}
    ref = oracle.ssd_training_step(wt, [torch.from_numpy(a).double() for a in x], torch.from_numpy(y_true).double(), archi,
                                   lr=0.001, momentum=0.9)
    den = sum(float(((ref["new_weights"][k] - wt[k]) ** 2).sum()) for k in w0 if k in ref["new_weights"])
    got = {}
    for mode in ("float32x3", "float32x6", "float32_mfma"):
        if mode != "float32x3":
            floatx.clear_session()              # (the same auto-generated layer names as the first build)
        m = model if mode == "float32x3" else workloads.build_ssd(archi)[0]
        assert m.set_weights_dict(w0, strict=True) == len(w0)
        floatx.set_floatx(mode)
        loss = m.train_on_batch(x, y_true)
        torch.cuda.synchronize()
        plan = m._plan(2, True, True)
        assert plan.compute_mode == {"float32x3": 3, "float32x6": 4, "float32_mfma": 5}[mode] and not plan.store16
        y_pred = plan.outputs[0].buf.cpu().double()
        floatx.set_floatx("float32")
        e_pred = float((y_pred - ref["y_pred"]).abs().max()) / float(ref["y_pred"].abs().max())
        e_loss = abs(loss - ref["loss"]) / abs(ref["loss"])
        w1 = m.get_weights_dict()
        num = sum(float(((torch.from_numpy(w1[k]).double() - ref["new_weights"][k]) ** 2).sum()) for k in w0 if k in ref["new_weights"])
        got[mode] = (e_pred, e_loss, (num / den) ** 0.5)
        print("%s %s: pred max-norm %.2e, loss %.2e, update rel-L2 %.2e" % (mode, archi, e_pred, e_loss, got[mode][2]))
    for mode in got:
        e_pred, e_loss, e_upd = got[mode]
        assert e_pred <= 1e-3 and e_loss <= 1e-3 and e_upd <= 5e-3, got
    # float32x6 is an fp32 arithmetic: as close to the oracle as the fp32 MFMA kernels' step, within 3x (both are a few ulps
    # of fp32 per GEMM, amplified alike by the batch-of-2 BatchNormalization)
    assert all(a <= 3.0 * b + 1e-6 for a, b in zip(got["float32x6"], got["float32_mfma"])), got

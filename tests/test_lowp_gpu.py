"""Reduced-precision MFMA modes (BASELINE config 5: 'fp16 MFMA', fp32 master tensors + fp32 accumulation).
Tolerances are those of the operand rounding -- fp16: 2^-11, bf16: 2^-8 per operand, random-sign accumulation -- and
are stated per assertion; the exact-fp32 mode stays the one every 1e-3 claim refers to."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def floatx():
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    yield K
    K.set_floatx("float32")


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("mode,tol_fwd,tol_bwd", [("float16", 1.5e-3, 8e-3), ("bfloat16", 8e-3, 8e-3)])
@pytest.mark.parametrize("geom", [(8, 19, 19, 256, 256, 3, 1), (4, 38, 38, 128, 512, 1, 1), (8, 10, 10, 512, 512, 3, 1)])
def test_conv_directions_in_reduced_precision(mode, tol_fwd, tol_bwd, geom, floatx):
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    b, h, w, ci, co, k, s = geom
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    bias = torch.randn(co, generator=g)
    dy = torch.randn(b, h, w, co, generator=g) * 1e-4          # gradients are small: needs the bf16 exponent range
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = ko.conv2d(xr, wr, bias.double(), (s, s), "same")
    yr.backward(dy.double())
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), "same", (1, 1))
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    y, dx, dw = torch.empty_like(dyd), torch.empty_like(xd), torch.zeros_like(wd)
    floatx.set_floatx(mode)
    assert floatx.floatx() == mode
    Kn.conv2d_fwd(desc, xd, wd, bias.cuda(), y)
    Kn.conv2d_dgrad(desc, dyd, wd, dx)
    Kn.conv2d_wgrad(desc, xd, dyd, dw)
    torch.cuda.synchronize()
    e = (rel_l2(y.cpu(), yr.detach()), rel_l2(dx.cpu(), xr.grad), rel_l2(dw.cpu(), wr.grad))
    assert e[0] <= tol_fwd and e[1] <= tol_bwd and e[2] <= tol_bwd, e
    assert e[0] > 1e-5                                            # the reduced-precision kernels did run
    floatx.set_floatx("float32")
    Kn.conv2d_fwd(desc, xd, wd, bias.cuda(), y)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), yr.detach()) <= 2e-6


@pytest.mark.parametrize("archi", ["ssd_custom", "up_sampling"])
def test_config5_training_step_fp16_mfma(archi, floatx):
    """late_concat_rfa_thinner backbone (= ssd_custom) and the up_sampling(_rfa) fusion archi, SSD300 training step with
    fp16/bf16 MFMA convolutions against the fp64 oracle: predictions 1e-2 rel-L2 and 6e-2 max-norm
    (operand rounding through ~70 conv + BN layers of a random-init net), loss 1e-2."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from oracle import ssd_resnet_dct as oracle
    model, sizes = workloads.build_ssd(archi)
    x, y_true = workloads.synthetic_batch(archi, sizes, 2)
    w0 = model.get_weights_dict()
    floatx.set_floatx("float16")
    loss = model.train_on_batch(x, y_true)
    torch.cuda.synchronize()
    y_pred = model._plan(2, True, True).outputs[0].buf.cpu().double()
    floatx.set_floatx("float32")
    wt = {k: torch.from_numpy(v).double() for k, v in w0.items()}
    ref = oracle.ssd_training_step(wt, [torch.from_numpy(a).double() for a in x], torch.from_numpy(y_true).double(), archi,
                                   lr=0.001, momentum=0.9)
    e_pred = float((y_pred - ref["y_pred"]).abs().max()) / float(ref["y_pred"].abs().max())
    e_l2 = rel_l2(y_pred[..., :25], ref["y_pred"][..., :25])
    e_loss = abs(loss - ref["loss"]) / abs(ref["loss"])
    print("fp16-MFMA %s: max-norm %.2e, rel-L2 %.2e, loss %.2e" % (archi, e_pred, e_l2, e_loss))
    assert e_pred <= 6e-2 and e_l2 <= 1e-2 and e_loss <= 1e-2, (e_pred, e_l2, e_loss)
    w1 = model.get_weights_dict()
    moved = max(float(np.abs(w1[k] - w0[k]).max()) for k in w0)
    assert np.isfinite(loss) and moved > 0


@pytest.mark.parametrize("geom", [(4, 19, 19, 256, 192, 3, 1, "same"), (3, 10, 10, 512, 256, 1, 1, "valid"),
                                  (2, 38, 38, 64, 128, 3, 1, "same"), (5, 10, 10, 128, 320, 3, 2, "same")])
def test_h16_every_tile_and_k_depth(geom, floatx):
    """Every configuration index of the tuner in the reduced-precision mode = every variant of the 16-bit-tile kernel
    (dj_igemm_h16.h: tiles 128x128 / 128x64 / 64x64, K-steps of 32 and 64, one and two register prefetch sets; the map is
    in dj_conv_launch.h): forward with BN prologue + statistics, residual-add prologue, input gradient, and the weight
    gradient (both operands through the transposing LDS read; with and without the BN prologue), each with and without
    split-K -- against the fp64 oracle at the operand-rounding tolerance, and against the first configuration at
    fp32-accumulation-order tolerance (all variants round the same operands)."""
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
    w_plain = wt.double().requires_grad_(True)          # weight gradient without the prologue: x as it is
    ko.conv2d(x.double(), w_plain, None, (s, s), pad).backward(dy.double())
    xs = torch.relu(x * sc + sh + res)
    ys = ko.conv2d(xs.double(), wt.double(), None, (s, s), pad) if (k == 1 and s == 1) else None
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (1, 1))
    rows = Kn.conv2d_stats_rows(desc)
    xd, wd, dyd, scd, shd, resd = [t.cuda() for t in (x, wt, dy, sc, sh, res)]
    floatx.set_floatx("float16")
    first = None
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
                Kn.conv2d_wgrad(desc, xd, dyd, dw_plain)          # clears dw itself when it splits
                outs = [y, y0, dx, dw_pro, dw_plain]
                if ys is not None:
                    y3, sm = torch.empty(yr.shape, device="cuda"), torch.empty(x.shape, device="cuda")
                    Kn.conv2d_fwd_addrelu(desc, xd, wd, None, y3, scd, shd, resd, None, None, sm)
                    outs += [y3, sm]
                torch.cuda.synchronize()
                outs = [o.cpu().double() for o in outs]
                assert rel_l2(outs[0], yr.detach()) <= 1.5e-3 and rel_l2(outs[1], yr.detach()) <= 1.5e-3, tag
                assert rel_l2(outs[2], xr.grad) <= 8e-3, tag
                assert rel_l2(outs[3], wr.grad) <= 8e-3 and rel_l2(outs[4], w_plain.grad) <= 8e-3, tag
                st = stats.cpu().double()
                assert (st[:, 0].sum(0) - outs[0].reshape(-1, co).sum(0)).abs().max() <= 1e-3 * float(yr.detach().abs().max()) * b * h * w, tag
                if ys is not None:
                    assert rel_l2(outs[5], ys) <= 1.5e-3 and rel_l2(outs[6], xs.double()) <= 1e-6, tag
                if first is None:
                    first = outs
                for i, (a, r) in enumerate(zip(outs, first)):
                    # (weight gradients: ~1000-term fp32 sums whose grouping changes with the tile and the split)
                    assert rel_l2(a, r) <= (1e-5 if i in (3, 4) else 2e-6), (tag, i)
    finally:
        for direction in (0, 4, 1, 2):
            _lib.check(lib.dj_conv2d_tune_set(direction, desc, -1, 1), "tune_set")

"""GPU parity of the implicit-GEMM convolution (fwd / dgrad / wgrad) against the CPU oracle.
Tolerance: 1e-3 relative (north_star), checked as max|diff| <= 1e-3 * max|ref| + 1e-5."""
import numpy as np
import pytest
import torch

from oracle import keras_ops as ko

pytestmark = pytest.mark.gpu

# (batch, H, W, Cin, Cout, k, stride, padding, dilation)  -- shapes taken from the SSD300 graphs
CASES = [
    (2, 38, 38, 64, 256, 1, 1, "valid", 1),     # res1a2_branch2a
    (2, 38, 38, 256, 256, 2, 1, "same", 1),     # identity_block(..., 2, ...): asymmetric same padding
    (2, 38, 38, 128, 128, 3, 1, "same", 1),     # res2b3_branch2b
    (2, 38, 38, 384, 256, 1, 2, "valid", 1),    # res2a4_branch2a stride 2 -> 19
    (2, 19, 19, 512, 1024, 1, 2, "valid", 1),   # res4a_branch1 -> 10
    (4, 10, 10, 256, 256, 3, 1, "same", 1),     # res4b_branch2b
    (4, 5, 5, 2048, 1024, 3, 1, "same", 6),     # fc6 dilation 6 (split-K path)
    (4, 10, 10, 2048, 1024, 3, 1, "same", 6),   # fc6 in the `identical` archis
    (4, 5, 5, 256, 256, 3, 2, ((1, 1), (1, 1)), 1),  # conv6_2: ZeroPadding2D(1) + 3x3 s2 valid
    (4, 3, 3, 128, 256, 3, 1, "valid", 1),      # conv9_2 -> 1x1
    (2, 38, 38, 384, 84, 3, 1, "same", 1),      # conv4_3_norm_mbox_conf (N=84)
    (2, 19, 19, 512, 126, 3, 1, "same", 1),     # fc7_mbox_conf (N=126: not a multiple of 4)
    (2, 38, 38, 64, 16, 3, 1, "same", 1),       # conv4_3_norm_mbox_loc on raw Y (identical archis)
    (3, 1, 1, 256, 24, 3, 1, "same", 1),        # conv9_2_mbox_loc on a 1x1 map
    (2, 12, 12, 4, 32, 7, 2, ((3, 3), (3, 3)), 1),  # stem-like 7x7 s2 with Cin=4
    (2, 9, 9, 3, 8, 3, 1, "same", 1),           # scalar-gather fallback (Cin=3)
]


def _tol(ref):
    return 1e-3 * float(ref.abs().max()) + 1e-5


def _geometry(case):
    b, h, w, ci, co, k, s, pad, d = case
    return b, h, w, ci, co, (k, k), (s, s), pad, (d, d)


def _oracle_conv(x, wt, bias, case):
    b, h, w, ci, co, k, s, pad, d = case
    if isinstance(pad, tuple):
        x = ko.zero_padding(x, pad)
        pad = "valid"
    return ko.conv2d(x, wt, bias, (s, s), pad, (d, d))


@pytest.fixture(params=["fast", "generic"])
def path(request):
    """Every case runs through the launcher's default choice (branch-free kernel where its preconditions
    hold) and again with the generic kernel forced."""
    from jpeg_detection_resnet_ssd_amd import _lib
    _lib.load().dj_set_fast_path(1 if request.param == "fast" else 0)
    yield request.param
    _lib.load().dj_set_fast_path(1)


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, cuda, path):
    from jpeg_detection_resnet_ssd_amd import kernels as K
    b, h, w, ci, co, kk, ss, pad, dd = _geometry(case)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(kk[0], kk[1], ci, co, generator=g) * (2.0 / (kk[0] * kk[1] * ci)) ** 0.5
    bias = torch.randn(co, generator=g)
    xr = x.double().requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    br = bias.double().requires_grad_(True)
    yr = _oracle_conv(xr, wr, br, case)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    desc = K.make_conv_desc(b, h, w, ci, co, kk, ss, pad, dd)
    assert (desc.out_h, desc.out_w) == tuple(yr.shape[1:3])
    xd, wd, bd, dyd = x.to(cuda), wt.to(cuda), bias.to(cuda), dy.to(cuda)
    y = torch.empty(yr.shape, device=cuda)
    K.conv2d_fwd(desc, xd, wd, bd, y)
    torch.cuda.synchronize()
    assert (y.cpu().double() - yr.detach()).abs().max() <= _tol(yr.detach())

    dx = torch.full(x.shape, float("nan"), device=cuda)
    K.conv2d_dgrad(desc, dyd, wd, dx)
    torch.cuda.synchronize()
    assert (dx.cpu().double() - xr.grad).abs().max() <= _tol(xr.grad)

    # beta = 1 accumulates
    dx2 = torch.ones(x.shape, device=cuda)
    K.conv2d_dgrad(desc, dyd, wd, dx2, beta=True)
    torch.cuda.synchronize()
    assert (dx2.cpu().double() - 1.0 - xr.grad).abs().max() <= _tol(xr.grad)

    dw = torch.full(wt.shape, float("nan"), device=cuda)
    K.conv2d_wgrad(desc, xd, dyd, dw)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - wr.grad).abs().max() <= _tol(wr.grad)


def test_conv_prologue_relu_stats_and_slices(cuda, path):
    """BN+ReLU folded into the A load, ReLU epilogue, per-tile BN statistics, and
    channel-slice views (ld > channels) on both sides."""
    from jpeg_detection_resnet_ssd_amd import kernels as K
    g = torch.Generator().manual_seed(7)
    b, h, w, ci, co = 3, 19, 19, 128, 192
    xbuf = torch.randn(b, h, w, ci + 64, generator=g)
    x = xbuf[..., 32:32 + ci]
    wt = torch.randn(3, 3, ci, co, generator=g) * 0.05
    bias = torch.randn(co, generator=g)
    sc = torch.rand(ci, generator=g) + 0.5
    sh = torch.randn(ci, generator=g) * 0.3
    a = ko.relu(x.double() * sc.double() + sh.double())
    raw = ko.conv2d(a, wt.double(), None, (1, 1), "same", (1, 1))
    yr = ko.relu(raw + bias.double())

    desc = K.make_conv_desc(b, h, w, ci, co, (3, 3), (1, 1), "same", (1, 1))
    xb = xbuf.to(cuda)
    ybuf = torch.zeros(b, h, w, co + 8, device=cuda)
    rows = K.conv2d_stats_rows(desc)
    stats = torch.zeros(rows, 2, co, device=cuda)
    K.conv2d_fwd(desc, xb[..., 32:32 + ci], wt.to(cuda), bias.to(cuda), ybuf[..., 4:4 + co],
                 pro_scale=sc.to(cuda), pro_shift=sh.to(cuda), pro_relu=True, relu=True, stats=stats)
    torch.cuda.synchronize()
    y = ybuf[..., 4:4 + co].cpu().double()
    assert (y - yr).abs().max() <= _tol(yr)
    assert float(ybuf[..., :4].abs().max()) == 0.0 and float(ybuf[..., 4 + co:].abs().max()) == 0.0
    st = stats.cpu().double().sum(dim=0)
    ref_s = raw.sum(dim=(0, 1, 2))
    ref_q = (raw * raw).sum(dim=(0, 1, 2))
    assert (st[0] - ref_s).abs().max() <= 1e-3 * ref_s.abs().max() + 1e-3
    assert (st[1] - ref_q).abs().max() <= 1e-3 * ref_q.abs().max()

    # wgrad sees the same folded activation
    dy = torch.randn(b, h, w, co, generator=g)
    ar = a.clone().requires_grad_(False)
    wr = wt.double().requires_grad_(True)
    ko.conv2d(ar, wr, None, (1, 1), "same", (1, 1)).backward(dy.double())
    dw = torch.empty_like(wt, device=cuda)
    K.conv2d_wgrad(desc, xb[..., 32:32 + ci], dy.to(cuda), dw, pro_scale=sc.to(cuda), pro_shift=sh.to(cuda),
                   pro_relu=True)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - wr.grad).abs().max() <= _tol(wr.grad)


def test_conv_transpose_via_dgrad(cuda):
    """Conv2DTranspose(64, 2, strides 2) forward == dgrad of the k2s2 conv with the same kernel."""
    from jpeg_detection_resnet_ssd_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    b, h, w, ci, co = 2, 19, 19, 64, 64
    x = torch.randn(b, h, w, ci, generator=g)
    kern = torch.randn(2, 2, co, ci, generator=g) * 0.1   # (kh, kw, out, in)
    bias = torch.randn(co, generator=g)
    yr = ko.conv2d_transpose(x.double(), kern.double(), bias.double(), (2, 2))
    # the transposed layer's output plays the role of the conv's input (38x38xco), its input the conv's output
    desc = K.make_conv_desc(b, 2 * h, 2 * w, co, ci, (2, 2), (2, 2), "valid", (1, 1))
    ybuf = torch.zeros(b, 2 * h, 2 * w, 192, device=cuda)
    K.conv2d_dgrad(desc, x.to(cuda), kern.to(cuda), ybuf[..., 64:128], bias=bias.to(cuda))
    torch.cuda.synchronize()
    assert (ybuf[..., 64:128].cpu().double() - yr).abs().max() <= _tol(yr)


VARIANT_CASES = [
    (3, 19, 19, 96, 128, 3, 1, "same", 1),      # K = 27 K-steps (odd: the K groups of the *_PK2 variants get 14 / 13)
    (2, 10, 10, 64, 192, 1, 1, "valid", 1),     # K = 2 K-steps, N = 1.5 tiles of 128
    (2, 38, 38, 32, 64, 2, 1, "same", 1),       # one K-step per tap (srcC = 32), asymmetric padding
    (3, 19, 19, 128, 96, 3, 1, "same", 1),      # pixel-walk wgrad, one tap per 128-row tile, odd pixel count (1083)
    (3, 19, 19, 256, 40, 3, 1, "same", 2),      # ... dilation 2, N not a multiple of 32
    (5, 7, 9, 128, 64, 1, 1, "valid", 1),       # 1x1 unpadded (the NP kernels: no bounds arithmetic), 315 pixels
    (2, 11, 13, 128, 64, 3, 2, "same", 1),      # pixel-walk wgrad (in_c % BM == 0): stride 2, output grid != input grid
    (7, 3, 3, 128, 64, 3, 1, "same", 1),        # ... a map smaller than a K-step (9 pixels): image carries every step
    (3, 9, 5, 256, 32, 3, 2, "valid", 1),       # ... stride 2 without padding, 4x2 outputs per image
]


@pytest.mark.parametrize("case", VARIANT_CASES)
def test_every_tile_variant(case, cuda):
    """Each tile / schedule variant the tuner may register (dj_conv2d_tune_set), in the three directions, with the BN
    prologue and the statistics epilogue, with and without split-K."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as K
    lib = _lib.load()
    b, h, w, ci, co, kk, ss, pad, dd = _geometry(case)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(kk[0], kk[1], ci, co, generator=g) * (2.0 / (kk[0] * kk[1] * ci)) ** 0.5
    bias = torch.randn(co, generator=g)
    sc, sh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g)
    xa = torch.relu(x * sc + sh)                                   # what the prologue feeds the GEMM
    xr, wr = xa.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = _oracle_conv(xr, wr, bias.double(), case)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    y_nobias = (yr.detach() - bias.double()).reshape(-1, co)
    desc = K.make_conv_desc(b, h, w, ci, co, kk, ss, pad, dd)
    xd, wd, bd, dyd, scd, shd = (t.to(cuda) for t in (x, wt, bias, dy, sc, sh))
    rows = K.conv2d_stats_rows(desc)
    for cfg in range(lib.dj_conv2d_tune_configs()):
        for splits in (1, 3):
            # forward with prologue + statistics (no split-K there), dgrad of the same geometry, wgrad with prologue
            y = torch.empty(yr.shape, device=cuda)
            stats = torch.zeros(rows, 2, co, device=cuda)
            _lib.check(lib.dj_conv2d_tune_set(4, desc, cfg, 1), "tune_set")
            K.conv2d_fwd(desc, xd, wd, bd, y, scd, shd, True, False, stats)
            dx = torch.empty(x.shape, device=cuda)
            _lib.check(lib.dj_conv2d_tune_set(1, desc, cfg, splits), "tune_set")
            K.conv2d_dgrad(desc, dyd, wd, dx)
            dw = torch.zeros(wt.shape, device=cuda)
            _lib.check(lib.dj_conv2d_tune_set(2, desc, cfg, splits), "tune_set")
            K.conv2d_wgrad(desc, xd, dyd, dw, scd, shd, True)
            torch.cuda.synchronize()
            tag = "cfg %d splits %d" % (cfg, splits)
            assert (y.cpu().double() - yr.detach()).abs().max() <= _tol(yr.detach()), tag
            s = stats.cpu().double()
            assert (s[:, 0].sum(0) - y_nobias.sum(0)).abs().max() <= 1e-3 * float(y_nobias.sum(0).abs().max()) + 1e-3, tag
            assert (s[:, 1].sum(0) - (y_nobias ** 2).sum(0)).abs().max() <= 1e-3 * float((y_nobias ** 2).sum(0).max()), tag
            assert (dx.cpu().double() - xr.grad).abs().max() <= _tol(xr.grad), tag
            assert (dw.cpu().double() - wr.grad).abs().max() <= _tol(wr.grad), tag
    for direction in (4, 1, 2):
        _lib.check(lib.dj_conv2d_tune_set(direction, desc, -1, 1), "tune_set")


SLAB_CASES = [
    (3, 10, 10, 256, 96, 3, 1, "same", 1),      # 300 rows (partial row tile and partial 64-row statistics group), N = 96
    (2, 5, 5, 512, 126, 1, 1, "valid", 1),      # N = 126: scalar reduction kernel, generic GEMM
    (4, 3, 3, 128, 160, 3, 1, "same", 1),       # 36 rows, 1.25 column tiles
]


@pytest.mark.parametrize("case", SLAB_CASES)
def test_split_forward_through_slabs_is_exact_and_reproducible(case, cuda):
    """dj_conv2d_nhwc_fwd_ws: a split-K forward launch stores its partial tiles to the caller's workspace and a second
    kernel adds them in a fixed order (+ bias, ReLU, BatchNormalization statistics).  Every tile variant, splits 1 / 2 /
    5: result against the fp64 oracle, statistics against the result, and two launches bit-identical (the path without
    workspace accumulates with atomics and is not).  Too small a workspace: atomics again, or -- when the statistics
    must come from the reduction -- an unsplit launch; still correct."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as K
    lib = _lib.load()
    b, h, w, ci, co, kk, ss, pad, dd = _geometry(case)
    g = torch.Generator().manual_seed(91)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(kk[0], kk[1], ci, co, generator=g) * (2.0 / (kk[0] * kk[1] * ci)) ** 0.5
    bias = torch.randn(co, generator=g)
    sc, sh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g)
    xa = torch.relu(x * sc + sh).double()
    y_lin = _oracle_conv(xa, wt.double(), bias.double(), case)
    y_nobias = (y_lin - bias.double()).reshape(-1, co)
    desc = K.make_conv_desc(b, h, w, ci, co, kk, ss, pad, dd)
    xd, wd, bd, scd, shd = (t.to(cuda) for t in (x, wt, bias, sc, sh))
    rows = K.conv2d_stats_rows(desc)
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for splits in (1, 2, 5):
                _lib.check(lib.dj_conv2d_tune_set(0, desc, cfg, splits), "tune_set")
                need = K.conv2d_fwd_workspace_floats(desc, True)
                tag = "cfg %d splits %d" % (cfg, splits)
                assert (need > 0) == (splits > 1) and need % (y_nobias.numel()) == 0, tag
                ws = torch.full((max(need, 4),), float("nan"), device=cuda)
                for relu in (False, True):
                    yr = torch.relu(y_lin) if relu else y_lin
                    outs = []
                    for _ in range(2):
                        y = torch.full(y_lin.shape, float("nan"), device=cuda)
                        stats = torch.full((rows, 2, co), float("nan"), device=cuda)
                        K.conv2d_fwd(desc, xd, wd, bd, y, scd, shd, True, relu, stats, workspace=ws, stats_may_split=True)
                        outs.append((y, stats))
                    torch.cuda.synchronize()
                    (y, stats), (y2, stats2) = outs
                    assert torch.equal(y, y2) and torch.equal(stats, stats2), tag
                    assert (y.cpu().double() - yr).abs().max() <= _tol(yr), tag
                    s = stats.cpu().double()
                    assert (s[:, 0].sum(0) - y_nobias.sum(0)).abs().max() <= 1e-3 * float(y_nobias.sum(0).abs().max()) + 1e-3, tag
                    assert (s[:, 1].sum(0) - (y_nobias ** 2).sum(0)).abs().max() <= 1e-3 * float((y_nobias ** 2).sum(0).max()), tag
                # without statistics, and with a workspace that is too small (atomic fallback / unsplit launch)
                y = torch.full(y_lin.shape, float("nan"), device=cuda)
                K.conv2d_fwd(desc, xd, wd, bd, y, scd, shd, True, False, None, workspace=ws)
                small = torch.empty(8, device=cuda)
                y3 = torch.full(y_lin.shape, float("nan"), device=cuda)
                K.conv2d_fwd(desc, xd, wd, bd, y3, scd, shd, True, False, None, workspace=small)
                y4 = torch.full(y_lin.shape, float("nan"), device=cuda)
                stats4 = torch.full((rows, 2, co), float("nan"), device=cuda)
                K.conv2d_fwd(desc, xd, wd, bd, y4, scd, shd, True, False, stats4, workspace=small, stats_may_split=True)
                torch.cuda.synchronize()
                for t in (y, y3, y4):
                    assert (t.cpu().double() - y_lin).abs().max() <= _tol(y_lin), tag
                assert (stats4.cpu().double()[:, 0].sum(0) - y_nobias.sum(0)).abs().max() <= \
                    1e-3 * float(y_nobias.sum(0).abs().max()) + 1e-3, tag
    finally:
        _lib.check(lib.dj_conv2d_tune_set(0, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("geom", [(3, 19, 19, 256, 64), (2, 38, 38, 64, 256), (2, 10, 10, 512, 128), (1, 5, 5, 96, 32)])
@pytest.mark.parametrize("res_affine", [False, True])
def test_fwd_with_residual_add_prologue(geom, res_affine, cuda):
    """dj_conv2d_nhwc_fwd_addrelu: conv1x1(relu(bn(z) + shortcut)) in one launch, the sum written as a by-product,
    BN statistics of the conv output in the epilogue -- every tile variant."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as K
    lib = _lib.load()
    b, h, w, ci, co = geom
    g = torch.Generator().manual_seed(5)
    z, r = torch.randn(b, h, w, ci, generator=g), torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(1, 1, ci, co, generator=g) * (2.0 / ci) ** 0.5
    bias = torch.randn(co, generator=g)
    sc, sh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g)
    rs, rt = (torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g)) if res_affine else (None, None)
    a = torch.relu(z.double() * sc.double() + sh.double() + (r.double() * rs.double() + rt.double() if res_affine else r.double()))
    yr = ko.conv2d(a, wt.double(), bias.double(), (1, 1), "valid")
    y_nobias = (yr - bias.double()).reshape(-1, co)
    desc = K.make_conv_desc(b, h, w, ci, co, (1, 1), (1, 1), "valid", (1, 1))
    assert K.conv2d_fwd_addrelu_supported(desc)
    dev = [t.to(cuda) if t is not None else None for t in (z, r, wt, bias, sc, sh, rs, rt)]
    zd, rd, wd, bd, scd, shd, rsd, rtd = dev
    rows = K.conv2d_stats_rows(desc)
    for cfg in range(lib.dj_conv2d_tune_configs()):
        _lib.check(lib.dj_conv2d_tune_set(4, desc, cfg, 1), "tune_set")
        y = torch.empty(b, h, w, co, device=cuda)
        s_out = torch.full((b, h, w, ci), float("nan"), device=cuda)
        stats = torch.zeros(rows, 2, co, device=cuda)
        K.conv2d_fwd_addrelu(desc, zd, wd, bd, y, scd, shd, rd, rsd, rtd, s_out, False, stats)
        torch.cuda.synchronize()
        assert (s_out.cpu().double() - a).abs().max() <= 1e-5 * float(a.abs().max()) + 1e-6, cfg
        assert (y.cpu().double() - yr).abs().max() <= _tol(yr), cfg
        st = stats.cpu().double()
        assert (st[:, 0].sum(0) - y_nobias.sum(0)).abs().max() <= 1e-3 * float(y_nobias.sum(0).abs().max()) + 1e-3, cfg
    _lib.check(lib.dj_conv2d_tune_set(4, desc, -1, 1), "tune_set")
    bad = K.make_conv_desc(b, h, w, ci, co, (3, 3), (1, 1), "same", (1, 1))
    assert not K.conv2d_fwd_addrelu_supported(bad)


@pytest.mark.parametrize("geom", [(3, 19, 19, 96, 128, 3), (2, 38, 38, 64, 256, 1), (4, 10, 10, 256, 64, 1),
                                  (8, 75, 75, 64, 64, 1)])   # the last one spreads over 5 accumulator replicas
def test_fwd_with_fused_batchnorm_finalize(geom, cuda):
    """dj_conv2d_nhwc_fwd_bn: scale / shift / saved and moving statistics written by the conv launch itself (fp64 atomics
    + last-workgroup ticket) == conv statistics + dj_bn_train_finalize, for every tile variant, twice in a row (the
    accumulators and the ticket must be left clean)."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as K
    from jpeg_detection_resnet_ssd_amd.engine import call
    lib = _lib.load()
    b, h, w, ci, co, k = geom
    g = torch.Generator().manual_seed(8)
    x = torch.randn(b, h, w, ci, generator=g).to(cuda)
    wt = (torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5).to(cuda)
    bias = torch.randn(co, generator=g).to(cuda)
    gamma, beta = (torch.rand(co, generator=g) + 0.5).to(cuda), torch.randn(co, generator=g).to(cuda)
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), "same", (1, 1))
    rows = b * h * w
    # reference: the two-launch path
    y0 = torch.empty(b, h, w, co, device=cuda)
    nr = K.conv2d_stats_rows(desc)
    stats = torch.zeros(nr, 2, co, device=cuda)
    mm0, mv0 = torch.zeros(co, device=cuda), torch.ones(co, device=cuda)
    ref = [torch.empty(co, device=cuda) for _ in range(4)]
    K.conv2d_fwd(desc, x, wt, bias, y0, None, None, False, False, stats)
    call("dj_bn_train_finalize", stats, nr, rows, bias, gamma, beta, 1e-3, 0.99, mm0, mv0, *ref, co)
    acc = torch.zeros(K.BN_ACC_REPLICAS * 2 * co, dtype=torch.float64, device=cuda)
    ticket = torch.zeros(1, dtype=torch.int32, device=cuda)
    for cfg in range(lib.dj_conv2d_tune_configs()):
        _lib.check(lib.dj_conv2d_tune_set(4, desc, cfg, 1), "tune_set")
        mm, mv = torch.zeros(co, device=cuda), torch.ones(co, device=cuda)
        out = [torch.full((co,), float("nan"), device=cuda) for _ in range(4)]
        bn = K.make_bn_train(acc, ticket, gamma, beta, mm, mv, out[0], out[1], out[2], out[3], 1e-3, 0.99)
        y = torch.empty(b, h, w, co, device=cuda)
        for rep in range(2):
            K.conv2d_fwd_bn(desc, x, wt, bias, y, bn)
        torch.cuda.synchronize()
        assert (y - y0).abs().max() <= 1e-5 * y0.abs().max(), cfg   # other tile variant: other summation order
        for got, want, name in zip(out, ref, ("scale", "shift", "mean", "invstd")):
            assert (got - want).abs().max() <= 2e-6 * want.abs().max() + 1e-7, (cfg, name)
        # two launches = two momentum updates of the moving statistics
        mean, var_u = ref[2], None
        assert (mm - (0.0 * 0.99 + ref[2] * 0.01) * (1 + 0.99)).abs().max() <= 1e-5 * ref[2].abs().max() + 1e-7, cfg
        assert int(ticket.item()) == 0 and float(acc.abs().max()) == 0.0, cfg
    _lib.check(lib.dj_conv2d_tune_set(4, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("geom", [(3, 19, 19, 192, 256, 3, "same"), (2, 38, 38, 128, 512, 1, "valid"),
                                  (5, 10, 10, 320, 128, 3, "same"), (1, 5, 5, 64, 96, 1, "valid")])
@pytest.mark.parametrize("masked", [True, False])
def test_dgrad_takes_bn_backward_statistics(geom, masked, cuda):
    """dj_conv2d_nhwc_dgrad_bnbwd: the input gradient g of a convolution whose input is relu(bn(z)) (or bn(z)) and, from the
    same launch, the BatchNormalization backward sums over g and z -- every tile variant; dx identical to the plain
    launch, column totals of the partial rows against fp64 and against dj_bn_bwd_reduce on the stored dx."""
    from jpeg_detection_resnet_ssd_amd import _lib, engine
    from jpeg_detection_resnet_ssd_amd import kernels as K
    lib = _lib.load()
    b, h, w, ci, co, k, pad = geom
    g = torch.Generator().manual_seed(11)
    z = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), pad, (1, 1))
    dy = torch.randn(b, desc.out_h, desc.out_w, co, generator=g)
    mean, invstd = torch.randn(ci, generator=g) * 0.3, torch.rand(ci, generator=g) + 0.5
    sc, sh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.5
    zd, wd, dyd, md, isd, scd, shd = [t.to(cuda) for t in (z, wt, dy, mean, invstd, sc, sh)]
    rows = b * h * w
    dx_ref = torch.empty(b, h, w, ci, device=cuda)
    _lib.check(lib.dj_conv2d_tune_set(1, desc, 0, 1), "tune_set")
    K.conv2d_dgrad(desc, dyd, wd, dx_ref)
    gm = dx_ref.cpu().double().reshape(rows, ci)
    z2 = z.double().reshape(rows, ci)
    if masked:
        gm = torch.where(z2 * sc.double() + sh.double() > 0, gm, torch.zeros_like(gm))
    want0, want1 = gm.sum(0), (gm * (z2 - mean.double()) * invstd.double()).sum(0)
    nr2 = engine.query("dj_reduce_rows", rows)
    part2 = torch.empty(nr2, 2, ci, device=cuda)
    engine.call("dj_bn_bwd_reduce", dx_ref, ci, zd, ci, None, 0, md, isd, scd, shd, 2 if masked else 0, rows, ci, part2)
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            _lib.check(lib.dj_conv2d_tune_set(1, desc, cfg, 2), "tune_set")     # a registered split is overridden
            dx = torch.full((b, h, w, ci), float("nan"), device=cuda)
            part = torch.full(((rows + 63) // 64, 2, ci), float("nan"), device=cuda)
            K.conv2d_dgrad_bnbwd(desc, dyd, wd, dx, zd, md, isd, scd if masked else None, shd if masked else None, part)
            torch.cuda.synchronize()
            tag = "cfg %d" % cfg
            assert (dx.cpu().double() - dx_ref.cpu().double()).abs().max() <= _tol(dx_ref.cpu().double()), tag
            got = part.cpu().double()
            assert torch.isfinite(got).all(), tag
            for slot, want in ((0, want0), (1, want1)):
                err = (got[:, slot].sum(0) - want).abs().max()
                assert err <= 1e-4 * float(want.abs().max()) + 1e-3, (tag, slot, float(err))
                ref2 = part2.cpu().double()[:, slot].sum(0)
                assert (got[:, slot].sum(0) - ref2).abs().max() <= 1e-4 * float(want.abs().max()) + 1e-3, (tag, slot)
    finally:
        _lib.check(lib.dj_conv2d_tune_set(1, desc, -1, 1), "tune_set")

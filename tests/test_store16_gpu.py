"""BASELINE config 5 proper: activations held as fp16 and their gradients as bf16 in HBM (K.set_floatx('float16'),
include/dj_hip.h `_t` entry points).  Every check feeds the kernel the ROUNDED tensors and compares with the fp64 oracle
on those same rounded values, so what is measured is the kernel's own error: operand rounding inside the GEMM (fp16
forward 1.5e-3, bf16 gradients 8e-3 rel-L2, the tolerances of tests/test_lowp_gpu.py) plus ONE storage rounding where a
16-bit tensor is written (fp16 2^-11 = 4.9e-4, bf16 2^-8 = 3.9e-3 per element, so <= 5e-4 / 4e-3 rel-L2)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32
TOL_FWD, TOL_BWD = 1.5e-3 + 5e-4, 8e-3 + 4e-3


@pytest.fixture()
def floatx():
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    K.set_floatx("float16")
    yield K
    K.set_floatx("float32")


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


def _case(geom, seed=5):
    from oracle import keras_ops as ko
    b, h, w, ci, co, k, s, pad = geom
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, h, w, ci, generator=g)
    wt = torch.randn(k, k, ci, co, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    sc, sh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.5
    y_shape = ko.conv2d(x[:1].double(), wt.double(), None, (s, s), pad).shape
    dy = torch.randn(b, *y_shape[1:], generator=g) * 1e-3
    return x, wt, sc, sh, dy


GEOMS = [(4, 19, 19, 256, 192, 3, 1, "same"), (3, 10, 10, 512, 256, 1, 1, "valid"), (2, 38, 38, 64, 128, 3, 1, "same"),
         (2, 20, 20, 128, 96, 2, 1, "same")]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("x_dt,y_dt", [(F16, F16), (F32, F16), (F16, F32)])
def test_forward_with_16bit_tensors_every_variant(geom, x_dt, y_dt, floatx):
    """Forward GEMM reading fp16 x (plain, and with the BatchNormalization + ReLU prologue and the statistics epilogue) and
    writing fp16 y, for every (tile, K depth, prefetch depth) variant; statistics are those of the UNROUNDED result."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    lib = _lib.load()
    b, h, w, ci, co, k, s, pad = geom
    x, wt, sc, sh, _ = _case(geom)
    xq = x.to(x_dt)
    bias = torch.linspace(-1, 1, co)
    ref_plain = ko.conv2d(xq.double(), wt.double(), bias.double(), (s, s), pad)
    ref_pro = ko.conv2d(torch.relu(xq.double() * sc.double() + sh.double()), wt.double(), None, (s, s), pad)
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (1, 1))
    rows = Kn.conv2d_stats_rows(desc)
    xd, wd, scd, shd, bd = xq.cuda(), wt.cuda(), sc.cuda(), sh.cuda(), bias.cuda()
    tol = 1.5e-3 + (5e-4 if y_dt != F32 else 0.0)
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for d in (0, 4):
                _lib.check(lib.dj_conv2d_tune_set(d, desc, cfg, 2), "tune_set")    # a registered split must be ignored for 16-bit y
            y0 = torch.full(ref_plain.shape, 3.0, dtype=y_dt, device="cuda")
            y1 = torch.empty(ref_plain.shape, dtype=y_dt, device="cuda")
            stats = torch.zeros(rows, 2, co, device="cuda")
            ws = torch.empty(2 * y0.numel(), device="cuda")
            Kn.conv2d_fwd(desc, xd, wd, bd, y0, relu=True, workspace=ws)
            Kn.conv2d_fwd(desc, xd, wd, None, y1, scd, shd, True, False, stats)
            torch.cuda.synchronize()
            assert rel_l2(y0, torch.relu(ref_plain)) <= tol, ("plain", cfg)
            assert rel_l2(y1, ref_pro) <= tol, ("prologue", cfg)
            st = stats.double().cpu().sum(0)
            flat = ref_pro.reshape(-1, co)
            assert float(((st[0] - flat.sum(0)).abs() / ((flat.shape[0] * (flat * flat).sum(0)).sqrt() + 1e-30)).max()) <= 1.5e-3, cfg
            assert rel_l2(st[1], (flat * flat).sum(0)) <= 3e-3, cfg
    finally:
        for d in (0, 4):
            _lib.check(lib.dj_conv2d_tune_set(d, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("sum_dt,y_dt", [(F16, F16), (F32, F16), (F16, F32)])
def test_residual_add_prologue_with_16bit_tensors(sum_dt, y_dt, floatx):
    """relu(bn(z) + shortcut) evaluated in the prologue of the next block's 1x1 convolution: z and the shortcut as fp16,
    the stored sum and the result as fp16 or fp32, every variant."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    lib = _lib.load()
    geom = (3, 19, 19, 256, 128, 1, 1, "valid")
    b, h, w, ci, co, k, s, pad = geom
    x, wt, sc, sh, _ = _case(geom, seed=9)
    g = torch.Generator().manual_seed(10)
    res = torch.randn(b, h, w, ci, generator=g)
    rsc, rsh = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.2
    xq, rq = x.to(F16), res.to(F16)
    a = torch.relu(xq.double() * sc.double() + sh.double() + rq.double() * rsc.double() + rsh.double())
    ref = ko.conv2d(a, wt.double(), None, (1, 1), "valid")
    desc = Kn.make_conv_desc(b, h, w, ci, co, (1, 1), (1, 1), "valid", (1, 1))
    rows = Kn.conv2d_stats_rows(desc)
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for d in (0, 4):
                _lib.check(lib.dj_conv2d_tune_set(d, desc, cfg, 1), "tune_set")
            y = torch.empty(ref.shape, dtype=y_dt, device="cuda")
            sm = torch.empty(x.shape, dtype=sum_dt, device="cuda")
            stats = torch.zeros(rows, 2, co, device="cuda")
            Kn.conv2d_fwd_addrelu(desc, xq.cuda(), wt.cuda(), None, y, sc.cuda(), sh.cuda(), rq.cuda(), rsc.cuda(), rsh.cuda(),
                                  sm, False, stats)
            torch.cuda.synchronize()
            assert rel_l2(y, ref) <= 1.5e-3 + (5e-4 if y_dt != F32 else 0), cfg
            assert rel_l2(sm, a) <= (5e-4 if sum_dt != F32 else 1e-6), cfg
            st = stats.double().cpu().sum(0)
            assert rel_l2(st[1], (ref.reshape(-1, co) ** 2).sum(0)) <= 3e-3, cfg
    finally:
        for d in (0, 4):
            _lib.check(lib.dj_conv2d_tune_set(d, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("geom", GEOMS + [(3, 19, 19, 256, 128, 1, 2, "valid")])
@pytest.mark.parametrize("dy_dt,dx_dt", [(BF16, BF16), (F32, BF16), (BF16, F32)])
def test_input_gradient_with_16bit_tensors_every_variant(geom, dy_dt, dx_dt, floatx):
    """Input gradient reading a bf16 dy and writing (and accumulating into) a bf16 dx, every variant; the stride-2 1x1 case
    is the compact GEMM + scatter; with the BatchNormalization backward statistics epilogue reading a fp16 z."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    lib = _lib.load()
    b, h, w, ci, co, k, s, pad = geom
    x, wt, sc, sh, dy = _case(geom)
    dyq = dy.to(dy_dt)
    x0 = torch.zeros(x.shape, dtype=torch.float64, requires_grad=True)
    ko.conv2d(x0, wt.double(), None, (s, s), pad).backward(dyq.double())
    ref = x0.grad
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (1, 1))
    g = torch.Generator().manual_seed(2)
    prev = (torch.randn(x.shape, generator=g) * float(ref.abs().mean())).to(dx_dt)
    z = torch.randn(x.shape, generator=g).to(F16)
    mean, invstd = torch.randn(ci, generator=g) * 0.1, torch.rand(ci, generator=g) + 0.5
    tol = 8e-3 + (4e-3 if dx_dt != F32 else 0.0)
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for d in (1, 9):
                _lib.check(lib.dj_conv2d_tune_set(d, desc, cfg, 2), "tune_set")
            dx = torch.full(x.shape, 5.0, dtype=dx_dt, device="cuda")
            acc = prev.cuda().clone()
            Kn.conv2d_dgrad(desc, dyq.cuda(), wt.cuda(), dx)
            Kn.conv2d_dgrad(desc, dyq.cuda(), wt.cuda(), acc, None, True)
            outs = [dx, acc]
            if s == 1:
                dx2 = torch.empty(x.shape, dtype=dx_dt, device="cuda")
                part = torch.zeros((b * h * w + 63) // 64, 2, ci, device="cuda")
                Kn.conv2d_dgrad_bnbwd(desc, dyq.cuda(), wt.cuda(), dx2, z.cuda(), mean.cuda(), invstd.cuda(), sc.cuda(), sh.cuda(),
                                      part)
                outs.append(dx2)
            torch.cuda.synchronize()
            assert rel_l2(dx, ref) <= tol, cfg
            assert rel_l2(acc, ref + prev.double()) <= tol, cfg
            if s == 1:
                assert rel_l2(dx2, ref) <= tol, cfg
                # the statistics are taken of the fp32 accumulators (before dx is rounded), masked by z*scale+shift > 0
                gm = ref.reshape(-1, ci) * ((z.double().reshape(-1, ci) * sc.double() + sh.double()) > 0)
                want0 = gm.sum(0)
                want1 = (gm * (z.double().reshape(-1, ci) - mean.double()) * invstd.double()).sum(0)
                got = part.double().cpu().sum(0)
                nat = (gm.shape[0] ** 0.5) * gm.norm(dim=0).clamp_min(1e-30)
                assert float(((got[0] - want0).abs() / nat).max()) <= 8e-3, cfg
                assert rel_l2(got[1], want1) <= 3e-2, cfg
    finally:
        for d in (1, 9):
            _lib.check(lib.dj_conv2d_tune_set(d, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("geom", GEOMS[:3] + [(5, 11, 13, 128, 64, 3, 2, "same")])
@pytest.mark.parametrize("x_dt,dy_dt", [(F16, BF16), (F32, BF16), (F16, F32)])
def test_weight_gradient_with_16bit_tensors_every_variant(geom, x_dt, dy_dt, floatx):
    """Weight gradient reading fp16 x (plain and through the BatchNormalization + ReLU prologue) and bf16 dy, every variant
    and two split-K factors."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from oracle import keras_ops as ko
    lib = _lib.load()
    b, h, w, ci, co, k, s, pad = geom
    x, wt, sc, sh, dy = _case(geom)
    xq, dyq = x.to(x_dt), dy.to(dy_dt)
    w_plain = wt.double().requires_grad_(True)
    ko.conv2d(xq.double(), w_plain, None, (s, s), pad).backward(dyq.double())
    w_pro = wt.double().requires_grad_(True)
    ko.conv2d(torch.relu(xq.double() * sc.double() + sh.double()), w_pro, None, (s, s), pad).backward(dyq.double())
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (1, 1))
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for splits in (1, 3):
                _lib.check(lib.dj_conv2d_tune_set(2, desc, cfg, splits), "tune_set")
                dw0 = torch.full(wt.shape, 2.0, device="cuda")
                dw1 = torch.zeros(wt.shape, device="cuda")
                Kn.conv2d_wgrad(desc, xq.cuda(), dyq.cuda(), dw0)
                Kn.conv2d_wgrad(desc, xq.cuda(), dyq.cuda(), dw1, sc.cuda(), sh.cuda(), True, dw_zeroed=True)
                torch.cuda.synchronize()
                assert rel_l2(dw0, w_plain.grad) <= 8e-3, (cfg, splits)
                assert rel_l2(dw1, w_pro.grad) <= 8e-3, (cfg, splits)
    finally:
        _lib.check(lib.dj_conv2d_tune_set(2, desc, -1, 1), "tune_set")


def test_typed_launches_are_refused_outside_their_domain(floatx):
    """A 16-bit tensor never goes through a silent conversion or the generic kernel: channel counts the branch-free
    kernels cannot take, a strided 3x3 input gradient and the exact-fp32 mode are errors."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    x = torch.zeros(2, 8, 8, 24, dtype=F16, device="cuda")
    w = torch.zeros(3, 3, 24, 32, device="cuda")
    y = torch.zeros(2, 8, 8, 32, dtype=F16, device="cuda")
    desc = Kn.make_conv_desc(2, 8, 8, 24, 32, (3, 3), (1, 1), "same", (1, 1))
    with pytest.raises(_lib.DjError, match="16-bit"):
        Kn.conv2d_fwd(desc, x, w, None, y)
    x2 = torch.zeros(2, 9, 9, 64, dtype=BF16, device="cuda")
    w2 = torch.zeros(3, 3, 64, 64, device="cuda")
    dy2 = torch.zeros(2, 5, 5, 64, dtype=BF16, device="cuda")
    desc2 = Kn.make_conv_desc(2, 9, 9, 64, 64, (3, 3), (2, 2), "same", (1, 1))
    with pytest.raises(_lib.DjError, match="16-bit"):
        Kn.conv2d_dgrad(desc2, dy2, w2, x2)
    floatx.set_floatx("float32")
    x3 = torch.zeros(2, 8, 8, 64, dtype=F16, device="cuda")
    w3 = torch.zeros(1, 1, 64, 64, device="cuda")
    y3 = torch.zeros(2, 8, 8, 64, device="cuda")
    with pytest.raises(_lib.DjError, match="mode 1"):
        Kn.conv2d_fwd(Kn.make_conv_desc(2, 8, 8, 64, 64, (1, 1)), x3, w3, None, y3)
    torch.cuda.synchronize()


@pytest.mark.parametrize("c,ld_extra", [(256, 0), (96, 32), (100, 0), (30, 2)])
def test_typed_elementwise_passes_match_the_float_ones(c, ld_extra, cuda):
    """dj_affine_act_t / dj_bn_bwd_reduce_t / dj_bn_bwd_apply_t / dj_relu_bwd_t / dj_copy2d_t against the float entry points
    on the same (rounded) values: identical up to the one rounding of a 16-bit result; vector and scalar code paths,
    channel slices (ld > C)."""
    from jpeg_detection_resnet_ssd_amd.engine import call, query
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    rows, ld = 3000, c + ld_extra
    g = torch.Generator().manual_seed(4)

    def mk(dt, scale=1.0):
        t = (torch.randn(rows, ld, generator=g) * scale).to(dt).cuda()
        return t, t.float()

    z16, z32 = mk(F16)
    dy16, dy32 = mk(BF16, 1e-3)
    m16, m32 = mk(F16)
    r16, r32 = mk(F16)
    sc, sh = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    mean, invstd = (torch.randn(c, generator=g) * 0.1).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    k0, k1, k2 = [(torch.randn(c, generator=g) * s_).cuda() for s_ in (1.0, 1e-4, 1e-4)]
    view = lambda t: t[:, :c]

    # affine + residual + relu -> fp16 and fp32
    for out_dt in (F16, F32):
        y_t = torch.zeros(rows, ld, dtype=out_dt, device="cuda")
        y_f = torch.zeros(rows, ld, device="cuda")
        call(*Kn.affine_act_call(z16, ld, sc, sh, r16, ld, sc, sh, y_t, ld, rows, c, 1))
        call(*Kn.affine_act_call(z32, ld, sc, sh, r32, ld, sc, sh, y_f, ld, rows, c, 1))
        torch.cuda.synchronize()
        assert torch.equal(view(y_t).float(), view(y_f).to(out_dt).float())
        assert float(y_t[:, c:].abs().max() if ld_extra else 0.0) == 0.0      # nothing written outside the C columns
    # BN backward statistics, every mask mode
    nr = query("dj_reduce_rows", rows)
    for mode in (0, 1, 2):
        p_t, p_f = torch.zeros(nr, 2, c, device="cuda"), torch.zeros(nr, 2, c, device="cuda")
        call(*Kn.bn_bwd_reduce_call(dy16, ld, z16, ld, m16, ld, mean, invstd, sc, sh, mode, rows, c, p_t))
        call(*Kn.bn_bwd_reduce_call(dy32, ld, z32, ld, m32, ld, mean, invstd, sc, sh, mode, rows, c, p_f))
        torch.cuda.synchronize()
        assert torch.equal(p_t, p_f), mode
        # apply: dz as bf16, the masked gradient accumulated into a bf16 tensor
        dz_t = torch.zeros(rows, ld, dtype=BF16, device="cuda")
        dz_f = torch.zeros(rows, ld, device="cuda")
        dm0 = (torch.randn(rows, ld, generator=g) * 1e-3).to(BF16).cuda()
        dm_t, dm_f = dm0.clone(), dm0.float()
        call(*Kn.bn_bwd_apply_call(dy16, ld, z16, ld, m16, ld, sc, sh, mode, k0, k1, k2, dz_t, ld, rows, c, dm_t, ld, 1))
        call(*Kn.bn_bwd_apply_call(dy32, ld, z32, ld, m32, ld, sc, sh, mode, k0, k1, k2, dz_f, ld, rows, c, dm_f, ld, 1))
        torch.cuda.synchronize()
        assert torch.equal(view(dz_t).float(), view(dz_f).to(BF16).float()), mode
        assert torch.equal(view(dm_t).float(), view(dm_f).to(BF16).float()), mode
    # relu backward with accumulate, copy with a change of type
    dx_t = (torch.ones(rows, ld) * 1e-3).to(BF16).cuda()
    dx_f = dx_t.float()
    call(*Kn.relu_bwd_call(dy16, ld, m16, ld, dx_t, ld, rows, c, 1))
    call(*Kn.relu_bwd_call(dy32, ld, m32, ld, dx_f, ld, rows, c, 1))
    cp = torch.zeros(rows, ld, device="cuda")
    call(*Kn.copy2d_call(z16, ld, cp, ld, rows, c, 0))
    back = torch.zeros(rows, ld, dtype=BF16, device="cuda")
    call(*Kn.copy2d_call(dy32, ld, back, ld, rows, c, 0))
    torch.cuda.synchronize()
    assert torch.equal(view(dx_t).float(), view(dx_f).to(BF16).float())
    assert torch.equal(view(cp), view(z32)) and torch.equal(view(back).float(), view(dy32).to(BF16).float())


def test_weight_shadows_as_gemm_operands(floatx):
    """dj_shadow_weights + the forward GEMM reading the fp16 shadow / the input gradient reading the bf16 shadow: the same
    results as with the fp32 master weights to the last bit (the kernels round the master weights the same way), every
    variant; shadows that sit 8-byte (not 16-byte) aligned, as slices of the flat shadow buffers do."""
    from jpeg_detection_resnet_ssd_amd import _lib
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from jpeg_detection_resnet_ssd_amd.engine import call
    lib = _lib.load()
    geom = (3, 19, 19, 128, 96, 3, 1, "same")
    b, h, w, ci, co, k, s, pad = geom
    x, wt, sc, sh, dy = _case(geom)
    n = wt.numel()
    flat = torch.zeros(n + 8, device="cuda")
    flat[4:4 + n] = wt.reshape(-1).cuda()
    f16 = torch.zeros(n + 8, dtype=F16, device="cuda")
    fbf = torch.zeros(n + 8, dtype=BF16, device="cuda")
    call("dj_shadow_weights", flat, f16, fbf, n + 8)
    w32, w16, wbf = (t[4:4 + n].view(wt.shape) for t in (flat, f16, fbf))
    assert w16.data_ptr() % 16 == 8 and torch.equal(w16.float(), w32.to(F16).float()) and torch.equal(wbf.float(), w32.to(BF16).float())
    desc = Kn.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (1, 1))
    xq, dyq = x.to(F16).cuda(), dy.to(BF16).cuda()
    try:
        for cfg in range(lib.dj_conv2d_tune_configs()):
            for d in (0, 1):
                _lib.check(lib.dj_conv2d_tune_set(d, desc, cfg, 1), "tune_set")
            ys = [torch.empty(b, h, w, co, dtype=F16, device="cuda") for _ in range(2)]
            dxs = [torch.empty(b, h, w, ci, dtype=BF16, device="cuda") for _ in range(2)]
            Kn.conv2d_fwd(desc, xq, w32, None, ys[0], sc.cuda(), sh.cuda(), True)
            Kn.conv2d_fwd(desc, xq, w16, None, ys[1], sc.cuda(), sh.cuda(), True)
            Kn.conv2d_dgrad(desc, dyq, w32, dxs[0])
            Kn.conv2d_dgrad(desc, dyq, wbf, dxs[1])
            torch.cuda.synchronize()
            assert torch.equal(ys[0], ys[1]) and torch.equal(dxs[0], dxs[1]), cfg
    finally:
        for d in (0, 1):
            _lib.check(lib.dj_conv2d_tune_set(d, desc, -1, 1), "tune_set")


@pytest.mark.parametrize("archi", ["deconv", "ssd_custom"])
def test_training_step_with_16bit_backbone_tensors(archi, floatx, monkeypatch):
    """SSD300 training step with the backbone's conv outputs / block sums held as fp16, their gradients as bf16 and the
    GEMMs reading 16-bit weight shadows (DJ_STORE16_MIN_ROWS=0: also at this test's batch of 2) against the fp64 oracle:
    the tolerances of the fp16-MFMA mode with fp32 tensors (tests/test_lowp_gpu.py: predictions 1e-2 rel-L2, 6e-2 max-norm,
    loss 1e-2) plus the storage rounding of ~100 fp16 tensors on the way (2^-11 each, random sign: +5e-3 rel-L2); the plan
    must really hold 16-bit tensors.  Gradients: at this batch size the full-graph gradient of the random-init network is
    chaotic under ANY perturbation of the forward pass -- the float16 arithmetic mode with fp32 tensors sits 0.4 (rel-L2)
    away from the exact-fp32 mode (measured, tools/store16_grad_noise.py; DESIGN.md section 5 "conditioning note") -- so the
    16-bit-storage run is held to the same distance from the exact-fp32 gradient as the fp32-storage run of its mode is;
    per-layer gradient parity at the stated tolerances is what tests/test_replay_gpu.py checks."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from oracle import ssd_resnet_dct as oracle
    monkeypatch.setenv("DJ_STORE16_MIN_ROWS", "0")
    model, sizes = workloads.build_ssd(archi)
    x, y_true = workloads.synthetic_batch(archi, sizes, 2)
    w0 = model.get_weights_dict()
    loss = model.train_on_batch(x, y_true)
    torch.cuda.synchronize()
    plan = model._plan(2, True, True)
    n16 = sum(1 for v in plan.values.values() if getattr(v, "buf", None) is not None and v.buf.dtype == F16)
    g16 = sum(1 for v in plan.values.values() if getattr(v, "grad", None) is not None and v.grad.buf.dtype == BF16)
    assert n16 >= 30 and g16 >= 30, (n16, g16)
    assert "flat16" in model._store        # the GEMMs read weight shadows
    y_pred = plan.outputs[0].buf.cpu().double()
    grads16 = model.flat_gradients.clone()
    wt = {k: torch.from_numpy(v).double() for k, v in w0.items()}
    ref = oracle.ssd_training_step(wt, [torch.from_numpy(a).double() for a in x], torch.from_numpy(y_true).double(), archi,
                                   lr=0.001, momentum=0.9)
    e_pred = float((y_pred - ref["y_pred"]).abs().max()) / float(ref["y_pred"].abs().max())
    e_l2 = rel_l2(y_pred[..., :25], ref["y_pred"][..., :25])
    e_loss = abs(loss - ref["loss"]) / abs(ref["loss"])

    def grads_of(fx, store16):
        monkeypatch.setenv("DJ_STORE16", store16)
        floatx.set_floatx(fx)
        m, _ = workloads.build_ssd(archi)
        m.set_weights_dict(w0)
        m.train_on_batch(x, y_true)
        torch.cuda.synchronize()
        assert not any(v.buf.dtype != F32 for v in m._plan(2, True, True).values.values() if getattr(v, "buf", None) is not None)
        return m.flat_gradients.clone()

    g_mode = grads_of("float16", "0")          # this arithmetic mode with fp32 tensors (round 2)
    g_exact = grads_of("float32", "0")         # exact fp32
    floatx.set_floatx("float16")
    e16, e32 = rel_l2(grads16, g_exact), rel_l2(g_mode, g_exact)
    print("16-bit storage %s: %d fp16 tensors, %d bf16 gradients; predictions max-norm %.2e rel-L2 %.2e, loss %.2e; "
          "gradient vs exact fp32: %.3f with 16-bit tensors, %.3f with fp32 tensors" % (archi, n16, g16, e_pred, e_l2, e_loss,
                                                                                      e16, e32))
    assert e_pred <= 6e-2 and e_l2 <= 1.5e-2 and e_loss <= 1e-2, (e_pred, e_l2, e_loss)
    assert np.isfinite(loss) and e16 <= 1.25 * e32 + 0.05, (e16, e32)

"""Per-launch oracle parity AT THE BENCHED SIZE ("make the parity evidence bite").

The training plan of the bench workload (SSD300 ResNet50-DCT `deconv`, batch 32) is lowered with the in-tree tuning
table (DJ_AUTOTUNE=table: exactly the registered tile variant / split-K factor of every geometry), one forward +
backward pass is run on the bench's synthetic batch, and every DISTINCT launch of the hot kernels is checked, at the
moment it is issued and on its own operands as they sit in HBM, against the CPU oracle in fp64:

  * implicit-GEMM convolutions in all three directions -- forward (with the BatchNormalization(+ReLU) prologue, the
    residual-add prologue, the fused bias / ReLU epilogue and the BN partial statistics), input gradient (plain,
    accumulating, strided scatter, Conv2DTranspose forward) and weight gradient (split-K into the cleared buffer):
    1e-3 max-norm forward, 1e-3 relative L2 for the gradients;
  * every training-mode BatchNormalization: the scale / shift the forward pass derived from the batch statistics, and
    the backward pass (dj_bn_bwd_reduce + finalize + apply) -- dz, dgamma, dbeta and the masked shortcut gradient --
    against torch.autograd through oracle.keras_ops.batch_norm_train on the same z and upstream gradient: 1e-3 rel-L2.

This is teacher forcing in the direction that needs no second pass: each layer's kernel is fed by the engine's own
upstream tensor and compared with the oracle's op on that same tensor, so no error of an earlier layer can hide or
compound.  Distinct = (entry point, geometry, prologue / epilogue flags); repeats of a geometry run the same code on
other data and are skipped to keep the oracle's CPU time to about a minute."""
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import keras_ops as ko

pytestmark = pytest.mark.gpu

ARCHI, BATCH = "deconv", 32
TOL = 1e-3


_DT = [torch.float64]     # dtype of the oracle's arithmetic (Replay sets it: fp64, fp32 in the reduced-precision cases)


def _f64(t):
    return None if t is None else t.detach().to("cpu", _DT[0])


def _pad_for(desc):
    """Explicit (left, right, top, bottom) zero padding that reproduces the descriptor's geometry (may crop)."""
    pb = (desc.out_h - 1) * desc.stride_h + (desc.kernel_h - 1) * desc.dilation_h + 1 - desc.in_h - desc.pad_top
    pr = (desc.out_w - 1) * desc.stride_w + (desc.kernel_w - 1) * desc.dilation_w + 1 - desc.in_w - desc.pad_left
    return (desc.pad_left, pr, desc.pad_top, pb)


def _oracle_conv(a, w, bias, desc):
    """ko.conv2d with the descriptor's explicit padding: a (B,H,W,Cin) fp64, w HWIO."""
    l, r, t, b = _pad_for(desc)
    a = F.pad(a.permute(0, 3, 1, 2), (l, r, t, b)).permute(0, 2, 3, 1)
    return ko.conv2d(a, w, bias, (desc.stride_h, desc.stride_w), "valid", (desc.dilation_h, desc.dilation_w))


def invstd_of(var):
    return torch.rsqrt(var + ko.BN_EPSILON)


def _rel_l2(got, ref):
    return float((got - ref).norm()) / (float(ref.norm()) + 1e-300)


def _rel_max(got, ref):
    return float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-300)


class Replay(object):
    """Wraps the kernels.conv2d_* wrappers and the BatchNormalization launches of `engine.call`."""

    def __init__(self, lowp=False):
        # lowp: a reduced-precision mode -- the oracle runs in fp32 there (the bounds are 1.5e-3 and up; fp64 would only
        # double the CPU time), in fp64 for the exact-fp32 mode
        self.dt = _DT[0] = torch.float32 if lowp else torch.float64
        self.seen = set()
        self.new = set()        # the launches THIS recorder checked (self.seen may be shared across test cases)
        self.rows = []          # (kind, geometry, metric name, value)
        self.cpu_s = 0.0
        self.bn_fin = {}        # id(k0 tensor) -> args of the dj_bn_bwd_finalize that produced k0 / k1 / k2
        self.bn_params = {}     # data_ptr(gamma) -> (gamma, beta, eps)

    # ---- convolutions ------------------------------------------------------------------------------------------
    @staticmethod
    def _geom(desc):
        return tuple(getattr(desc, n) for n, _ in type(desc)._fields_[:15])

    def _note(self, kind, geom, **metrics):
        for k, v in metrics.items():
            self.rows.append((kind, geom, k, v))

    def fwd(self, orig):
        def run(desc, x, w, bias, y, pro_scale=None, pro_shift=None, pro_relu=False, relu=False, stats=None,
                y_zeroed=False, **kw):     # kw: workspace / stats_may_split (split-K through slabs)
            key = ("fwd", self._geom(desc), pro_scale is not None, bool(pro_relu), bool(relu), stats is not None,
                   bool(y_zeroed), bias is not None, bool(kw.get("stats_may_split")), str(x.dtype), str(y.dtype))
            if key in self.seen:
                return orig(desc, x, w, bias, y, pro_scale, pro_shift, pro_relu, relu, stats, y_zeroed, **kw)
            self.seen.add(key)
            self.new.add(key)
            before = y.clone() if y_zeroed else None
            orig(desc, x, w, bias, y, pro_scale, pro_shift, pro_relu, relu, stats, y_zeroed, **kw)
            torch.cuda.synchronize()
            t0 = time.time()
            a = _f64(x)
            if pro_scale is not None:
                a = a * _f64(pro_scale) + _f64(pro_shift)
                if pro_relu:
                    a = ko.relu(a)
            raw = _oracle_conv(a, _f64(w), None, desc)
            ref = raw + _f64(bias) if bias is not None else raw
            if relu:
                ref = ko.relu(ref)
            if before is not None:
                assert float(before.abs().max()) == 0.0, "split-K forward output was not cleared"
            m = dict(y_max=_rel_max(_f64(y), ref), y_l2=_rel_l2(_f64(y), ref))
            if stats is not None:
                st = _f64(stats).sum(dim=0)
                s_ref, q_ref = raw.sum(dim=(0, 1, 2)), (raw * raw).sum(dim=(0, 1, 2))
                rows = raw.shape[0] * raw.shape[1] * raw.shape[2]
                # a column sum may cancel: measure it against sqrt(rows * sum of squares), its natural scale
                m["stats_sum"] = float(((st[0] - s_ref).abs() / (rows * q_ref).sqrt().clamp_min(1e-30)).max())
                m["stats_sq"] = _rel_max(st[1], q_ref)
            self.cpu_s += time.time() - t0
            self._note("fwd", key[1:], **m)
        return run

    def fwd_addrelu(self, orig):
        def run(desc, x, w, bias, y, pro_scale, pro_shift, res, res_scale=None, res_shift=None, sum_out=None,
                relu=False, stats=None, **kw):
            key = ("fwd_addrelu", self._geom(desc), res_scale is not None, sum_out is not None, bool(relu),
                   stats is not None, str(x.dtype), str(y.dtype), str(sum_out.dtype) if sum_out is not None else "")
            if key in self.seen:
                return orig(desc, x, w, bias, y, pro_scale, pro_shift, res, res_scale, res_shift, sum_out, relu, stats, **kw)
            self.seen.add(key)
            self.new.add(key)
            orig(desc, x, w, bias, y, pro_scale, pro_shift, res, res_scale, res_shift, sum_out, relu, stats, **kw)
            torch.cuda.synchronize()
            t0 = time.time()
            r = _f64(res)
            if res_scale is not None:
                r = r * _f64(res_scale) + _f64(res_shift)
            a = ko.relu(_f64(x) * _f64(pro_scale) + _f64(pro_shift) + r)
            raw = _oracle_conv(a, _f64(w), None, desc)
            ref = raw + _f64(bias) if bias is not None else raw
            if relu:
                ref = ko.relu(ref)
            m = dict(y_max=_rel_max(_f64(y), ref), y_l2=_rel_l2(_f64(y), ref))
            if sum_out is not None:
                m["sum_max"] = _rel_max(_f64(sum_out), a)
            if stats is not None:
                st = _f64(stats).sum(dim=0)
                q_ref = (raw * raw).sum(dim=(0, 1, 2))
                rows = raw.shape[0] * raw.shape[1] * raw.shape[2]
                m["stats_sum"] = float(((st[0] - raw.sum(dim=(0, 1, 2))).abs() / (rows * q_ref).sqrt().clamp_min(1e-30)).max())
                m["stats_sq"] = _rel_max(st[1], q_ref)
            self.cpu_s += time.time() - t0
            self._note("fwd_addrelu", key[1:], **m)
        return run

    def dgrad(self, orig):
        def run(desc, dy, w, dx, bias=None, beta=False, **kw):
            key = ("dgrad", self._geom(desc), bias is not None, bool(beta), str(dy.dtype), str(dx.dtype))
            if key in self.seen:
                return orig(desc, dy, w, dx, bias, beta, **kw)
            self.seen.add(key)
            self.new.add(key)
            before = dx.clone() if beta else None
            orig(desc, dy, w, dx, bias, beta, **kw)
            torch.cuda.synchronize()
            t0 = time.time()
            x0 = torch.zeros(tuple(dx.shape), dtype=_DT[0], requires_grad=True)
            _oracle_conv(x0, _f64(w), None, desc).backward(_f64(dy))
            ref = x0.grad
            if bias is not None:          # Conv2DTranspose forward: + bias[channel of dx]
                ref = ref + _f64(bias)
            got = _f64(dx) - _f64(before) if before is not None else _f64(dx)
            if before is not None and dx.dtype != torch.float32:
                # accumulating into a bf16 tensor: the sum is rounded, so the sum is what can be compared (the difference
                # of two rounded tensors carries the rounding of the larger one)
                got, ref = _f64(dx), ref + _f64(before)
            self.cpu_s += time.time() - t0
            self._note("dgrad", key[1:], dx_l2=_rel_l2(got, ref), dx_max=_rel_max(got, ref))
        return run

    def dgrad_bnbwd(self, orig):
        """Input gradient whose epilogue also takes the BatchNormalization backward sums of the tensor it writes
        (dj_conv2d_nhwc_dgrad_bnbwd): dx against the oracle, the column totals of the partial rows against fp64 sums over
        (dx, z) on their natural scale.  (dgamma / dbeta / dz of that layer are checked again where its apply pass runs.)"""
        def run(desc, dy, w, dx, z, mean, invstd, scale, shift, partial):
            key = ("dgrad", self._geom(desc), "bnbwd", scale is not None, str(dy.dtype), str(dx.dtype), str(z.dtype))
            if key in self.seen:
                return orig(desc, dy, w, dx, z, mean, invstd, scale, shift, partial)
            self.seen.add(key)
            self.new.add(key)
            orig(desc, dy, w, dx, z, mean, invstd, scale, shift, partial)
            torch.cuda.synchronize()
            t0 = time.time()
            x0 = torch.zeros(tuple(dx.shape), dtype=_DT[0], requires_grad=True)
            _oracle_conv(x0, _f64(w), None, desc).backward(_f64(dy))
            c = dx.shape[-1]
            g, zz = _f64(dx).reshape(-1, c), _f64(z).reshape(-1, c)
            if scale is not None:
                g = g * ((zz * _f64(scale) + _f64(shift)) > 0).to(_DT[0])
            want0, want1 = g.sum(0), (g * (zz - _f64(mean)) * _f64(invstd)).sum(0)
            got = _f64(partial).sum(0)
            n_rows = float(g.shape[0])
            nat0 = (n_rows ** 0.5) * g.norm(dim=0).clamp_min(1e-300)
            nat1 = (n_rows ** 0.5) * (g * (zz - _f64(mean)) * _f64(invstd)).norm(dim=0).clamp_min(1e-300)
            self.cpu_s += time.time() - t0
            self._note("dgrad", key[1:], dx_l2=_rel_l2(_f64(dx), x0.grad), dx_max=_rel_max(_f64(dx), x0.grad),
                       sum_g_nat=float(((got[0] - want0).abs() / nat0).max()),
                       sum_gxhat_nat=float(((got[1] - want1).abs() / nat1).max()))
        return run

    def wgrad(self, orig):
        def run(desc, x, dy, dw, pro_scale=None, pro_shift=None, pro_relu=False, dw_zeroed=False):
            key = ("wgrad", self._geom(desc), pro_scale is not None, bool(pro_relu), bool(dw_zeroed), str(x.dtype),
                   str(dy.dtype))
            if key in self.seen:
                return orig(desc, x, dy, dw, pro_scale, pro_shift, pro_relu, dw_zeroed)
            self.seen.add(key)
            self.new.add(key)
            before = dw.clone() if dw_zeroed else None
            orig(desc, x, dy, dw, pro_scale, pro_shift, pro_relu, dw_zeroed)
            torch.cuda.synchronize()
            t0 = time.time()
            a = _f64(x)
            if pro_scale is not None:
                a = a * _f64(pro_scale) + _f64(pro_shift)
                if pro_relu:
                    a = ko.relu(a)
            w0 = torch.zeros(tuple(dw.shape), dtype=_DT[0], requires_grad=True)
            _oracle_conv(a, w0, None, desc).backward(_f64(dy))
            if before is not None:
                assert float(before.abs().max()) == 0.0, "weight-gradient buffer was not cleared before its only writer"
            self.cpu_s += time.time() - t0
            self._note("wgrad", key[1:], dw_l2=_rel_l2(_f64(dw), w0.grad), dw_max=_rel_max(_f64(dw), w0.grad))
        return run

    # ---- BatchNormalization ------------------------------------------------------------------------------------
    def call(self, orig):
        def run(name, *args):
            if name == "dj_bn_bwd_finalize":
                # (part, nr, rows, gamma, mean, invstd, dgamma, dbeta, k0, k1, k2, c)
                self.bn_fin[args[8].data_ptr()] = args
                return orig(name, *args)
            if name == "dj_bn_bwd_apply_t":      # tensors that carry their storage type (16-bit activations / gradients)
                (dy, _, ld_dy, z, _, ld_z, mask_y, _, ld_y, scale, shift, mode, k0, k1, k2, dz, _, ld_dz, rows, c, dm, _, ld_dm,
                 dm_beta) = args
            elif name == "dj_bn_bwd_apply":
                (dy, ld_dy, z, ld_z, mask_y, ld_y, scale, shift, mode, k0, k1, k2, dz, ld_dz, rows, c, dm, ld_dm,
                 dm_beta) = args
            else:
                return orig(name, *args)
            fin = self.bn_fin[k0.data_ptr()]
            gamma, dgamma, dbeta = fin[3], fin[6], fin[7]
            key = ("bn", int(rows), int(c), int(mode), dm is not None, int(dm_beta), int(ld_dy), int(ld_z),
                   str(dy.dtype), str(z.dtype), str(dz.dtype))
            if key in self.seen:
                return orig(name, *args)
            self.seen.add(key)
            self.new.add(key)
            dm_before = dm.clone() if (dm is not None and dm_beta) else None
            orig(name, *args)
            torch.cuda.synchronize()
            t0 = time.time()

            def rows_view(t, ld):
                # (rows, c) view of a buffer whose pixel stride may exceed c (channel slice of a concat buffer)
                return _f64(torch.as_strided(t, (int(rows), int(c)), (int(ld), 1)))

            z64 = rows_view(z, ld_z).requires_grad_(True)
            g64 = _f64(gamma).clone().requires_grad_(True)
            beta_param = self.bn_params[gamma.data_ptr()][1]
            b64 = _f64(beta_param).clone().requires_grad_(True)
            y, mean, var = ko.batch_norm_train(z64.view(1, 1, int(rows), int(c)), g64, b64)
            y = y.view(int(rows), int(c))
            up = rows_view(dy, ld_dy)
            if mode == 1:
                up = up * (rows_view(mask_y, ld_y) > 0).to(_DT[0])
            out = ko.relu(y) if mode == 2 else y
            (out * up).sum().backward()
            invstd = torch.rsqrt(var + ko.BN_EPSILON)
            sc_ref = (g64 * invstd).detach()
            sh_ref = (b64 - mean * g64 * invstd).detach()
            # dgamma / dbeta are column sums over `rows` pixels and may cancel -- for a BatchNormalization whose consumers
            # are 1x1 convolutions followed by another BatchNormalization (the input BN of this graph) dbeta is
            # ANALYTICALLY zero and both sides hold rounding noise -- so they are measured against the natural scale of
            # such a sum, sqrt(rows) * ||column||_2, and additionally as a relative L2 error when the reference is not
            # itself at the noise level of that scale
            n_rows = float(int(rows))
            xhat = ((z64 - mean) * invstd_of(var)).detach()
            nat_b = (n_rows ** 0.5) * up.norm(dim=0).clamp_min(1e-300)
            nat_g = (n_rows ** 0.5) * (up * xhat).norm(dim=0).clamp_min(1e-300)
            m = dict(dz_l2=_rel_l2(rows_view(dz, ld_dz), z64.grad),
                     dgamma_nat=float(((_f64(dgamma) - g64.grad).abs() / nat_g).max()),
                     dbeta_nat=float(((_f64(dbeta) - b64.grad).abs() / nat_b).max()),
                     scale_max=_rel_max(_f64(scale), sc_ref),
                     shift_max=float((_f64(shift) - sh_ref).abs().max())
                     / (float(sh_ref.abs().max()) + float((mean.detach() * sc_ref).abs().max()) + 1e-300))
            if float(g64.grad.norm()) > 1e-4 * float(nat_g.norm()):
                m["dgamma_l2"] = _rel_l2(_f64(dgamma), g64.grad)
            if float(b64.grad.norm()) > 1e-4 * float(nat_b.norm()):
                m["dbeta_l2"] = _rel_l2(_f64(dbeta), b64.grad)
            if dm is not None:
                got = rows_view(dm, ld_dm)
                if dm_before is not None:
                    got = got - rows_view(dm_before, ld_dm)
                m["shortcut_l2"] = _rel_l2(got, up)
            self.cpu_s += time.time() - t0
            self._note("bn", key[1:], **m)
        return run


# One case per BASELINE.json configuration that fits one GPU, at the size its tuning-table entries were measured for
# (VERDICT r2 item 1): (name, builder, batch, floatx).  Cases of one arithmetic mode share the set of launches already
# checked, so a geometry that several workloads have in common (the 19x19 stages, the heads) costs oracle time once.
CASES = [
    ("deconv_ssd_b32_f32", "ssd:deconv", 32, "float32"),            # the bench workload (north star)
    ("ssd_custom_b32_f32", "ssd:ssd_custom", 32, "float32"),        # config 3
    ("deconv_classifier_b64_f32", "cls:deconv", 64, "float32"),     # config 2
    ("ssd_custom_b32_f16", "ssd:ssd_custom", 32, "float16"),        # config 5 (late_concat_rfa_thinner backbone)
    ("up_sampling_b32_f16", "ssd:up_sampling", 32, "float16"),      # config 5 (up_sampling_rfa fusion)
    ("deconv_ssd_b32_x3", "ssd:deconv", 32, "float32x3"),           # the bench workload in the split-bf16 fp32 arithmetic
]
_SEEN = {}      # floatx -> launches already replayed in an earlier case of that mode

# fp32: 1e-3 everywhere (max-norm forward, rel-L2 gradients).  float16 (fp16 forward / bf16 gradient operands, fp32
# accumulate): the operand-rounding tolerances of tests/test_lowp_gpu.py, rel-L2 1.5e-3 forward / 8e-3 gradients; squares of
# forward values (the BatchNormalization statistics) twice the forward tolerance.  Tensors held in 16 bits in HBM add
# their storage rounding: 2^-11 per fp16 tensor read or written (activations), 2^-8 per bf16 tensor (gradients), i.e.
# BatchNormalization's dz, read from a bf16 gradient and a fp16 z and stored as bf16, is bounded by 8e-3 like the GEMM
# gradients.
# float32x3 (fp32 tensors, three bf16 MFMAs per product, ~4e-6 rel-L2 per GEMM): held to the bounds of the exact-fp32 mode,
# against the fp64 oracle; only the "natural scale" bound on the statistics sums, which is a rounding-level bound (3e-5 for
# fp32 MFMA), widens to 1e-4.
TOLS = {
    "float32": dict(y_max=TOL, y_l2=TOL, sum_max=TOL, stats_sq=TOL, stats_sum=TOL, dx_l2=TOL, dx_max=TOL, dw_l2=TOL, dw_max=TOL, dz_l2=TOL,
                    dgamma_l2=TOL, dbeta_l2=TOL, scale_max=TOL, shift_max=TOL, shortcut_l2=TOL, nat=3e-5),
    "float32x3": dict(y_max=TOL, y_l2=TOL, sum_max=TOL, stats_sq=TOL, stats_sum=TOL, dx_l2=TOL, dx_max=TOL, dw_l2=TOL, dw_max=TOL, dz_l2=TOL,
                      dgamma_l2=TOL, dbeta_l2=TOL, scale_max=TOL, shift_max=TOL, shortcut_l2=TOL, nat=1e-4),
    "float16": dict(y_max=None, y_l2=1.5e-3, sum_max=1.5e-3, stats_sq=3e-3, stats_sum=1.5e-3, dx_l2=8e-3, dx_max=None, dw_l2=8e-3, dw_max=None,
                    dz_l2=8e-3, dgamma_l2=8e-3, dbeta_l2=8e-3, scale_max=TOL, shift_max=TOL, shortcut_l2=8e-3, nat=1.5e-3),
}


def _build(kind, batch):
    """-> (model, x, y) of one workload, weights perturbed so that no bias / beta sits at its zero initialisation."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from test_ssd_gpu import perturb_weights
    what, archi = kind.split(":")
    if what == "ssd":
        model, sizes = workloads.build_ssd(archi)
        perturb_weights(model)
        x, y = workloads.synthetic_batch(archi, sizes, batch, fast=True)
        return model, x, y
    from jpeg_detection_resnet_ssd_amd.data import synthetic_dct as sd
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
    from jpeg_detection_resnet_ssd_amd.vgg_jpeg_keras.networks.resnet_dct import ResNet50Custom
    K.clear_session()
    K.set_random_seed(42)
    model = ResNet50Custom(weights=None, archi=archi)      # classification_part/vgg_jpeg_keras/networks/resnet_dct.py:603-642
    model.compile(optimizer=SGD(lr=0.1, momentum=0.9, decay=1e-4, nesterov=True), loss="categorical_crossentropy")
    perturb_weights(model)
    x = sd.fast_dct_batch(batch, seed=1234, grid=28, split_chroma=(archi == "deconv"))
    rng = np.random.default_rng(1234)
    y = np.zeros((batch, 1000), np.float32)
    y[np.arange(batch), rng.integers(0, 1000, batch)] = 1.0
    return model, x, y


@pytest.mark.parametrize("name,kind,batch,floatx", CASES, ids=[c[0] for c in CASES])
def test_every_distinct_launch_of_the_step_matches_the_oracle(name, kind, batch, floatx, cuda, monkeypatch):
    assert os.environ.get("DJ_AUTOTUNE") == "table", "the registered (tile variant, split-K) choices must be the ones that run"
    from jpeg_detection_resnet_ssd_amd import engine
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.keras import layers as L
    K.set_floatx(floatx)
    try:
        model, x, y = _build(kind, batch)
        plan = model._plan(batch, True, True)
        # the table OF THIS MODE must really cover the plan: an untuned geometry would run the launcher's default instead
        names = [n for n, _ in type(plan.conv_calls[0][1])._fields_][:15]
        missing = [(d,) + tuple(getattr(desc, n) for n in names) for d, desc, _ in plan.conv_calls
                   if (d,) + tuple(getattr(desc, n) for n in names) not in engine._TUNED]
        assert not missing, "%d conv launches of the %s plan have no entry in the %s tuning table: %s" % (
            len(missing), name, floatx, missing[:4])
        model._upload(plan, x, y)
        plan.run_forward()           # warm pass (moving statistics, lazy module loads) without the recorder
        plan.run_backward()
        torch.cuda.synchronize()

        rp = Replay(lowp=(floatx not in ("float32", "float32x3")))
        rp.seen = _SEEN.setdefault(floatx, set())
        n_before = len(rp.seen)
        for lyr in model.layers:
            if isinstance(lyr, L.BatchNormalization):
                rp.bn_params[lyr.gamma.param.data_ptr()] = (lyr.gamma.param, lyr.beta.param, lyr.epsilon)
                assert lyr.epsilon == ko.BN_EPSILON
        monkeypatch.setattr(Kn, "conv2d_fwd", rp.fwd(Kn.conv2d_fwd))
        monkeypatch.setattr(Kn, "conv2d_fwd_addrelu", rp.fwd_addrelu(Kn.conv2d_fwd_addrelu))
        monkeypatch.setattr(Kn, "conv2d_dgrad", rp.dgrad(Kn.conv2d_dgrad))
        monkeypatch.setattr(Kn, "conv2d_dgrad_bnbwd", rp.dgrad_bnbwd(Kn.conv2d_dgrad_bnbwd))
        monkeypatch.setattr(Kn, "conv2d_wgrad", rp.wgrad(Kn.conv2d_wgrad))
        monkeypatch.setattr(L, "call", rp.call(L.call))
        t0 = time.time()
        plan.run_forward()
        plan.run_backward()
        torch.cuda.synchronize()
        wall = time.time() - t0
    finally:
        K.set_floatx("float32")

    kinds = {}
    for kind_, geom, metric, value in rp.rows:
        kinds.setdefault((kind_, metric), []).append((value, geom))
    new = [s_ for s_ in rp.seen if s_ in rp.new]
    print("\n%s: replayed %d distinct launches not seen in an earlier %s case (%d known) in %.0f s (oracle CPU time %.0f s)"
          % (name, len(new), floatx, n_before, wall, rp.cpu_s))
    for (kind_, metric), vals in sorted(kinds.items()):
        worst = max(vals)
        print("  %-12s %-12s n=%3d  median %.2e  max %.2e  at %s"
              % (kind_, metric, len(vals), float(np.median([v for v, _ in vals])), worst[0], worst[1]))
    n_kind = {k: sum(1 for s_ in new if s_[0] == k) for k in ("fwd", "fwd_addrelu", "dgrad", "wgrad", "bn")}
    if name == "deconv_ssd_b32_f32":
        # the deconv SSD300 graph: 76 convolutions + 2 transposed ones, 53 BatchNormalization layers
        assert n_kind["fwd"] >= 25 and n_kind["dgrad"] >= 25 and n_kind["wgrad"] >= 30 and n_kind["bn"] >= 10, n_kind
        assert n_kind["fwd_addrelu"] >= 3, n_kind
        # the conv -> BN -> ReLU -> conv chains of the bottleneck blocks take the BatchNormalization backward sums in the GEMM
        assert sum(1 for s_ in new if s_[0] == "dgrad" and "bnbwd" in s_) >= 5, sorted(s_ for s_ in new if s_[0] == "dgrad")
    else:
        # every other workload brings geometries of its own (38x38x64 stages and the small-BN split stages of ssd_custom,
        # the 28x28 / 14x14 / 7x7 maps of the classifier at batch 64, ...)
        assert n_kind["fwd"] + n_kind["fwd_addrelu"] >= 8 and n_kind["dgrad"] >= 8 and n_kind["wgrad"] >= 8, n_kind
    tol = TOLS[floatx]
    bad = []
    for kind_, geom, metric, value in rp.rows:
        bound = tol["nat"] if metric.endswith("_nat") else tol[metric]
        if bound is not None and not value <= bound:
            bad.append((kind_, geom, metric, value, bound))
    assert not bad, bad[:10]

#!/usr/bin/env python
"""Training-throughput benchmark of the hot path: images/sec of ResNet50-DCT SSD300 training steps
(forward + multibox loss + backward + Keras-SGD update, data-parallel gradient all-reduce when N > 1).

    python bench.py --gpus N --steps K --warmup W [--archi deconv|ssd_custom|up_sampling] [--batch 32]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`,
one rank per GPU over RCCL.  Inputs (synthetic JPEG-DCT coefficients + encoder-made targets) are resident
in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s measured with a float4 copy)


def conv_bytes(desc, x_bytes=4, y_bytes=4):
    """Algorithmic HBM bytes of one conv launch in any direction: its two operands read once, its result written once, each
    at the width it has in HBM (x / dx and y / dy of the geometry: 4 bytes per element, or 2 where the plan holds the
    tensor as fp16 / bf16; the weights and their gradient are always fp32; padding columns of a fused predictor-head filter
    bank are not counted)."""
    out_c = getattr(desc, "algorithmic_out_c", desc.out_c)
    return (float(x_bytes) * desc.batch * desc.in_h * desc.in_w * desc.in_c
            + 4.0 * desc.kernel_h * desc.kernel_w * desc.in_c * out_c
            + float(y_bytes) * desc.batch * desc.out_h * desc.out_w * out_c)


def conv_flops(desc):
    out_c = getattr(desc, "algorithmic_out_c", desc.out_c)     # fused predictor heads run on a zero-padded filter bank
    return 2.0 * desc.batch * desc.out_h * desc.out_w * out_c * desc.kernel_h * desc.kernel_w * desc.in_c


def measure_conv_kernels(model, plan):
    """One extra (untimed-for-throughput) step with every implicit-GEMM launch bracketed by HIP events
    on the launch stream: -> dict(total_ms, flop, launches)."""
    from jpeg_detection_resnet_ssd_amd import kernels as Kn
    records = []
    originals = {}

    def wrap(name):
        f = getattr(Kn, name)
        originals[name] = f

        def timed(desc, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(desc, *a, **kw)
            e1.record()
            # positional tensors: fwd (x, w, bias, y ...), fwd_addrelu (x, w, bias, y, sc, sh, res, rsc, rsh, sum_out ...),
            # dgrad[_bnbwd] (dy, w, dx[, z ...]), wgrad (x, dy, dw ...)
            extra, n_in = 0.0, desc.batch * desc.in_h * desc.in_w * desc.in_c
            if name in ("conv2d_fwd", "conv2d_fwd_addrelu"):
                xb, yb = a[0].element_size(), a[3].element_size()
                if name == "conv2d_fwd_addrelu":   # residual operand read, the sum written
                    sum_out = a[9] if len(a) > 9 else kw.get("sum_out")
                    extra = float(a[6].element_size() + (sum_out.element_size() if sum_out is not None else 0)) * n_in
            elif name in ("conv2d_dgrad", "conv2d_dgrad_bnbwd"):
                xb, yb = a[2].element_size(), a[0].element_size()
                if name == "conv2d_dgrad_bnbwd":   # the BatchNormalization input z is read beside the GEMM operands
                    extra = float(a[3].element_size()) * n_in
            else:
                xb, yb = a[0].element_size(), a[1].element_size()
            records.append((e0, e1, conv_flops(desc), conv_bytes(desc, xb, yb) + extra))
            return r
        setattr(Kn, name, timed)

    for n in ("conv2d_fwd", "conv2d_fwd_addrelu", "conv2d_dgrad", "conv2d_dgrad_bnbwd", "conv2d_wgrad"):
        wrap(n)
    try:
        plan.side_enabled = False      # one stream: a launch's events bracket that launch alone
        model.run_train_step(plan)
        torch.cuda.synchronize()
    finally:
        plan.side_enabled = True
        for n, f in originals.items():
            setattr(Kn, n, f)
    total_ms = sum(r[0].elapsed_time(r[1]) for r in records)
    flop = sum(r[2] for r in records)
    # what an event pair measures around NOTHING on this stream (the two marker packets): a bracket's share that is not the
    # kernel -- reported beside the raw figures, which stay the conservative ones `achieved` is computed from
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a, b in pairs:
        a.record()
        b.record()
    torch.cuda.synchronize()
    empty_us = 1e3 * float(np.median([a.elapsed_time(b) for a, b in pairs]))
    return dict(total_ms=total_ms, flop=flop, launches=len(records), bytes=sum(r[3] for r in records), empty_pair_us=empty_us)


def available_cores():
    """CPU cores this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(archi, batch, budget_s=25.0):
    """The oracle (CPU restatement of the reference's Keras path, fp32) timed on this host's cores on a
    bounded sample: whole training steps at a small batch until `budget_s` seconds are used."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from oracle import ssd_resnet_dct as oracle
    cores = min(available_cores(), 32)
    torch.set_num_threads(cores)
    model, sizes = workloads.build_ssd(archi, compile_model=False)
    x, y = workloads.synthetic_batch(archi, sizes, batch, fast=True)
    w = {s.key: torch.as_tensor(s.initializer(s.shape), dtype=torch.float32) for s in model.weight_specs}
    xs = [torch.from_numpy(a) for a in x]
    yt = torch.from_numpy(y)
    times, t_start = [], time.perf_counter()
    while True:
        t0 = time.perf_counter()
        oracle.ssd_training_step(w, xs, yt, archi)
        times.append(time.perf_counter() - t0)
        print("cpu_baseline: step %d took %.2f s on %d threads" % (len(times), times[-1], cores), file=sys.stderr,
              flush=True)
        if time.perf_counter() - t_start > budget_s or len(times) >= 40:
            break
    use = times[1:] if len(times) > 1 else times   # first step pays one-time allocator / thread-pool start-up
    med = float(np.median(use))
    return {"value": batch / med, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d oracle training steps (PyTorch-CPU fp32 restatement of the reference's Keras graph, %s SSD300) "
                      "at batch %d; median step %.2f s" % (len(use), archi, batch, med)}


# `dtype`: the arithmetic type of the results (a short token); `arithmetic`: how the convolutions get there
DTYPE_NAME = {"float32": "f32", "float32_mfma": "f32", "float32x6": "f32", "float32x3": "f32-bf16x3",
              "float16": "f16/bf16-mfma+f32-acc", "bfloat16": "bf16-mfma+f32-acc"}
ARITHMETIC = {
    "float32": "fp32 tensors, fp32 accumulation, fp32 results: per layer the fp32 MFMA kernel or the split-bf16 kernel (operands cut "
               "into three bf16 pieces = all 24 significant bits, six bf16 MFMAs per product, dropped terms 2^-24), whichever the "
               "tuning table measured faster; both 1.4e-7..4.6e-7 rel-L2 from the fp64 oracle per GEMM (tests/test_x3_gpu.py)",
    "float32_mfma": "fp32 tensors, fp32 MFMA instructions only (v_mfma_f32_32x32x2_f32)",
    "float32x6": "fp32 tensors, every product as 6 bf16 MFMAs on hi/mid/lo operand pieces (fp32-grade), fp32 accumulation",
    "float32x3": "fp32 tensors, every product as 3 bf16 MFMAs on hi/lo operand pieces (~4e-6 rel-L2 per GEMM), fp32 accumulation",
    "float16": "fp16 MFMA forward / bf16 MFMA gradients, fp32 accumulation; fp16 activations, bf16 gradients, 16-bit weight shadows "
               "in HBM; fp32 master weights, statistics and optimizer state",
    "bfloat16": "bf16 MFMA in every GEMM, fp32 accumulation, fp32 tensors",
}


def build_classifier(archi, batch, seed=1234):
    """BASELINE config 2: the ResNet50-DCT classifier (`ResNet50Custom`, classification_part/.../resnet_dct.py:317-452) with
    the classification trainer's optimizer and loss (config/resnet/config_file.py:58-65), synthetic 224x224 JPEG-DCT
    inputs (Y 28x28x64, chroma 14x14) and one-hot ImageNet labels."""
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
    from jpeg_detection_resnet_ssd_amd.vgg_jpeg_keras.networks.resnet_dct import ResNet50Custom
    K.clear_session()
    K.set_random_seed(42)
    model = ResNet50Custom(weights=None, archi=archi)
    model.compile(optimizer=SGD(lr=0.1, momentum=0.9, decay=1e-4, nesterov=True), loss="categorical_crossentropy")
    rng = np.random.default_rng(seed)
    x = [rng.normal(0, 30, (batch,) + tuple(int(d) for d in t.shape[1:])).astype(np.float32) for t in model.inputs]
    y = np.eye(1000, dtype=np.float32)[rng.integers(0, 1000, batch)]
    return model, (x if len(x) > 1 else x[0]), y


def plan_gflop_per_image(plan, batch):
    """Algorithmic conv FLOPs (forward + both gradients) of one training step of `plan`, per image."""
    return sum(conv_flops(desc) for _d, desc, _f in plan.conv_calls) / batch / 1e9


def run_workload(archi, floatx, batch, steps, warmup, rank=0, world=1):
    """Build the SSD300 training workload `archi` under arithmetic mode `floatx`, run `warmup` untimed and `steps` timed
    training steps on a resident batch (barrier + synchronize on both sides, max over ranks) -> (elapsed seconds, last
    loss, model, plan).  The caller's floatx is restored."""
    from jpeg_detection_resnet_ssd_amd import dist as djdist
    from jpeg_detection_resnet_ssd_amd import workloads
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    prev = K.floatx()
    K.set_floatx(floatx)
    try:
        if archi.startswith("cls:"):
            model, x, y = build_classifier(archi[4:], batch, seed=1234 + rank)
        else:
            model, sizes = workloads.build_ssd(archi)
        model._ensure_params()
        if torch.distributed.is_initialized():
            dp = djdist.DataParallel(model)
            dp.broadcast_weights(0)
        # rank r draws its own shard of the global batch (data seed 1234 + r)
        if not archi.startswith("cls:"):
            x, y = workloads.synthetic_batch(archi, sizes, batch, seed=1234 + rank, fast=True)
        plan = model._plan(batch, True, True)
        model._upload(plan, x, y)
        torch.cuda.synchronize()

        def barrier():
            if torch.distributed.is_initialized():
                torch.distributed.barrier()

        for _ in range(warmup):
            model.run_train_step(plan)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.run_train_step(plan)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        loss = model._loss_value(plan)
        return elapsed, loss, model, plan
    finally:
        K.set_floatx(prev)


def roofline_of(archi, floatx, batch, value, world, model, plan):
    """The `roofline` object of one workload: whole-step MFMA fraction, and for world == 1 the implicit-GEMM family
    timed launch by launch with HIP events (one extra step)."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    gflop = workloads.TRAIN_GFLOP_PER_IMAGE[archi] if archi in workloads.TRAIN_GFLOP_PER_IMAGE else plan_gflop_per_image(plan, batch)
    per_gpu_tflops = value * gflop / 1e3 / world
    roof = {"bound": "mfma", "achieved": per_gpu_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": per_gpu_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
            "note": "whole-step algorithmic conv FLOPs (SURVEY 8(d) table) / step time, per GPU"}
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for rnd in ("r03", "r02"):
        traffic_file = os.path.join(prof, "%s_igemm_traffic%s.json" % (rnd, {"float32": "", "float32_mfma": "_mfma", "float32x3": "_x3", "float32x6": "_x6"}.get(floatx, "_f16")))
        if os.path.exists(traffic_file) and archi == "deconv" and batch == 32:
            # offline rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_f_hbm_traffic_pmc.md), bytes per launch
            with open(traffic_file) as f:
                roof["traffic"] = json.load(f)["bytes_per_launch"]
            roof["traffic_source"] = os.path.basename(traffic_file) + " (separate --pmc passes of this workload, not this run)"
            break
    if world == 1:
        prev = K.floatx()
        K.set_floatx(floatx)
        try:
            k = measure_conv_kernels(model, plan)
        finally:
            K.set_floatx(prev)
        ktf = k["flop"] / (k["total_ms"] * 1e-3) / 1e12
        roof["dominant_kernel"] = {"name": "dj_igemm_kernel (conv fwd/dgrad/wgrad)", "launches_per_step": k["launches"],
                                   "avg_launch_us": 1e3 * k["total_ms"] / k["launches"],
                                   "event_pair_overhead_us": k["empty_pair_us"],
                                   "avg_launch_us_net_of_event_overhead": 1e3 * k["total_ms"] / k["launches"] - k["empty_pair_us"],
                                   "ms_per_step": k["total_ms"], "achieved": ktf, "frac": ktf / PEAK_FP32_MFMA_TFLOPS,
                                   "algorithmic_gflop_per_step": k["flop"] / 1e9,
                                   "algorithmic_gbyte_per_step": k["bytes"] / 1e9,
                                   "algorithmic_gbyte_per_s": k["bytes"] / (k["total_ms"] * 1e-3) / 1e9}
        if floatx in ("float32x3", "float32x6"):
            # three / six bf16 MFMAs per fp32 product: the matrix pipe executes 3x / 6x the algorithmic FLOPs, at the bf16 rate
            nx = 3 if floatx == "float32x3" else 6
            roof.update({"achieved": nx * per_gpu_tflops, "peak": PEAK_F16_MFMA_TFLOPS, "frac": nx * per_gpu_tflops / PEAK_F16_MFMA_TFLOPS,
                         "note": "whole-step conv FLOPs x%d (each fp32 product = that many bf16 MFMA products) / step time, per GPU, " % nx +
                                 "against the dense bf16 MFMA peak; `useful_fp32` is the algorithmic rate next to the fp32 "
                                 "MFMA peak the exact mode is bounded by",
                         "useful_fp32": {"achieved": per_gpu_tflops, "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                                         "ratio": per_gpu_tflops / PEAK_FP32_MFMA_TFLOPS, "kernel_family_tflops": ktf}})
            roof["dominant_kernel"]["frac"] = nx * ktf / PEAK_F16_MFMA_TFLOPS
            roof["dominant_kernel"]["achieved"] = nx * ktf
        elif floatx == "float32":
            # which launches of the plan run the split-bf16 kernels (upper half of the configuration indices)
            from jpeg_detection_resnet_ssd_amd import _lib, engine
            names = [n for n, _ in _lib.ConvDesc._fields_][:15]
            prev = K.floatx()
            K.set_floatx(floatx)
            try:
                n_mfma = _lib.load().dj_conv2d_tune_configs() // 2
                f_split = f_all = 0.0
                for direction, desc, _fn in plan.conv_calls:
                    t = engine._TUNED._cur().get((direction,) + tuple(getattr(desc, n) for n in names))
                    f_all += conv_flops(desc)
                    if t is not None and t[1] >= n_mfma:
                        f_split += conv_flops(desc)
            finally:
                K.set_floatx(prev)
            share = f_split / max(f_all, 1.0)
            p_split = PEAK_F16_MFMA_TFLOPS / 6.0
            p_eff = 1.0 / ((1.0 - share) / PEAK_FP32_MFMA_TFLOPS + share / p_split)
            roof["note"] += ("; `peak` is the dense fp32 MFMA peak (the dtype's peak, as in earlier rounds). %.0f %% of the conv FLOPs "
                             "run on the split-bf16 kernels, whose matrix-pipe ceiling is 2500 / 6 = %.0f TFLOP/s of fp32 work: "
                             "`matrix_pipe` prices the step against the ceiling of this kernel mix" % (100 * share, p_split))
            roof["matrix_pipe"] = {"flop_share_split_bf16x6": share, "peak_split_bf16x6": p_split,
                                   "peak_of_this_mix": p_eff, "frac_of_mix_peak": per_gpu_tflops / p_eff,
                                   "kernel_family_frac_of_mix_peak": ktf / p_eff}
        elif floatx != "float32_mfma":
            # reduced-precision MFMA: the GEMMs are ~16x cheaper, reading / writing the operands is what bounds the
            # family (2.4 TF of arithmetic per GB moved at these shapes against a machine balance of 2500 TF / 8 TB/s =
            # 312 FLOP/B): the roofline of this mode is HBM
            gbps = k["bytes"] / (k["total_ms"] * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                         "note": "implicit-GEMM family: algorithmic bytes (each operand read once, result written once, at "
                                 "the width it has in HBM) / its kernel time; the arithmetic side is in `mfma`",
                         "mfma": {"achieved": per_gpu_tflops, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": per_gpu_tflops / PEAK_F16_MFMA_TFLOPS, "kernel_family_tflops": ktf}})
            roof["dominant_kernel"]["frac"] = gbps / PEAK_HBM_GBPS
    return roof


# the non-headline single-GPU configurations of BASELINE.json, measured by the same run so that the driver's record holds
# them too: config 3 and the two workloads of config 5 -- and the headline workload with fp32 MFMA
# instructions only (what `value` measured in rounds 1-2) and in the float32x3 arithmetic (fp32 tensors, ~4e-6 per GEMM:
# inside the 1e-3 parity bar but not an fp32 arithmetic, hence never the headline)
SECONDARY = [("deconv", "float32_mfma"), ("deconv", "float32x3"), ("ssd_custom", "float32"), ("ssd_custom", "float32x3"),
             ("cls:deconv", "float32"),      # config 2: the ResNet50-DCT classifier, batch 64
             ("deconv", "float16"), ("ssd_custom", "float16"), ("up_sampling", "float16")]


def main(json_out=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--archi", default="deconv", choices=["deconv", "ssd_custom", "up_sampling"])
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (the reference trainer's batch_size)")
    ap.add_argument("--floatx", default="float32", choices=["float32", "float32_mfma", "float16", "bfloat16", "float32x3", "float32x6"],
                    help="conv arithmetic: float32 = fp32 results, fp32 MFMA or split-bf16 (float32x6) kernel per layer (the "
                         "headline); float32_mfma = fp32 MFMA instructions only; float32x6 / float32x3 = the split kernels "
                         "everywhere; float16 / bfloat16 = BASELINE config 5's reduced-precision MFMA with fp32 master weights "
                         "and accumulation (each reported with its own dtype)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` list (config 3 / config 5 workloads)")
    ap.add_argument("--cpu-batch", type=int, default=8, help="batch of the CPU-oracle sample (BASELINE.md: 8)")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU work for the cpu_baseline sample")
    args = ap.parse_args()

    from jpeg_detection_resnet_ssd_amd import dist as djdist
    from jpeg_detection_resnet_ssd_amd import workloads
    # (DJ_BENCH_BACKEND=gloo DJ_BENCH_ONE_CARD=1: rehearsal of the N > 1 command on a one-GPU box -- every rank on card 0,
    # gradients exchanged over gloo; the driver's run uses neither: RCCL, one card per rank)
    rank, world, local = djdist.init_from_env(os.environ.get("DJ_BENCH_BACKEND"))      # before any GPU call
    if os.environ.get("DJ_BENCH_ONE_CARD") == "1":
        local = 0
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the compute path has no CPU fallback")
    torch.cuda.set_device(local)

    elapsed, loss, model, plan = run_workload(args.archi, args.floatx, args.batch, args.steps, args.warmup, rank, world)
    print("rank %d: %d steps in %.3f s, last loss %.4f" % (rank, args.steps, elapsed, loss), file=sys.stderr, flush=True)

    if rank != 0:
        return
    images = world * args.batch * args.steps
    value = images / elapsed
    gflop = workloads.TRAIN_GFLOP_PER_IMAGE[args.archi]
    out = {
        "metric": "images/sec (train) ResNet50-DCT-SSD300",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NAME[args.floatx], "arithmetic": ARITHMETIC[args.floatx],
        "data": "synthetic",
        "config": {"workload": "SSD300 ResNet50-DCT '%s' archi, %d images/GPU, 300x300 JPEG-DCT inputs "
                               "(Y 38x38x64 + chroma 19x19), fwd+loss+bwd+SGD(+RCCL all-reduce); batch and encoded targets "
                               "resident in HBM: host->device upload, target encoding and the per-step loss read-back of "
                               "fit_generator are outside the timed region (DESIGN.md section 6 has those rates)"
                               % (args.archi, args.batch),
                   "archi": args.archi, "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                   "train_gflop_per_image": gflop, "last_loss": loss,
                   "multi_gpu": ("data parallel over RCCL, %d ranks" % world) if world > 1 else
                                "1 GPU; the RCCL exchange has only ever run on a 1-rank communicator (no multi-GPU box "
                                "was available to the builder): no scaling curve has been measured"},
    }
    out["roofline"] = roofline_of(args.archi, args.floatx, args.batch, value, world, model, plan)
    del model, plan
    torch.cuda.empty_cache()
    if world == 1 and not args.no_secondary and (args.archi, args.floatx, args.batch) == ("deconv", "float32", 32):
        sec = []
        for archi, floatx in SECONDARY:
            steps = min(args.steps, 20)
            b2 = 64 if archi.startswith("cls:") else args.batch
            el, ls, m2, p2 = run_workload(archi, floatx, b2, steps, min(args.warmup, 3))
            v = b2 * steps / el
            if archi.startswith("cls:"):
                what = ("ResNet50-DCT classifier '%s' archi (BASELINE config 2), %d images/GPU, 224x224 JPEG-DCT inputs, "
                        "categorical cross-entropy, Nesterov SGD" % (archi[4:], b2))
                gf = plan_gflop_per_image(p2, b2)
            else:
                what, gf = "SSD300 ResNet50-DCT '%s' archi, %d images/GPU" % (archi, b2), workloads.TRAIN_GFLOP_PER_IMAGE[archi]
            sec.append({"config": {"workload": what, "archi": archi, "floatx": floatx, "train_gflop_per_image": gf,
                                   "last_loss": ls},
                        "value": v, "unit": "images/sec", "ms_per_step": 1e3 * el / steps, "steps": steps,
                        "dtype": DTYPE_NAME[floatx], "arithmetic": ARITHMETIC[floatx],
                        "roofline": roofline_of(archi, floatx, b2, v, 1, m2, p2)})
            print("secondary %s %s: %.1f img/s (%.2f ms/step)" % (archi, floatx, v, 1e3 * el / steps), file=sys.stderr,
                  flush=True)
            del m2, p2
            torch.cuda.empty_cache()
        out["secondary"] = sec
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.archi, args.cpu_batch, args.cpu_budget)
    print(json.dumps(out), file=json_out or sys.stdout, flush=True)


if __name__ == "__main__":
    # stdout carries exactly one JSON line: libraries that print banners on fd 1 (RCCL's version block at communicator
    # creation) are sent to stderr while the benchmark runs
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)
    _json_out = os.fdopen(_real_stdout, "w")
    main(_json_out)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()

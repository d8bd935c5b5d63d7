"""float32x3 against the exact-fp32 kernels on representative layers of the SSD300 step: best tile variant (and split-K for
the weight gradient) of each mode, TFLOP/s of useful fp32 work.   python tools/x3_micro.py [fwd|dgrad|wgrad|all] [float32x3|float32x6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import _lib, kernels as K
from jpeg_detection_resnet_ssd_amd.keras import backend as KB
SHAPES = [  # name, B,H,W,Cin,Cout,k
    ("1x1 38x38 256->1024", 32, 38, 38, 256, 1024, 1),
    ("1x1 38x38 1024->256", 32, 38, 38, 1024, 256, 1),
    ("1x1 38x38 512->128", 32, 38, 38, 512, 128, 1),
    ("3x3 38x38 128->128", 32, 38, 38, 128, 128, 3),
    ("3x3 38x38 256->256", 32, 38, 38, 256, 256, 3),
    ("1x1 19x19 1024->256", 32, 19, 19, 1024, 256, 1),
    ("3x3 19x19 256->256", 32, 19, 19, 256, 256, 3),
    ("3x3 10x10 512->512", 32, 10, 10, 512, 512, 3),
    ("3x3 19x19 1024->1024 fc6-like", 32, 19, 19, 1024, 1024, 3),
]
lib = _lib.load()
dev = torch.device("cuda:0")
ncfg = lib.dj_conv2d_tune_configs()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
X = sys.argv[2] if len(sys.argv) > 2 else "float32x3"      # or float32x6


def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {}
for name, b, h, w, ci, co, k in SHAPES:
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), "same", (1, 1))
    x = torch.randn(b, h, w, ci, device=dev)
    dy = torch.randn(b, h, w, co, device=dev) * 1e-3
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev)
    wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    y = torch.empty(b, h, w, co, device=dev)
    dx = torch.empty(b, h, w, ci, device=dev)
    dw = torch.zeros(k, k, ci, co, device=dev)
    stats = torch.zeros(K.conv2d_stats_rows(desc), 2, co, device=dev)
    flop = 2.0 * b * h * w * co * k * k * ci
    runs = {
        "fwd": (4, lambda: K.conv2d_fwd(desc, x, wt, None, y, sc, sh, True, False, stats), (1,)),
        "dgrad": (1, lambda: K.conv2d_dgrad(desc, dy, wt, dx), (1,)),
        "wgrad": (2, lambda: K.conv2d_wgrad(desc, x, dy, dw, sc, sh, True, dw_zeroed=True), (1, 2, 4, 7, 14, 28)),
    }
    for dname, (direction, fn, splits) in runs.items():
        if which not in ("all", dname):
            continue
        best = {}
        for mode in ("float32", X):
            KB.set_floatx(mode)
            res = []
            for cfg in range(ncfg):
                for sp in splits:
                    if lib.dj_conv2d_tune_set(direction, desc, cfg, sp) != 0:
                        continue
                    try:
                        res.append((timeit(fn), cfg, sp))
                    except Exception:
                        pass
            lib.dj_conv2d_tune_set(direction, desc, -1, 1)
            best[mode] = min(res)
            t = tot.setdefault((dname, mode), [0.0, 0.0]); t[0] += best[mode][0]; t[1] += flop
        KB.set_floatx("float32")
        a, c = best["float32"], best[X]
        print("%-32s %-6s fp32 %.3f ms %6.1f TF (cfg %d/%d) | split %.3f ms %6.1f TF (cfg %d/%d)  x%.2f"
              % (name, dname, a[0], flop / a[0] / 1e9, a[1], a[2], c[0], flop / c[0] / 1e9, c[1], c[2], a[0] / c[0]), flush=True)
for (dname, mode), (t, f) in sorted(tot.items()):
    print("%s %s: %.3f ms, %.1f TF" % (dname, mode, t, f / t / 1e9))

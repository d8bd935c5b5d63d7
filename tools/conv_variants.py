"""One conv direction (fwd | dgrad | wgrad): every tile variant x split factor on representative geometries of the bench
plan, each checked against the first variant's result.   python tools/conv_variants.py dgrad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import _lib, kernels as K

SHAPES = [  # name, B,H,W,Cin,Cout,k,s,pad,dil
    ("3x3 38x38 256->256", 32, 38, 38, 256, 256, 3, 1, "same", 1),
    ("1x1 38x38 256->1024", 32, 38, 38, 256, 1024, 1, 1, "valid", 1),
    ("1x1 38x38 1024->256", 32, 38, 38, 1024, 256, 1, 1, "valid", 1),
    ("1x1 38x38 192->256", 32, 38, 38, 192, 256, 1, 1, "valid", 1),
    ("2x2 38x38 256->256", 32, 38, 38, 256, 256, 2, 1, "same", 1),
    ("3x3 19x19 256->256", 32, 19, 19, 256, 256, 3, 1, "same", 1),
    ("1x1 19x19 1024->256", 32, 19, 19, 1024, 256, 1, 1, "valid", 1),
    ("1x1 19x19 256->1024", 32, 19, 19, 256, 1024, 1, 1, "valid", 1),
    ("3x3 10x10 512->512", 32, 10, 10, 512, 512, 3, 1, "same", 1),
    ("1x1 10x10 2048->512", 32, 10, 10, 2048, 512, 1, 1, "valid", 1),
    ("fc6 10x10 2048->1024 d6", 32, 10, 10, 2048, 1024, 3, 1, "same", 6),
    ("fc7 10x10 1024->1024", 32, 10, 10, 1024, 1024, 1, 1, "valid", 1),
    ("conf 19x19 1024->126", 32, 10, 10, 1024, 126, 3, 1, "same", 1),
    ("conf 38x38 64->84", 32, 38, 38, 64, 84, 3, 1, "same", 1),
]
lib = _lib.load()
if len(sys.argv) > 2:      # arithmetic mode: float16 | bfloat16 (the tile names then read: *_P / *_PK2 = 64-deep K-steps)
    from jpeg_detection_resnet_ssd_amd.keras import backend as KB
    KB.set_floatx(sys.argv[2])
dev = torch.device("cuda:0")
ncfg = lib.dj_conv2d_tune_configs()
DIR = {"fwd": 0, "dgrad": 1, "wgrad": 2}[sys.argv[1] if len(sys.argv) > 1 else "wgrad"]
names = ["128x128", "128x64", "64x64", "128x32", "128x128_S1", "128x64_S1", "64x64_S1", "64x64_S1P", "128x64_S1P", "128x128_P",
         "128x64_P", "64x64_P", "64x64_PK2", "128x64_PK2"]


def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, b, h, w, ci, co, k, s, pad, d in SHAPES:
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (d, d))
    x = torch.randn(b, h, w, ci, device=dev)
    dy = torch.randn(b, desc.out_h, desc.out_w, co, device=dev)
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev)
    dw = torch.zeros(k, k, ci, co, device=dev)
    wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    dx = torch.zeros(b, h, w, ci, device=dev)
    y = torch.zeros(b, desc.out_h, desc.out_w, co, device=dev)
    flop = 2.0 * b * desc.out_h * desc.out_w * co * k * k * ci
    kk = b * desc.out_h * desc.out_w
    ref, rows = None, []
    for cfg in range(ncfg):
        direct = False
        for sp in ((1, 2, 4, 7, 14, 28, 56) if DIR == 2 else ((1,) if direct else (1, 2, 4))):
            if DIR == 2 and sp > 1 and kk // sp < 256:
                continue
            if DIR != 2 and sp > 1 and (k * k * (co if DIR == 1 else ci)) // sp < 256:
                continue
            _lib.check(lib.dj_conv2d_tune_set(DIR, desc, cfg, sp), "tune_set")
            if DIR == 2:
                fn = lambda: K.conv2d_wgrad(desc, x, dy, dw, sc, sh, True)
                out = dw
            elif DIR == 1:
                fn = lambda: K.conv2d_dgrad(desc, dy, wt, dx)
                out = dx
            else:
                fn = lambda: K.conv2d_fwd(desc, x, wt, None, y, sc, sh, True)
                out = y
            t = timeit(fn)
            got = out.clone()
            if ref is None:
                ref = got
            err = float((got - ref).abs().max() / ref.abs().max())
            rows.append((t, names[cfg], sp, err))
    rows.sort()
    best_old = min(r for r in rows if not r[1][:2] in ("WD", "DD"))
    new = [r for r in rows if r[1][:2] in ("WD", "DD")]
    best_new = min(new) if new else (float("nan"), "-", 0, 0.0)
    print("%-26s %6.1f GFLOP | old best %-11s sp %2d %7.3f ms %6.1f TF | direct best %-7s sp %2d %7.3f ms %6.1f TF | max err %.1e"
          % (name, flop / 1e9, best_old[1], best_old[2], best_old[0], flop / best_old[0] / 1e9, best_new[1], best_new[2],
             best_new[0], flop / best_new[0] / 1e9, max(r[3] for r in rows)), flush=True)
    for t, n, sp, err in rows[:(14 if len(sys.argv) > 2 else 6)]:
        print("      %-11s sp %2d %7.3f ms %6.1f TF err %.1e" % (n, sp, t, flop / t / 1e9, err))
    _lib.check(lib.dj_conv2d_tune_set(DIR, desc, -1, 1), "tune_set")

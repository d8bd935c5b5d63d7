"""float16 mode, tensors as the plan holds them (fp16 activations, bf16 gradients, 16-bit weight shadows): every tile variant
of the three GEMM directions on representative layers of the SSD300 step, with the bytes each launch has to move and the
rate that is of the HBM roofline.   python tools/h16_micro.py [fwd|dgrad|wgrad|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import _lib, kernels as K
from jpeg_detection_resnet_ssd_amd.keras import backend as KB
KB.set_floatx("float16")
SHAPES = [  # name, B,H,W,Cin,Cout,k
    ("1x1 38x38 256->1024", 32, 38, 38, 256, 1024, 1),
    ("1x1 38x38 1024->256", 32, 38, 38, 1024, 256, 1),
    ("1x1 38x38 128->512", 32, 38, 38, 128, 512, 1),
    ("1x1 38x38 512->128", 32, 38, 38, 512, 128, 1),
    ("3x3 38x38 128->128", 32, 38, 38, 128, 128, 3),
    ("3x3 38x38 256->256", 32, 38, 38, 256, 256, 3),
    ("1x1 19x19 256->1024", 32, 19, 19, 256, 1024, 1),
    ("1x1 19x19 1024->256", 32, 19, 19, 1024, 256, 1),
    ("3x3 19x19 256->256", 32, 19, 19, 256, 256, 3),
]
lib = _lib.load()
dev = torch.device("cuda:0")
ncfg = lib.dj_conv2d_tune_configs()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
names = {0: "128x128/32", 4: "128x128/32/pf2", 9: "128x128/64", 13: "128x128/64/pf2", 1: "128x64/32", 5: "128x64/32/pf2",
         10: "128x64/64", 8: "128x64/64/pf2", 2: "64x64/32", 3: "64x64/32/pf2", 11: "64x64/64", 7: "64x64/64/pf2"}


def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, b, h, w, ci, co, k in SHAPES:
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), "same", (1, 1))
    x = torch.randn(b, h, w, ci, device=dev).half()
    z = torch.randn(b, h, w, ci, device=dev).half()
    dy = (torch.randn(b, h, w, co, device=dev) * 1e-3).bfloat16()
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev)
    mean, invstd = torch.randn(ci, device=dev) * 0.1, torch.rand(ci, device=dev) + 0.5
    wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    w16, wbf = wt.half(), wt.bfloat16()
    y = torch.empty(b, h, w, co, device=dev, dtype=torch.float16)
    dx = torch.empty(b, h, w, ci, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(k, k, ci, co, device=dev)
    stats = torch.zeros(K.conv2d_stats_rows(desc), 2, co, device=dev)
    part = torch.zeros((b * h * w + 63) // 64, 2, ci, device=dev)
    n_in, n_out, n_w = b * h * w * ci, b * h * w * co, k * k * ci * co
    flop = 2.0 * n_out * k * k * ci
    runs = {
        "fwd": (4, lambda: K.conv2d_fwd(desc, x, w16, None, y, sc, sh, True, False, stats), 2 * n_in + 2 * n_out + 2 * n_w, (1,)),
        "dgrad": (9, lambda: K.conv2d_dgrad_bnbwd(desc, dy, wbf, dx, z, mean, invstd, sc, sh, part), 2 * n_out + 4 * n_in + 2 * n_w, (1,)),
        "wgrad": (2, lambda: K.conv2d_wgrad(desc, x, dy, dw, sc, sh, True, dw_zeroed=True), 2 * n_in + 2 * n_out + 4 * n_w, (1, 2, 4, 7, 14, 28)),
    }
    for dname, (direction, fn, nbytes, splits) in runs.items():
        if which not in ("all", dname):
            continue
        rows = []
        for cfg in sorted(names):
            for sp in splits:
                _lib.check(lib.dj_conv2d_tune_set(direction, desc, cfg, sp), "tune_set")
                rows.append((timeit(fn), names[cfg], sp))
        _lib.check(lib.dj_conv2d_tune_set(direction, desc, -1, 1), "tune_set")
        rows.sort()
        t = rows[0][0]
        print("%-22s %-5s %6.1f GFLOP %6.1f MB | best %-15s sp %2d %7.1f us %6.0f TF %5.0f GB/s | %s" % (
            name, dname, flop / 1e9, nbytes / 1e6, rows[0][1], rows[0][2], t * 1e3, flop / t / 1e9, nbytes / t / 1e6,
            "  ".join("%s%s %.0f" % (n, "" if sp == 1 else "/s%d" % sp, tt * 1e3) for tt, n, sp in rows[1:6])), flush=True)

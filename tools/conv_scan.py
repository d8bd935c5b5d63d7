"""TFLOP/s of one conv geometry as the batch (number of tiles) varies, per tile variant: separates wave-quantisation
loss from per-tile efficiency.  args: H W Cin Cout k mode"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import kernels as K, _lib
h, w, ci, co, k = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6]
dev = torch.device("cuda:0")
lib = _lib.load()
names = ["128x128", "128x64", "64x64", "128x32", "128x128_S1", "128x64_S1", "64x64_S1", "64x64_S1P", "128x64_S1P"]
for b in (16, 32, 45, 64, 90, 128):
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), "same", (1, 1))
    x = torch.randn(b, h, w, ci, device=dev); wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    y = torch.empty(b, h, w, co, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(wt)
    flop = 2.0 * b * h * w * co * k * k * ci
    out = []
    for cfg in (1, 2, 6, 7, 8):
        _lib.check(lib.dj_conv2d_tune_set({"fwd": 0, "dgrad": 1, "wgrad": 2}[mode], desc, cfg, 1), "tune_set")
        fn = {"fwd": lambda: K.conv2d_fwd(desc, x, wt, None, y), "dgrad": lambda: K.conv2d_dgrad(desc, dy, wt, dx),
              "wgrad": lambda: K.conv2d_wgrad(desc, x, dy, dw)}[mode]
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        out.append("%s %5.1f" % (names[cfg], flop / t / 1e9))
    print("B=%3d M=%6d  %s TF" % (b, b * h * w, " | ".join(out)), flush=True)

"""Time of a 1x1 conv GEMM (M = B*H*W, N = Cout) as K = Cin grows: intercept = per-launch fixed cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import kernels as K, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 6
b = 32
h = w = int(sys.argv[3]) if len(sys.argv) > 3 else 19
co = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
lib = _lib.load()
for ci in (32, 64, 128, 256, 512, 1024, 2048):
    desc = K.make_conv_desc(b, h, w, ci, co, (1, 1), (1, 1), "valid", (1, 1))
    x = torch.randn(b, h, w, ci, device=dev); wt = torch.randn(1, 1, ci, co, device=dev) * 0.05
    y = torch.empty(b, h, w, co, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x)
    _lib.check(lib.dj_conv2d_tune_set({"fwd": 0, "dgrad": 1}[mode], desc, cfg, 1), "tune_set")
    fn = (lambda: K.conv2d_fwd(desc, x, wt, None, y)) if mode == "fwd" else (lambda: K.conv2d_dgrad(desc, dy, wt, dx))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    kk = ci if mode == "fwd" else co
    flop = 2.0 * b * h * w * co * ci
    print("%s cfg %d Cin %5d: %7.1f us  %6.1f TF" % (mode, cfg, ci, t, flop / t / 1e6), flush=True)

"""MFMA efficiency vs workgroups per CU: 1x1 conv GEMM with M = 16384 (256 row tiles of 64), N = 64*j -> 256*j tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import kernels as K, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ci = 2304
for rows, co in ((8192, 64), (16384, 64), (16384, 128), (16384, 192), (16384, 256), (16384, 320), (16384, 384), (16384, 512), (16384, 1024)):
    b, h, w = rows // 256, 16, 16
    desc = K.make_conv_desc(b, h, w, ci, co, (1, 1), (1, 1), "valid", (1, 1))
    x = torch.randn(b, h, w, ci, device=dev); wt = torch.randn(1, 1, ci, co, device=dev) * 0.05
    y = torch.empty(b, h, w, co, device=dev)
    _lib.check(lib.dj_conv2d_tune_set(0, desc, cfg, 1), "tune_set")
    fn = lambda: K.conv2d_fwd(desc, x, wt, None, y)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    tiles = (rows // 64) * (co // 64)
    flop = 2.0 * rows * co * ci
    print("cfg %d tiles %5d (%.2f per CU): %7.1f us %6.1f TF" % (cfg, tiles, tiles / 256.0, t, flop / t / 1e6), flush=True)

"""Micro-benchmark of the implicit-GEMM conv on representative SSD300 layer shapes (B=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import kernels as K

SHAPES = [  # name, B,H,W,Cin,Cout,k,s,pad,dil
    ("3x3 38x38 256->256", 32, 38, 38, 256, 256, 3, 1, "same", 1),
    ("3x3 38x38 128->128", 32, 38, 38, 128, 128, 3, 1, "same", 1),
    ("1x1 38x38 256->1024", 32, 38, 38, 256, 1024, 1, 1, "valid", 1),
    ("1x1 38x38 1024->256", 32, 38, 38, 1024, 256, 1, 1, "valid", 1),
    ("2x2 38x38 256->256", 32, 38, 38, 256, 256, 2, 1, "same", 1),
    ("3x3 19x19 256->256", 32, 19, 19, 256, 256, 3, 1, "same", 1),
    ("1x1 19x19 1024->256", 32, 19, 19, 1024, 256, 1, 1, "valid", 1),
    ("3x3 10x10 512->512", 32, 10, 10, 512, 512, 3, 1, "same", 1),
    ("fc6 10x10 2048->1024 d6", 32, 10, 10, 2048, 1024, 3, 1, "same", 6),
    ("conf 38x38 384->84", 32, 38, 38, 384, 84, 3, 1, "same", 1),
]

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device("cuda:0")
for name, b, h, w, ci, co, k, s, pad, d in SHAPES:
    desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (d, d))
    x = torch.randn(b, h, w, ci, device=dev); wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    bias = torch.randn(co, device=dev)
    y = torch.empty(b, desc.out_h, desc.out_w, co, device=dev); dy = torch.randn_like(y)
    dx = torch.empty_like(x); dw = torch.empty_like(wt)
    flop = 2.0 * b * desc.out_h * desc.out_w * co * k * k * ci
    tf = timeit(lambda: K.conv2d_fwd(desc, x, wt, bias, y))
    td = timeit(lambda: K.conv2d_dgrad(desc, dy, wt, dx))
    tw = timeit(lambda: K.conv2d_wgrad(desc, x, dy, dw))
    print("%-28s GFLOP %7.2f | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" % (
        name, flop / 1e9, tf, flop / tf / 1e9, td, flop / td / 1e9, tw, flop / tw / 1e9), flush=True)

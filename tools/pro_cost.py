"""What the BatchNormalization + ReLU prologue costs a convolution: forward and weight-gradient launches of a few
geometries of the SSD step, with and without the per-channel affine on the gathered operand."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import kernels as K
dev = torch.device("cuda:0")
b = 32
def t_us(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = {"fwd": [0, 0], "wgrad": [0, 0]}
for (h, ci, co, k, cnt) in ((38, 256, 1024, 1, 3), (19, 256, 256, 3, 6), (38, 128, 128, 3, 4), (19, 1024, 256, 1, 5),
                            (10, 512, 512, 3, 3), (38, 1024, 256, 1, 2), (19, 256, 1024, 1, 6), (38, 128, 512, 1, 4)):
    desc = K.make_conv_desc(b, h, h, ci, co, (k, k), (1, 1), "same", (1, 1))
    x = torch.randn(b, h, h, ci, device=dev); w = torch.randn(k, k, ci, co, device=dev) * 0.05
    y = torch.empty(b, h, h, co, device=dev); dy = torch.randn_like(y); dw = torch.empty_like(w)
    sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev)
    f0 = t_us(lambda: K.conv2d_fwd(desc, x, w, None, y))
    f1 = t_us(lambda: K.conv2d_fwd(desc, x, w, None, y, sc, sh, True))
    g0 = t_us(lambda: K.conv2d_wgrad(desc, x, dy, dw))
    g1 = t_us(lambda: K.conv2d_wgrad(desc, x, dy, dw, sc, sh, True))
    tot["fwd"][0] += f0 * cnt; tot["fwd"][1] += f1 * cnt; tot["wgrad"][0] += g0 * cnt; tot["wgrad"][1] += g1 * cnt
    act_mb = b * h * h * ci * 4 / 1e6
    print("%2dx%2d %4d->%4d k%d: fwd %6.1f -> %6.1f us (+%4.1f %%)   wgrad %6.1f -> %6.1f us (+%4.1f %%)   activation %5.1f MB"
          % (h, h, ci, co, k, f0, f1, 100 * (f1 / f0 - 1), g0, g1, 100 * (g1 / g0 - 1), act_mb), flush=True)
for d in tot:
    print("%s weighted by launches per step: %.0f -> %.0f us" % (d, tot[d][0], tot[d][1]))

"""Run one conv shape repeatedly (for rocprofv3 --pmc studies). args: B H W Cin Cout k s pad dil [mode] [prologue]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import kernels as K
b, h, w, ci, co, k, s = [int(v) for v in sys.argv[1:8]]
pad, d = sys.argv[8], int(sys.argv[9])
mode = sys.argv[10] if len(sys.argv) > 10 else "fwd"
pro = len(sys.argv) > 11 and sys.argv[11] == "pro"
dev = torch.device("cuda:0")
desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (s, s), pad, (d, d))
x = torch.randn(b, h, w, ci, device=dev); wt = torch.randn(k, k, ci, co, device=dev) * 0.05; bias = torch.randn(co, device=dev)
y = torch.empty(b, desc.out_h, desc.out_w, co, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(wt)
sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev)
stats = torch.empty(K.conv2d_stats_rows(desc), 2, co, device=dev)
if os.environ.get("DJ_CFG"):   # "cfg,splits": pin the tile variant (dj_conv2d_tune_set)
    from jpeg_detection_resnet_ssd_amd import _lib
    cfg, sp = [int(v) for v in os.environ["DJ_CFG"].split(",")]
    direction = {"fwd": 4 if pro else 0, "dgrad": 1, "wgrad": 2}[mode]
    _lib.check(_lib.load().dj_conv2d_tune_set(direction, desc, cfg, sp), "tune_set")
for _ in range(5):
    if mode == "fwd":
        K.conv2d_fwd(desc, x, wt, bias, y, sc if pro else None, sh if pro else None, pro, False, stats if pro else None)
    elif mode == "dgrad":
        K.conv2d_dgrad(desc, dy, wt, dx)
    else:
        K.conv2d_wgrad(desc, x, dy, dw, sc if pro else None, sh if pro else None, pro)
torch.cuda.synchronize()

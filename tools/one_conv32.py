"""One conv on fp32 tensors under a chosen arithmetic mode and configuration index, run a few times (rocprofv3 --pmc studies).
args: B H W Cin Cout k mode(fwd|fwd_plain|dgrad|wgrad) cfg [floatx] [splits]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import _lib, kernels as K
from jpeg_detection_resnet_ssd_amd.keras import backend as KB
b, h, w, ci, co, k = [int(v) for v in sys.argv[1:7]]
mode, cfg = sys.argv[7], int(sys.argv[8])
KB.set_floatx(sys.argv[9] if len(sys.argv) > 9 else "float32")
splits = int(sys.argv[10]) if len(sys.argv) > 10 else 1
dev = torch.device("cuda:0")
desc = K.make_conv_desc(b, h, w, ci, co, (k, k), (1, 1), "same", (1, 1))
x = torch.randn(b, h, w, ci, device=dev)
wt = torch.randn(k, k, ci, co, device=dev) * 0.05
y = torch.empty(b, h, w, co, device=dev)
dy = torch.randn(b, h, w, co, device=dev) * 1e-3
dx = torch.empty(b, h, w, ci, device=dev)
dw = torch.zeros(k, k, ci, co, device=dev)
sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev)
stats = torch.zeros(K.conv2d_stats_rows(desc), 2, co, device=dev)
lib = _lib.load()
for d in (0, 4, 1, 2):
    _lib.check(lib.dj_conv2d_tune_set(d, desc, cfg, splits if d == 2 else 1), "tune_set")
for _ in range(5):
    if mode == "fwd":
        K.conv2d_fwd(desc, x, wt, None, y, sc, sh, True, False, stats)
    elif mode == "fwd_plain":
        K.conv2d_fwd(desc, x, wt, None, y)
    elif mode == "dgrad":
        K.conv2d_dgrad(desc, dy, wt, dx)
    else:
        K.conv2d_wgrad(desc, x, dy, dw, sc, sh, True, dw_zeroed=True)
torch.cuda.synchronize()

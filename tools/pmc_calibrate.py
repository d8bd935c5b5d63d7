"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads on gfx950; "other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern").  Each launch copies one tensor larger than the
256 MiB Infinity Cache once:
    dj_copy2d        fp32 -> fp32   16-byte loads and stores per lane (the fp32 kernels' access width)
    dj_copy2d_t      fp16 -> fp16    8-byte loads and stores per lane (the float16 mode's 4-element pieces of 16-bit tensors)
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; tools/pmc_traffic.py reads the factors from the rows of
these two kernels (bytes known / bytes reported)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd.engine import call
rows, cols = 1 << 18, 1024                      # 268 M elements: 1 GiB as fp32, 512 MiB as fp16
a32 = torch.randn(rows, cols, device="cuda")
b32 = torch.empty_like(a32)
a16 = torch.randn(rows, cols, device="cuda", dtype=torch.float16)
b16 = torch.empty_like(a16)
for _ in range(3):
    call("dj_copy2d", a32, cols, b32, cols, rows, cols, 0)
    call("dj_copy2d_t", a16, 1, cols, b16, 1, cols, rows, cols, 0)
torch.cuda.synchronize()
print("copied %d MiB (fp32) and %d MiB (fp16) three times each" % (a32.numel() * 4 >> 20, a16.numel() * 2 >> 20))

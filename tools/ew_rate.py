"""Achieved HBM rate of the BatchNormalization-backward passes (column reduce: 2 tensors read; apply: 2 read + 1 written)
at the tensor sizes of the SSD step, next to a plain device copy of the same bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd.engine import call, query
dev = torch.device("cuda:0")
def t_us(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# rotate over several buffers so that the 256 MB L2/MALL does not serve the re-reads
for rows, c in ((46208, 128), (46208, 256), (46208, 512), (46208, 1024), (11552, 256), (11552, 1024), (3200, 512), (3200, 2048)):
    nbuf = max(2, int(1.2e9 // (rows * c * 4 * 3)))
    dy = [torch.randn(rows, c, device=dev) for _ in range(nbuf)]
    z = [torch.randn(rows, c, device=dev) for _ in range(nbuf)]
    dz = [torch.empty(rows, c, device=dev) for _ in range(nbuf)]
    mean, invstd, scale, shift, gamma = (torch.randn(c, device=dev) for _ in range(5))
    k0, k1, k2, dg, db = (torch.randn(c, device=dev) for _ in range(5))
    nr = query("dj_reduce_rows", rows)
    part = torch.empty(nr, 2, c, device=dev)
    it = [0]
    def red():
        i = it[0] = (it[0] + 1) % nbuf
        call("dj_bn_bwd_reduce", dy[i], c, z[i], c, None, 0, mean, invstd, scale, shift, 2, rows, c, part)
    def app():
        i = it[0] = (it[0] + 1) % nbuf
        call("dj_bn_bwd_apply", dy[i], c, z[i], c, None, 0, scale, shift, 2, k0, k1, k2, dz[i], c, rows, c, None, 0, 0)
    def cpy():
        i = it[0] = (it[0] + 1) % nbuf
        dz[i].copy_(dy[i])
    mb = rows * c * 4 / 1e6
    tr, ta, tc = t_us(red), t_us(app), t_us(cpy)
    print("%6d x %4d (%6.1f MB): reduce %6.1f us %5.2f TB/s | apply %6.1f us %5.2f TB/s | copy %6.1f us %5.2f TB/s"
          % (rows, c, mb, tr, 2 * mb / tr, ta, 3 * mb / ta, tc, 2 * mb / tc), flush=True)
    del dy, z, dz
    torch.cuda.empty_cache()

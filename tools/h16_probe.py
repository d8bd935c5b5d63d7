import os, sys
sys.path.insert(0, os.getcwd())
import torch
from jpeg_detection_resnet_ssd_amd import _lib, kernels as K
from jpeg_detection_resnet_ssd_amd.keras import backend as KB
KB.set_floatx("float16")
lib = _lib.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (b,h,w,ci,co) in ((32,38,38,256,1024),(32,38,38,1024,256),(32,38,38,512,128)):
    desc = K.make_conv_desc(b,h,w,ci,co,(1,1),(1,1),"same",(1,1))
    x = torch.randn(b,h,w,ci,device=dev).half(); wt=(torch.randn(1,1,ci,co,device=dev)*0.05); w16=wt.half()
    y16 = torch.empty(b,h,w,co,device=dev,dtype=torch.float16); y32 = torch.empty(b,h,w,co,device=dev)
    sc, sh = torch.rand(ci,device=dev)+0.5, torch.randn(ci,device=dev)
    stats = torch.zeros(K.conv2d_stats_rows(desc),2,co,device=dev)
    x32 = x.float()
    for cfg in (0, 2):
        for d in (0,4): _lib.check(lib.dj_conv2d_tune_set(d, desc, cfg, 1), "t")
        r = {}
        r["plain x16 w16 y16"] = timeit(lambda: K.conv2d_fwd(desc, x, w16, None, y16))
        r["pro   x16 w16 y16"] = timeit(lambda: K.conv2d_fwd(desc, x, w16, None, y16, sc, sh, True))
        r["stats x16 w16 y16"] = timeit(lambda: K.conv2d_fwd(desc, x, w16, None, y16, None, None, False, False, stats))
        r["pro+st x16 w16 y16"] = timeit(lambda: K.conv2d_fwd(desc, x, w16, None, y16, sc, sh, True, False, stats))
        r["plain x16 w16 y32"] = timeit(lambda: K.conv2d_fwd(desc, x, w16, None, y32))
        r["plain x16 w32 y16"] = timeit(lambda: K.conv2d_fwd(desc, x, wt, None, y16))
        r["plain x32 w32 y32"] = timeit(lambda: K.conv2d_fwd(desc, x32, wt, None, y32))
        print((ci,co), "cfg", cfg, "  ".join("%s %.1f" % kv for kv in r.items()), flush=True)
        for d in (0,4): _lib.check(lib.dj_conv2d_tune_set(d, desc, -1, 1), "t")
# pure copy rate of the same bytes for reference: read x (fp16) write y (fp16)
from jpeg_detection_resnet_ssd_amd.engine import call
src = torch.randn(46208, 1024, device=dev).half(); dst = torch.empty_like(src)
print("copy 94.6 MB fp16 -> fp16: %.1f us" % timeit(lambda: call("dj_copy2d_t", src, 1, 1024, dst, 1, 1024, 46208, 1024, 0)))

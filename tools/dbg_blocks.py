import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import test_blocks_gpu as TB
cuda = torch.device("cuda:0")
orig = TB._check
def chk(out, grads, ref_out, ref_grads, tol=1e-3):
    print("  out rel", TB.rel_err(out, ref_out))
    errs = sorted(((float((grads[k].double()-g).abs().max())/(float(g.abs().max())+1e-30), k, float(g.abs().max())) for k,g in ref_grads.items()), reverse=True)
    for e in errs:
        if e[2] > 1e-6: print("   %.3e %s %.3e" % (e[0], e[1], e[2]))
TB._check = chk
for kind in ["conv_s1_k1", "conv_s2_k3", "identity_k2", "identity_k3"]:
    print(kind)
    TB.test_bottleneck_blocks.__wrapped__(kind, cuda) if hasattr(TB.test_bottleneck_blocks, "__wrapped__") else TB.test_bottleneck_blocks(kind, cuda)

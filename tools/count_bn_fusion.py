"""How many BatchNormalization backward passes of a workload's training plan take their statistics from the producing
input-gradient GEMM (dj_conv2d_nhwc_dgrad_bnbwd) and how many still launch dj_bn_bwd_reduce.  python tools/count_bn_fusion.py deconv 32 [floatx]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import workloads, engine, kernels as Kn
archi = sys.argv[1] if len(sys.argv) > 1 else "deconv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if len(sys.argv) > 3:
    from jpeg_detection_resnet_ssd_amd.keras import backend as KB
    KB.set_floatx(sys.argv[3])
model, sizes = workloads.build_ssd(archi)
x, y = workloads.synthetic_batch(archi, sizes, B, fast=True)
plan = model._plan(B, True, True)
model._upload(plan, x, y)
counts = collections.Counter()
from jpeg_detection_resnet_ssd_amd.keras import layers as L
real_call, real_fused = L.call, Kn.conv2d_dgrad_bnbwd
def call(name, *a):
    counts[name] += 1
    return real_call(name, *a)
def fused(desc, *a):
    counts["dj_conv2d_nhwc_dgrad_bnbwd"] += 1
    counts["fused %dx%d %d->%d k%d" % (desc.in_h, desc.in_w, desc.out_c, desc.in_c, desc.kernel_h)] += 1
    return real_fused(desc, *a)
L.call, Kn.conv2d_dgrad_bnbwd = call, fused
model.run_train_step(plan)
torch.cuda.synchronize()
for k in sorted(counts):
    if "bn_bwd" in k or "fused" in k or "dgrad" in k:
        print("%-50s %d" % (k, counts[k]))

"""Per-launch timing of every implicit-GEMM call in one SSD300 training step (HIP events)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import workloads, kernels as Kn
archi = sys.argv[1] if len(sys.argv) > 1 else "deconv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if len(sys.argv) > 3:      # arithmetic mode: float32 | float16 | bfloat16
    from jpeg_detection_resnet_ssd_amd.keras import backend as KB
    KB.set_floatx(sys.argv[3])
model, sizes = workloads.build_ssd(archi)
x, y = workloads.synthetic_batch(archi, sizes, B, fast=True)
plan = model._plan(B, True, True)
model._upload(plan, x, y)
for _ in range(2): model.run_train_step(plan)
torch.cuda.synchronize()
recs = []
def wrap(name):
    f = getattr(Kn, name)
    def timed(desc, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(desc, *a, **kw); e1.record()
        # bytes of the launch's tensors at the width they have in HBM (positional tensors as in bench.py)
        n_in, n_out = desc.batch * desc.in_h * desc.in_w * desc.in_c, desc.batch * desc.out_h * desc.out_w * desc.out_c
        nw = desc.kernel_h * desc.kernel_w * desc.in_c * desc.out_c
        if name in ("conv2d_fwd", "conv2d_fwd_addrelu"):
            xb, yb, tag = a[0].element_size(), a[3].element_size(), "%s>%s" % (a[0].element_size() * 8, a[3].element_size() * 8)
        elif name in ("conv2d_dgrad", "conv2d_dgrad_bnbwd"):
            xb, yb, tag = a[2].element_size(), a[0].element_size(), "%s>%s" % (a[0].element_size() * 8, a[2].element_size() * 8)
        else:
            xb, yb, tag = a[0].element_size(), a[1].element_size(), "%s,%s" % (a[0].element_size() * 8, a[1].element_size() * 8)
        nbytes = xb * n_in + yb * n_out + 4 * nw
        if name == "conv2d_fwd_addrelu":
            nbytes += (a[6].element_size() + (a[9].element_size() if len(a) > 9 and a[9] is not None else 0)) * n_in
        if name == "conv2d_dgrad_bnbwd":
            nbytes += a[3].element_size() * n_in
        recs.append((name + " " + tag, (desc.batch, desc.in_h, desc.in_w, desc.in_c, desc.out_h, desc.out_c, desc.kernel_h, desc.stride_h, desc.dilation_h), e0, e1,
                     2.0 * desc.batch * desc.out_h * desc.out_w * desc.out_c * desc.kernel_h * desc.kernel_w * desc.in_c, nbytes))
        return r
    setattr(Kn, name, timed)
for n in ("conv2d_fwd", "conv2d_fwd_addrelu", "conv2d_dgrad", "conv2d_dgrad_bnbwd", "conv2d_wgrad"): wrap(n)
plan.side_enabled = False      # one stream: an event pair brackets its launch alone
model.run_train_step(plan); torch.cuda.synchronize()
if os.environ.get("PROFILE_LAYERS_JSON"):
    # the implicit-GEMM calls of one step in issue order, with their algorithmic bytes (each operand read once, the
    # result written once): tools/pmc_layers.py joins this with per-dispatch rocprofv3 counters
    import json
    def alg_bytes(name, k):
        b, ih, iw, ci, oh, co, kk, st, dil = k
        ow = oh * iw // ih if ih == iw else oh
        xin, yout, wt = b * ih * iw * ci, b * oh * ow * co, kk * kk * ci * co
        extra = 2 * xin if name == "conv2d_fwd_addrelu" else 0     # residual operand read, sum written
        if name == "conv2d_dgrad_bnbwd":
            extra = xin                                            # z of the BatchNormalization whose statistics it takes
        return 4 * (xin + yout + wt + extra)
    json.dump([dict(op=n.split()[0], key=list(k), ms=e0.elapsed_time(e1), flop=fl, bytes=nb) for n, k, e0, e1, fl, nb in recs],
              open(os.environ["PROFILE_LAYERS_JSON"], "w"))
agg = collections.OrderedDict()
for name, key, e0, e1, fl, nb in recs:
    k = (name, key)
    t = e0.elapsed_time(e1)
    a = agg.setdefault(k, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += fl; a[3] += nb
tot = sum(a[1] for a in agg.values())
print("total conv ms %.2f, TF %.1f" % (tot, sum(a[2] for a in agg.values()) / tot / 1e9))
print("total algorithmic bytes %.2f GB -> %.0f GB/s" % (sum(a[3] for a in agg.values()) / 1e9, sum(a[3] for a in agg.values()) / tot / 1e6))
print("%-25s %-44s %4s %8s %7s %7s %6s" % ("op (bits in>out)", "B,H,W,Cin,OH,Cout,k,s,d", "n", "ms", "TF", "GB/s", "%"))
for (name, key), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-25s %-44s %4d %8.3f %7.1f %7.0f %6.1f" % (name, str(key), a[0], a[1], a[2] / a[1] / 1e9, a[3] / a[1] / 1e6, 100 * a[1] / tot))
for d in ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad"):      # (conv2d_fwd_addrelu, conv2d_dgrad_bnbwd count with their direction)
    sel = [a for (n, k), a in agg.items() if n.startswith(d)]
    t = sum(a[1] for a in sel); f = sum(a[2] for a in sel)
    print("%s: %.2f ms, %.1f TF, %d launches" % (d, t, f / t / 1e9, sum(a[0] for a in sel)))
ideal = sum(a[2] for a in agg.values()) / 130e9
print("time above 130 TF/s pace: %.2f ms of %.2f" % (tot - ideal, tot))
lost = sorted(((a[1] - a[2] / 130e9, n, k, a[0]) for (n, k), a in agg.items()), reverse=True)[:25]
for l, n, k, c in lost: print("lost %.3f ms  %s %s x%d" % (l, n, k, c))

"""Loss trajectory of the SSD300 trainer over several steps: the CPU oracle (fp64; the reference's Keras graph restated,
oracle/ssd_resnet_dct.py) against the GPU path in each arithmetic mode, same initial weights, same batches, the reference
trainer's optimizer (SGD 0.001, momentum 0.9).  Measurement tool (imports the oracle: not part of the product path).
    python tools/loss_trajectory.py [archi=deconv] [batch=8] [steps=8] [modes=float32,float32_mfma,float32x3,float16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jpeg_detection_resnet_ssd_amd import workloads
from jpeg_detection_resnet_ssd_amd.keras import backend as K
from oracle import ssd_resnet_dct as oracle

archi = sys.argv[1] if len(sys.argv) > 1 else "deconv"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
modes = (sys.argv[4] if len(sys.argv) > 4 else "float32,float32_mfma,float32x3,float16").split(",")

model, sizes = workloads.build_ssd(archi)
batches = [workloads.synthetic_batch(archi, sizes, batch, seed=500 + s) for s in range(steps)]
w0 = model.get_weights_dict()
torch.set_num_threads(min(16, os.cpu_count() or 1))
wt = {k: torch.from_numpy(v).double() for k, v in w0.items()}
vel, ref = None, []
t0 = time.time()
for s, (x, y) in enumerate(batches):
    r = oracle.ssd_training_step(wt, [torch.from_numpy(a).double() for a in x], torch.from_numpy(y).double(), archi,
                                 lr=0.001, momentum=0.9, velocities=vel, iterations=s)
    wt, vel = r["new_weights"], r["new_velocities"]
    ref.append(r["loss"])
    print("oracle step %d loss %.6f (%.0f s)" % (s + 1, r["loss"], time.time() - t0), flush=True)
rows = {}
for mode in modes:
    K.set_floatx(mode)
    m, _ = workloads.build_ssd(archi)
    assert m.set_weights_dict(w0, strict=True) == len(w0)
    rows[mode] = [float(m.train_on_batch(x, y)) for x, y in batches]
    torch.cuda.synchronize()
    K.set_floatx("float32")
    del m
print("\n%s SSD300, batch %d, %d steps; relative deviation of the step's loss from the fp64 oracle's" % (archi, batch, steps))
print("%-5s %-12s " % ("step", "oracle fp64") + " ".join("%-13s" % m for m in modes))
for s in range(steps):
    print("%-5d %-12.5f " % (s + 1, ref[s]) + " ".join("%-13.2e" % (abs(rows[m][s] - ref[s]) / abs(ref[s])) for m in modes))

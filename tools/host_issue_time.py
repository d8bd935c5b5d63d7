"""Host time to ISSUE one training step (no device sync inside) vs. device time per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import workloads
model, sizes = workloads.build_ssd("deconv")
x, y = workloads.synthetic_batch("deconv", sizes, 32, fast=True)
plan = model._plan(32, True, True)
model._upload(plan, x, y)
for _ in range(3): model.run_train_step(plan)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    n = 10
    issue = 0.0
    for _ in range(n):
        a = time.perf_counter(); model.run_train_step(plan); issue += time.perf_counter() - a
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("issue %.2f ms/step, wall %.2f ms/step, launches fwd %d bwd %d" % (issue / n * 1e3, (t1 - t0) / n * 1e3, len(plan.fwd), len(plan.bwd)))
# pure issue cost: time while GPU queue is deep is bounded by back-pressure; measure single step after sync
torch.cuda.synchronize()
a = time.perf_counter(); model.run_train_step(plan); b = time.perf_counter(); torch.cuda.synchronize()
print("single step issue after sync: %.2f ms" % ((b - a) * 1e3))

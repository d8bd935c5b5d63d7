"""Experiment: main launch chain on a high-priority HIP stream, weight-gradient side stream at normal priority."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import workloads
model, sizes = workloads.build_ssd("deconv")
x, y = workloads.synthetic_batch("deconv", sizes, 32, fast=True)
plan = model._plan(32, True, True)
model._upload(plan, x, y)
def run(stream, n=30):
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(3): model.run_train_step(plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): model.run_train_step(plan)
        torch.cuda.synchronize()
    return 32 * n / (time.perf_counter() - t0)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
for name, s in (("default stream", torch.cuda.current_stream()), ("high-priority stream", torch.cuda.Stream(priority=-1)),
                ("normal new stream", torch.cuda.Stream()), ("default stream", torch.cuda.current_stream()),
                ("high-priority stream", torch.cuda.Stream(priority=-1))):
    print("%-22s %.1f img/s" % (name, run(s)), flush=True)

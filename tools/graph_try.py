"""Experiment: one training step captured into a HIP graph (torch.cuda.CUDAGraph) and replayed, against the eager launch
sequence.  The plan is launch-only, so the capture is mechanical; NOT used by the product path: the optimizer's learning
rate, iteration count and 1/world are kernel ARGUMENTS, a replayed graph would freeze them (and a data-parallel step holds
collectives).   python tools/graph_try.py deconv 32 [float16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import workloads
archi = sys.argv[1] if len(sys.argv) > 1 else "deconv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if len(sys.argv) > 3:
    from jpeg_detection_resnet_ssd_amd.keras import backend as KB
    KB.set_floatx(sys.argv[3])
model, sizes = workloads.build_ssd(archi)
x, y = workloads.synthetic_batch(archi, sizes, B, fast=True)
plan = model._plan(B, True, True)
model._upload(plan, x, y)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


eager = timed(lambda: model.run_train_step(plan))
print("eager   %.3f ms/step" % eager, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        model.run_train_step(plan)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        model.run_train_step(plan)
    torch.cuda.synchronize()
    graph = timed(g.replay)
    print("graph   %.3f ms/step (%.1f %%)" % (graph, 100.0 * (graph - eager) / eager))
    loss_g = model._loss_value(plan)
    model.run_train_step(plan); torch.cuda.synchronize()
    print("loss after graph replays %.5f, after one more eager step %.5f" % (loss_g, model._loss_value(plan)))
except Exception as e:     # noqa: BLE001 -- an experiment: report what the capture tripped over
    print("capture failed: %s: %s" % (type(e).__name__, str(e)[:400]))

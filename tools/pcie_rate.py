"""Training rate when every step takes its batch from HOST memory (what fit_generator does), vs resident inputs:
(a) inputs + host-encoded y_true uploaded each step, (b) inputs + raw boxes, SSDInputEncoder on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jpeg_detection_resnet_ssd_amd import workloads
from jpeg_detection_resnet_ssd_amd.data import synthetic_dct as sd
from jpeg_detection_resnet_ssd_amd.ssd_encoder_decoder.ssd_input_encoder import DeviceLabelEncoder
archi, B, N = "deconv", 32, 30
model, sizes = workloads.build_ssd(archi)
enc = workloads.make_encoder(sizes)
batches = []
for s in range(4):
    x = sd.fast_dct_batch(B, seed=s, split_chroma=True)
    gt = sd.random_ground_truth(B, seed=s)
    batches.append((x, enc(gt).astype(np.float32), DeviceLabelEncoder(enc)(gt)))
plan = model._plan(B, True, True)
def run(mode):
    for i in range(3):
        x, y, pend = batches[i % 4]
        model._upload(plan, x, y if mode != "device-encoder" else pend); model.run_train_step(plan)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        x, y, pend = batches[i % 4]
        if mode != "resident":
            model._upload(plan, x, y if mode == "host-encoded" else pend)
        model.run_train_step(plan)
    torch.cuda.synchronize()
    return B * N / (time.perf_counter() - t0)
for mode in ("resident", "host-encoded", "device-encoder", "resident"):
    print("%-16s %.1f img/s" % (mode, run(mode)), flush=True)

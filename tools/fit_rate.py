"""images/s of Model.fit_generator on the synthetic JPEG-DCT generator (host batch production + upload + step)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_detection_resnet_ssd_amd import workloads
from jpeg_detection_resnet_ssd_amd.data.generators import SyntheticDataGeneratorDCT
from jpeg_detection_resnet_ssd_amd.ssd_encoder_decoder.ssd_input_encoder import DeviceLabelEncoder
archi, B, steps = "deconv", 32, 120
model, sizes = workloads.build_ssd(archi)
enc = workloads.make_encoder(sizes)
for name, encoder in (("host encoder", enc), ("device encoder", DeviceLabelEncoder(enc))):
    gen = SyntheticDataGeneratorDCT(n_images=512, seed=1).generate(batch_size=B, label_encoder=encoder, deconv=True)
    model.fit_generator(gen, steps_per_epoch=5, epochs=1, verbose=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.fit_generator(gen, steps_per_epoch=steps, epochs=1, verbose=0)
    torch.cuda.synchronize()
    print("%-15s %s%.1f img/s" % (name, "", B * steps / (time.perf_counter() - t0)), flush=True)

# framework overhead only: a generator that cycles over batches prepared in advance
import itertools
from jpeg_detection_resnet_ssd_amd.data import synthetic_dct as sd
pre = []
for i in range(4):
    x = sd.fast_dct_batch(B, seed=i, split_chroma=True)
    gt = sd.random_ground_truth(B, seed=i)
    pre.append((x, DeviceLabelEncoder(enc)(gt)))
gen = itertools.cycle(pre)
model.fit_generator(gen, steps_per_epoch=5, epochs=1, verbose=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
model.fit_generator(gen, steps_per_epoch=steps, epochs=1, verbose=0)
torch.cuda.synchronize()
print("%-15s %s%.1f img/s" % ("prepared batches", "", B * steps / (time.perf_counter() - t0)), flush=True)

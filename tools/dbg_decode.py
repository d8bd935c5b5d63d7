import numpy as np, torch
from jpeg_detection_resnet_ssd_amd import workloads
train, sizes = workloads.build_ssd("ssd_custom", compile_model=False)
w = train.get_weights_dict()
for k in w:
    if "mbox_loc" in k: w[k] = w[k]*1e-5
    if "mbox_conf" in k: w[k] = w[k]*1e-5
train.set_weights_dict(w)
x = workloads.synthetic_batch("ssd_custom", sizes, 2, seed=3)[0]
raw = train.predict(x, batch_size=2)
c = raw[..., 1:21]
print("conf max", c.max(), "n>=0.999", (c > 0.999).sum(), "uniq top", np.sort(c.ravel())[-10:])
print("loc absmax", np.abs(raw[..., 21:25]).max())

import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_ssd_gpu as T
from oracle import ssd_resnet_dct as oracle
archi = "deconv"
model, sizes = T.build(archi)
x, y_true = T.make_batch(archi, sizes, 2)
w0 = T.perturb_weights(model)
loss = model.train_on_batch(x, y_true)
torch.cuda.synchronize()
grads = {w.key: w.grad.detach().cpu().clone() for w in model.weight_specs if w.trainable}
wt = {k: torch.from_numpy(v).double() for k, v in w0.items()}
ref = oracle.ssd_training_step(wt, [torch.from_numpy(a).double() for a in x], torch.from_numpy(y_true).double(), archi)
gmax = max(float(v.abs().max()) for v in ref["grads"].values())
print("gmax", gmax, [k for k, v in ref["grads"].items() if float(v.abs().max()) == gmax])
for k in ["conv2d_transpose_1/kernel", "conv2d_transpose_1/bias", "conv2d_transpose_2/kernel", "conv2d_transpose_2/bias",
          "batch_normalization_1/gamma", "batch_normalization_1/beta", "res4a2_branch2a/kernel", "res4a2_branch1/kernel"]:
    a, r = grads[k].double(), ref["grads"][k]
    print("%-30s gpu max %.4e  ref max %.4e  relL2 %.3e" % (k, float(a.abs().max()), float(r.abs().max()), float((a - r).norm() / (r.norm() + 1e-30))))

import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_blocks_gpu as TB
from jpeg_detection_resnet_ssd_amd import engine
from jpeg_detection_resnet_ssd_amd.keras import backend as K
from jpeg_detection_resnet_ssd_amd.keras.layers import BatchNormalization, Input
from jpeg_detection_resnet_ssd_amd.keras.models import Model
from jpeg_detection_resnet_ssd_amd.models.resnet_dct_blocks import conv_block, identity_block
ksz = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K.clear_session(); K.set_random_seed(5)
b, hw, cin = 4, 19, 128
inp = Input((hw, hw, cin))
x = BatchNormalization()(inp)
x = conv_block(x, 3, [64, 64, 128], stage=9, block="p", strides=(1, 1))
y = identity_block(x, ksz, [64, 64, 128], stage=1, block="a")
model = Model(inp, y)
w0 = TB._perturb(model)
allocs = []
orig_empty = engine.Plan.empty
def rec_empty(self, *shape):
    t = orig_empty(self, *shape); t.fill_(float("nan")); allocs.append((len(self.fwd), len(self.bwd), len(self._bwd_builders), t)); return t
engine.Plan.empty = rec_empty
g = torch.Generator().manual_seed(1)
xin = (torch.randn(b, hw, hw, cin, generator=g) * 20).numpy()
dy = torch.randn(b, hw, hw, 128, generator=g).numpy()
plan = model._plan(b, True, False, external_grad=True)
nfwd_allocs = sum(1 for a in allocs if a[1] == 0 and a[2] > 0 or True)
model._upload(plan, [xin], None)
plan.external_grad.copy_(torch.from_numpy(dy))
plan.run_forward(); torch.cuda.synchronize()
snap = [a[3].clone() for a in allocs]
# step through backward, report first op after which a forward-time buffer changes, or NaN shows up in a written grad
fwd_time = [i for i, a in enumerate(allocs) if a[1] == 0 and len(plan.bwd) > 0]
for j, f in enumerate(plan.bwd):
    f(); torch.cuda.synchronize()
changed = []
for i, a in enumerate(allocs):
    same = torch.equal(torch.nan_to_num(a[3], nan=12345.0), torch.nan_to_num(snap[i], nan=12345.0))
    if not same: changed.append((i, a[0], a[1], tuple(a[3].shape), bool(torch.isnan(a[3]).any())))
print("ksz", ksz, "n allocs", len(allocs))
for c in changed: print("changed", c)

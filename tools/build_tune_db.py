"""Measure the conv tile choice of every geometry of the benchmark workloads on this MI355X and write the table
that ships in-tree (jpeg_detection_resnet_ssd_amd/tuned/gfx950_conv.json).
    DJ_TUNE_DB=0 DJ_AUTOTUNE_REPS=5 python tools/build_tune_db.py gpurun_out/gfx950_conv.json"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
floatx = sys.argv[2] if len(sys.argv) > 2 else "float32"     # float16: the table of the reduced-precision mode
os.environ["DJ_TUNE_SAVE"] = out
import torch
from jpeg_detection_resnet_ssd_amd import workloads
from jpeg_detection_resnet_ssd_amd.keras import backend as K
K.set_floatx(floatx)
t0 = time.time()
for archi in ("deconv", "ssd_custom", "up_sampling"):
    model, sizes = workloads.build_ssd(archi)
    model._plan(32, True, True)
    print("%s SSD B=32 training plan tuned, %.0f s" % (archi, time.time() - t0), flush=True)
    model._plan(32, False, False)
    del model
    torch.cuda.empty_cache()
from jpeg_detection_resnet_ssd_amd.vgg_jpeg_keras.networks.resnet_dct import ResNet50Custom, ResNet50RGB
from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
for name, build, b in (("deconv classifier", lambda: ResNet50Custom(weights=None, archi="deconv"), 64),
                       ("late_concat_rfa_thinner classifier", lambda: ResNet50Custom(weights=None, archi="late_concat_rfa_thinner"), 64),
                       ("resnet_rgb classifier", lambda: ResNet50RGB(weights=None), 64),
                       ("resnet_rgb classifier", lambda: ResNet50RGB(weights=None), 4)):
    K.clear_session()
    m = build()
    m.compile(optimizer=SGD(lr=0.1, momentum=0.9, decay=1e-4, nesterov=True), loss="categorical_crossentropy")
    m._plan(b, True, True)
    print("%s B=%d tuned, %.0f s" % (name, b, time.time() - t0), flush=True)
    del m
    torch.cuda.empty_cache()
from jpeg_detection_resnet_ssd_amd.engine import save_tune_db
print("entries:", save_tune_db(out))

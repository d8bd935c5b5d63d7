import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_encode_gpu import encoder, SIZES
from jpeg_detection_resnet_ssd_amd.bounding_box_utils.bounding_box_utils import iou
rng = np.random.default_rng(0)
same = np.array([[3, 50, 60, 150, 200], [7, 50, 60, 150, 200], [9, 50, 60, 150, 200]], dtype=float)
grid = np.array([[1 + (i % 20), 8 * i + 0.0, 8 * i + 0.0, 8 * i + 30.0, 8 * i + 30.0] for i in range(30)])
many = []
for _ in range(60):
    x0, y0 = rng.uniform(0, 250, 2)
    w, h = rng.uniform(20, 50, 2)
    many.append([rng.integers(1, 21), x0, y0, min(x0 + w, 300), min(y0 + h, 300)])
gt = [same, grid, np.array(many), np.zeros((0, 5)), np.array([[5, 0, 0, 300, 300]], dtype=float)]
enc = encoder(SIZES["custom"])
host = enc(gt).astype(np.float32); dev = enc.encode_on_device(gt).cpu().numpy()
bad = np.argwhere(host[..., :21] != dev[..., :21])
print(bad)
for img, a in sorted(set((int(b[0]), int(b[1])) for b in bad)):
    print("img", img, "anchor", a, "host cls", host[img, a, :21].argmax(), host[img,a,:21].sum(), "dev cls", dev[img, a, :21].argmax(), dev[img,a,:21].sum())
    tmpl = enc.generate_encoding_template(1)[0]
    lab = gt[img].astype(float).copy(); lab[:, 1:] /= 300.0
    from jpeg_detection_resnet_ssd_amd.bounding_box_utils.bounding_box_utils import convert_coordinates
    labc = convert_coordinates(lab, 1, "corners2centroids")
    s = iou(labc[:, 1:], tmpl[a:a+1, -12:-8], coords="centroids")
    order = np.argsort(-s[:, 0])[:4]
    print("   top gts for this anchor:", [(int(g), float(s[g, 0])) for g in order])

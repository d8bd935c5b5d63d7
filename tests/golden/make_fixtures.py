"""Generates tests/golden/*.npz by IMPORTING the reference's own numpy-only modules
(localisation_part/bounding_box_utils/bounding_box_utils.py, ssd_encoder_decoder/matching_utils.py,
ssd_encoder_decoder/ssd_input_encoder.py, ssd_encoder_decoder/ssd_output_decoder.py) from /root/reference.
Run in the build container only (the reference does not travel to the GPU box); the outputs are data:
inputs and expected outputs of those functions.  The reference calls NumPy aliases removed in NumPy >= 1.24
(np.float, np.int), so they are restored before the import."""
import os
import sys

import numpy as np

np.float = float  # noqa
np.int = int      # noqa
REF = "/root/reference/localisation_part"
sys.path.insert(0, REF)
from bounding_box_utils.bounding_box_utils import convert_coordinates, intersection_area, iou  # noqa: E402
from ssd_encoder_decoder.matching_utils import match_bipartite_greedy, match_multi  # noqa: E402
from ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TRAIN = dict(img_height=300, img_width=300, n_classes=20, scales=[0.1, 0.2, 0.37, 0.54, 0.71, 0.88, 1.05],
             aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                      [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]],
             two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=[0.5] * 6, clip_boxes=False,
             variances=[0.1, 0.1, 0.2, 0.2], matching_type="multi", pos_iou_threshold=0.5, neg_iou_limit=0.5,
             normalize_coords=True)
SIZES = {"custom": [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)],
         "identical": [(38, 38), (10, 10), (5, 5), (5, 5), (3, 3), (1, 1)]}

GT = [np.array([[5, 30., 40., 200., 250.], [12, 100., 100., 140., 160.], [1, 10., 200., 60., 290.]]),
      np.zeros((0, 5)),
      np.array([[20, 0., 0., 299., 299.]]),
      np.array([[3, 120., 130., 180., 170.], [3, 125., 128., 185., 175.], [7, 250., 20., 298., 90.],
                [15, 5., 5., 35., 45.], [9, 60., 220., 260., 280.]])]


def main():
    rng = np.random.default_rng(0)
    # --- box utilities -----------------------------------------------------------------------------
    a = np.concatenate([rng.uniform(0, 200, (7, 2)), rng.uniform(201, 300, (7, 2))], axis=1)   # corners
    b = np.concatenate([rng.uniform(0, 150, (11, 2)), rng.uniform(151, 300, (11, 2))], axis=1)
    conv = {}
    for name in ["minmax2centroids", "centroids2minmax", "corners2centroids", "centroids2corners", "minmax2corners",
                 "corners2minmax"]:
        for bp in ["half", "include", "exclude"]:
            conv["%s__%s" % (name, bp)] = convert_coordinates(a, 0, name, border_pixels=bp)
    np.savez(os.path.join(OUT, "box_utils.npz"), a=a, b=b,
             iou_corners_outer=iou(a, b, coords="corners", mode="outer_product"),
             iou_corners_elem=iou(a, a[::-1].copy(), coords="corners", mode="element-wise"),
             iou_centroids_outer=iou(convert_coordinates(a, 0, "corners2centroids"),
                                     convert_coordinates(b, 0, "corners2centroids"), coords="centroids"),
             iou_minmax_include=iou(convert_coordinates(a, 0, "corners2minmax"), convert_coordinates(b, 0, "corners2minmax"),
                                    coords="minmax", border_pixels="include"),
             inter_corners=intersection_area(a, b, coords="corners"), **conv)
    # --- matching -----------------------------------------------------------------------------------
    w = rng.uniform(0, 1, (6, 40))
    w[2, 5] = w[4, 5] = 0.99  # contested anchor
    gi, ai = match_multi(w, 0.8)
    np.savez(os.path.join(OUT, "matching.npz"), weights=w, bipartite=match_bipartite_greedy(w), multi_gt=gi, multi_anchor=ai)
    # --- anchors + encoded targets -------------------------------------------------------------------
    for key, sizes in SIZES.items():
        enc = SSDInputEncoder(predictor_sizes=sizes, **TRAIN)
        tmpl = enc.generate_encoding_template(batch_size=1)
        y = enc(GT)
        per_layer = {"layer%d" % i: b for i, b in enumerate(enc.boxes_list)}
        np.savez_compressed(os.path.join(OUT, "encoder_%s.npz" % key), template=tmpl[0], y_true=y.astype(np.float64),
                            gt_count=np.array([len(g) for g in GT]), gt=np.concatenate([g for g in GT if len(g)]),
                            **per_layer)
        print(key, tmpl.shape, y.shape, "positives per image", (y[:, :, 1:21].max(-1) > 0).sum(1),
              "neutral", ((y[:, :, :21].sum(-1)) == 0).sum(1))
    # other coords / matching variants on a small grid
    small = SSDInputEncoder(img_height=300, img_width=300, n_classes=3, predictor_sizes=[(4, 4), (2, 2)], min_scale=0.2,
                            max_scale=0.8, aspect_ratios_global=[0.5, 1.0, 2.0], two_boxes_for_ar1=True, clip_boxes=True,
                            matching_type="bipartite", coords="corners", normalize_coords=False, neg_iou_limit=0.3)
    gt_small = [np.array([[1, 20., 30., 150., 200.], [3, 160., 150., 290., 280.]])]
    np.savez(os.path.join(OUT, "encoder_small_corners.npz"), y_true=small(gt_small), gt=gt_small[0])
    small2 = SSDInputEncoder(img_height=200, img_width=300, n_classes=3, predictor_sizes=[(4, 6), (2, 3)], min_scale=0.2,
                             max_scale=0.8, aspect_ratios_global=[0.5, 1.0, 2.0], two_boxes_for_ar1=False,
                             coords="minmax", normalize_coords=True)
    np.savez(os.path.join(OUT, "encoder_small_minmax.npz"), y_true=small2(gt_small), gt=gt_small[0])


def decode_fixture():
    """Reference numpy decode_detections on synthetic predictions over the real anchor set (2 images, 8732 boxes)."""
    from ssd_encoder_decoder.ssd_output_decoder import decode_detections
    enc = SSDInputEncoder(predictor_sizes=SIZES["custom"], **TRAIN)
    tmpl = enc.generate_encoding_template(batch_size=2)
    rng = np.random.default_rng(5)
    n = tmpl.shape[1]
    logits = rng.normal(0, 1.0, (2, n, 21))
    logits[..., 0] += 4.0                                   # mostly background
    hot = rng.choice(n, size=(2, 300), replace=True)
    for b in range(2):
        logits[b, hot[b], rng.integers(1, 21, 300)] += rng.uniform(3, 9, 300)
    p = np.exp(logits - logits.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    y_pred = np.concatenate([p, rng.normal(0, 0.6, (2, n, 4)), tmpl[..., -8:]], axis=-1).astype(np.float32)
    dec = decode_detections(y_pred.astype(np.float64), confidence_thresh=0.3, iou_threshold=0.45, top_k=200,
                            normalize_coords=True, img_height=300, img_width=300)
    dec2 = decode_detections(y_pred.astype(np.float64), confidence_thresh=0.05, iou_threshold=0.45, top_k=50,
                             normalize_coords=True, img_height=300, img_width=300)
    from ssd_encoder_decoder.ssd_output_decoder import decode_detections_fast
    fast = decode_detections_fast(y_pred.astype(np.float64), confidence_thresh=0.3, iou_threshold=0.45, top_k=200,
                                  normalize_coords=True, img_height=300, img_width=300)
    np.savez_compressed(os.path.join(OUT, "decode.npz"), y_pred=y_pred, d0=dec[0], d1=dec[1], e0=dec2[0], e1=dec2[1],
                        f0=fast[0], f1=fast[1])
    print("decode fixture: kept", [d.shape for d in dec], [d.shape for d in dec2], [d.shape for d in fast])


if __name__ == "__main__":
    main()
    decode_fixture()

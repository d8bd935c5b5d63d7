"""Pascal-VOC evaluator (eval_utils/average_precision_evaluator.py) on hand-derived known answers.  The reference's
module is not importable here (cv2 / bs4 / h5py), so these pins come from the VOC definitions, not from its output."""
import numpy as np
import pytest

from jpeg_detection_resnet_ssd_amd.eval_utils.average_precision_evaluator import Evaluator


class FakeData(object):
    def __init__(self, labels, eval_neutral=None):
        self.labels = labels
        self.image_ids = ["img%d" % i for i in range(len(labels))]
        self.eval_neutral = eval_neutral

    def get_dataset_size(self):
        return len(self.labels)


def evaluator(labels, preds_per_class, n_classes=2, eval_neutral=None):
    ev = Evaluator(model=None, n_classes=n_classes, data_generator=FakeData(labels, eval_neutral), model_mode="inference")
    ev.prediction_results = [[]] + [list(p) for p in preds_per_class]
    return ev


def run(ev, **kw):
    ev.get_num_gt_per_class(ignore_neutral_boxes=True, verbose=False)
    ev.match_predictions(ignore_neutral_boxes=True, matching_iou_threshold=0.5, border_pixels="include",
                         sorting_algorithm="mergesort", verbose=False)
    ev.compute_precision_recall(verbose=False)
    ev.compute_average_precisions(verbose=False, **kw)
    return ev.compute_mean_average_precision()


def test_textbook_precision_recall_curve():
    """Class 1: 3 objects in 2 images; 5 detections ranked TP, FP, TP, FP(duplicate), TP.
    precision = 1, 1/2, 2/3, 2/4, 3/5; recall = 1/3, 1/3, 2/3, 2/3, 1.
    11-point AP = (4*1 + 3*2/3 + 4*3/5)/11."""
    labels = [np.array([[1, 10, 10, 50, 50], [1, 100, 100, 160, 160]]), np.array([[1, 20, 20, 80, 80], [2, 0, 0, 30, 30]])]
    c1 = [("img0", 0.95, 10, 10, 50, 50),        # TP
          ("img1", 0.90, 200, 200, 250, 250),    # FP: no overlap
          ("img0", 0.80, 101, 101, 160, 160),    # TP
          ("img0", 0.70, 12, 12, 50, 50),        # FP: duplicate of the first object
          ("img1", 0.60, 20, 22, 80, 80)]        # TP
    ev = evaluator(labels, [c1, []])
    m = run(ev)
    np.testing.assert_array_equal(ev.true_positives[1], [1, 0, 1, 0, 1])
    np.testing.assert_array_equal(ev.false_positives[1], [0, 1, 0, 1, 0])
    np.testing.assert_allclose(ev.cumulative_precisions[1], [1, 1 / 2, 2 / 3, 2 / 4, 3 / 5])
    np.testing.assert_allclose(ev.cumulative_recalls[1], [1 / 3, 1 / 3, 2 / 3, 2 / 3, 1])
    ap_sample = (4 * 1.0 + 3 * (2 / 3) + 4 * (3 / 5)) / 11
    np.testing.assert_allclose(ev.average_precisions[1], ap_sample)
    assert ev.average_precisions[2] == 0.0 and list(ev.num_gt_per_class) == [0, 3, 1]
    np.testing.assert_allclose(m, ap_sample / 2)
    run(ev, mode="integrate")
    # reference formula (:893-918): unique recalls r = (1/3, 2/3, 1); rectangle i spans r[i] -> r[i+1] with height
    # max{precision at recall in [r[i], r[i+1])} (monotone envelope): 1 * 1/3 + 2/3 * 1/3; the precision at the last
    # recall value and the step 0 -> r[0] do not enter
    np.testing.assert_allclose(ev.average_precisions[1], 1.0 * (1 / 3) + (2 / 3) * (1 / 3))


def test_neutral_boxes_and_threshold_edge():
    """A detection on a 'difficult' box is neither TP nor FP and the box is not counted; an overlap equal to the
    threshold is a match (`<` in :728).  As in the reference's iou(), the intersection is w*h and only the areas get
    the 'include' +1: gt 10x10 px box (0,0,9,9) vs (0,0,9,4): 36 / (100 + 50 - 36)."""
    from jpeg_detection_resnet_ssd_amd.bounding_box_utils.bounding_box_utils import iou
    labels = [np.array([[1, 0, 0, 9, 9], [1, 50, 50, 99, 99]])]
    neutral = [np.array([False, True])]
    preds = [("img0", 0.9, 50, 50, 99, 99),      # on the difficult box: ignored
             ("img0", 0.8, 0, 0, 9, 4),          # IoU == threshold -> TP
             ("img0", 0.7, 0, 0, 9, 3)]          # just below -> FP
    thr = float(iou(np.array([0., 0, 9, 9]), np.array([0., 0, 9, 4]), coords="corners", mode="element-wise",
                    border_pixels="include")[0])
    np.testing.assert_allclose(thr, 36.0 / 114.0)
    ev = evaluator(labels, [preds], n_classes=1, eval_neutral=neutral)
    ev.get_num_gt_per_class(ignore_neutral_boxes=True, verbose=False)
    ev.match_predictions(ignore_neutral_boxes=True, matching_iou_threshold=thr, border_pixels="include",
                         sorting_algorithm="mergesort", verbose=False)
    ev.compute_precision_recall(verbose=False)
    ev.compute_average_precisions(verbose=False)
    assert list(ev.num_gt_per_class) == [0, 1]
    np.testing.assert_array_equal(ev.true_positives[1], [0, 1, 0])
    np.testing.assert_array_equal(ev.false_positives[1], [0, 0, 1])
    np.testing.assert_allclose(ev.average_precisions[1], 1.0)


def test_full_call_on_a_model_stub():
    """__call__ end to end with a model whose `predict` returns the ground truth as (padded) decoded detections."""
    from jpeg_detection_resnet_ssd_amd.data.generators import SyntheticDataGeneratorDCT
    data = SyntheticDataGeneratorDCT(n_images=10, seed=3)

    class Oracle(object):
        def __init__(self):
            self.seen = 0

        def predict(self, batch_X):
            b = batch_X[0].shape[0]
            out = np.zeros((b, 8, 6))
            for k in range(b):
                gt = data.labels[(self.seen + k) % 10]
                out[k, :len(gt), 0] = gt[:, 0]
                out[k, :len(gt), 1] = 0.9
                out[k, :len(gt), 2:] = gt[:, 1:]
            self.seen += b
            return out

    ev = Evaluator(model=Oracle(), n_classes=20, data_generator=data, model_mode="inference")
    m, aps = ev(img_height=300, img_width=300, batch_size=4, verbose=False, return_average_precisions=True)
    present = sorted({int(c) for g in data.labels for c in g[:, 0]})
    for c in range(1, 21):
        assert aps[c] == (1.0 if c in present else 0.0)
    np.testing.assert_allclose(m, len(present) / 20.0)
    assert sum(len(r) for r in ev.prediction_results) == sum(len(g) for g in data.labels)

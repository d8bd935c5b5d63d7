"""Pascal-VOC mean average precision of an SSD model on a dataset (host numpy around `model.predict`).
Same class, method names, arguments and result containers as
localisation_part/eval_utils/average_precision_evaluator.py:32-947 (`Evaluator.__call__` :97, `predict_on_dataset`
:262, `get_num_gt_per_class` :489, `match_predictions` :570, `compute_precision_recall` :781,
`compute_average_precisions` :826, `compute_mean_average_precision` :927), so `evaluation.py:102-131` runs against it.

Parity unpinned: the reference module cannot be imported here (it pulls cv2 / bs4 / h5py through the data generator),
so the known answers in tests/test_evaluator_cpu.py are hand-derived from the VOC definitions.  One deliberate
difference: the reference iterates `range(len(predictions.shape))` (= 1 prediction) when verbose=False
(:699); here every prediction is matched regardless of `verbose`."""
from math import ceil

import numpy as np

from ..bounding_box_utils.bounding_box_utils import iou
from ..ssd_encoder_decoder.ssd_output_decoder import decode_detections


class Evaluator(object):
    def __init__(self, model, n_classes, data_generator, model_mode="inference",
                 pred_format={"class_id": 0, "conf": 1, "xmin": 2, "ymin": 3, "xmax": 4, "ymax": 5},
                 gt_format={"class_id": 0, "xmin": 1, "ymin": 2, "xmax": 3, "ymax": 4}, ignore_under_area=0):
        self.model = model
        self.data_generator = data_generator
        self.n_classes = n_classes
        self.model_mode = model_mode
        self.pred_format = pred_format
        self.gt_format = gt_format
        self.ignore_under_area = ignore_under_area
        self.prediction_results = None
        self.num_gt_per_class = None
        self.true_positives = None
        self.false_positives = None
        self.cumulative_true_positives = None
        self.cumulative_false_positives = None
        self.cumulative_precisions = None
        self.cumulative_recalls = None
        self.average_precisions = None
        self.mean_average_precision = None

    def __call__(self, img_height, img_width, batch_size, data_generator_mode="resize", round_confidences=False,
                 matching_iou_threshold=0.5, border_pixels="include", sorting_algorithm="quicksort",
                 average_precision_mode="sample", num_recall_points=11, ignore_neutral_boxes=True,
                 return_precisions=False, return_recalls=False, return_average_precisions=False, verbose=True,
                 decoding_confidence_thresh=0.01, decoding_iou_threshold=0.45, decoding_top_k=200,
                 decoding_pred_coords="centroids", decoding_normalize_coords=True):
        self.predict_on_dataset(img_height=img_height, img_width=img_width, batch_size=batch_size,
                                data_generator_mode=data_generator_mode,
                                decoding_confidence_thresh=decoding_confidence_thresh,
                                decoding_iou_threshold=decoding_iou_threshold, decoding_top_k=decoding_top_k,
                                decoding_pred_coords=decoding_pred_coords,
                                decoding_normalize_coords=decoding_normalize_coords,
                                decoding_border_pixels=border_pixels, round_confidences=round_confidences,
                                verbose=verbose, ret=False)
        self.get_num_gt_per_class(ignore_neutral_boxes=ignore_neutral_boxes, verbose=False, ret=False)
        self.match_predictions(ignore_neutral_boxes=ignore_neutral_boxes, matching_iou_threshold=matching_iou_threshold,
                               border_pixels=border_pixels, sorting_algorithm=sorting_algorithm, verbose=verbose, ret=False)
        self.compute_precision_recall(verbose=verbose, ret=False)
        self.compute_average_precisions(mode=average_precision_mode, num_recall_points=num_recall_points,
                                        verbose=verbose, ret=False)
        mean_average_precision = self.compute_mean_average_precision(ret=True)
        if return_precisions or return_recalls or return_average_precisions:
            ret = [mean_average_precision]
            if return_average_precisions:
                ret.append(self.average_precisions)
            if return_precisions:
                ret.append(self.cumulative_precisions)
            if return_recalls:
                ret.append(self.cumulative_recalls)
            return ret
        return mean_average_precision

    # ---- predictions --------------------------------------------------------------------------------------------
    def predict_on_dataset(self, img_height, img_width, batch_size, data_generator_mode="resize",
                           decoding_confidence_thresh=0.01, decoding_iou_threshold=0.45, decoding_top_k=200,
                           decoding_pred_coords="centroids", decoding_normalize_coords=True,
                           decoding_border_pixels="include", round_confidences=False, verbose=True, ret=False):
        """-> results[class_id] = list of (image_id, confidence, xmin, ymin, xmax, ymax)."""
        p = self.pred_format
        if data_generator_mode not in ("resize", "pad"):
            raise ValueError("`data_generator_mode` can be either of 'resize' or 'pad', but received '{}'."
                             .format(data_generator_mode))
        generator = self.data_generator.generate(batch_size=batch_size, shuffle=False, transformations=[],
                                                 label_encoder=None,
                                                 returns={"processed_images", "image_ids", "inverse_transform"},
                                                 keep_images_without_gt=True, degenerate_box_handling="remove")
        if self.data_generator.image_ids is None:
            self.data_generator.image_ids = list(range(self.data_generator.get_dataset_size()))
        results = [list() for _ in range(self.n_classes + 1)]
        n_images = self.data_generator.get_dataset_size()
        n_batches = int(ceil(n_images / batch_size))
        if verbose:
            print("Number of images in the evaluation dataset: {}".format(n_images))
        seen = 0
        for _ in range(n_batches):
            batch_X, batch_image_ids, batch_inverse_transforms = next(generator)
            y_pred = self.model.predict(batch_X)
            if self.model_mode == "training":
                y_pred = decode_detections(y_pred, confidence_thresh=decoding_confidence_thresh,
                                           iou_threshold=decoding_iou_threshold, top_k=decoding_top_k,
                                           input_coords=decoding_pred_coords, normalize_coords=decoding_normalize_coords,
                                           img_height=img_height, img_width=img_width,
                                           border_pixels=decoding_border_pixels)
            else:
                y_pred = [y_pred[i][y_pred[i, :, 0] != 0] for i in range(len(y_pred))]   # drop the zero padding
            y_pred = apply_inverse_transforms(y_pred, batch_inverse_transforms)
            for k, batch_item in enumerate(y_pred):
                if seen + k >= n_images:            # the last batch wraps around the dataset
                    break
                image_id = batch_image_ids[k]
                for box in np.asarray(batch_item).reshape(-1, 6):
                    conf = round(float(box[p["conf"]]), round_confidences) if round_confidences else box[p["conf"]]
                    results[int(box[p["class_id"]])].append(
                        (image_id, conf, round(float(box[p["xmin"]]), 1), round(float(box[p["ymin"]]), 1),
                         round(float(box[p["xmax"]]), 1), round(float(box[p["ymax"]]), 1)))
            seen += len(y_pred)
        self.prediction_results = results
        if ret:
            return results

    def write_predictions_to_txt(self, classes=None, out_file_prefix="comp3_det_test_", verbose=True):
        """One Pascal-VOC results file per class: `image_id confidence xmin ymin xmax ymax` rows."""
        if self.prediction_results is None:
            raise ValueError("There are no prediction results. You must run `predict_on_dataset()` before calling this method.")
        for class_id in range(1, self.n_classes + 1):
            suffix = "{:04d}".format(class_id) if classes is None else classes[class_id]
            with open("{}{}.txt".format(out_file_prefix, suffix), "w") as f:
                for pred in self.prediction_results[class_id]:
                    row = list(pred)
                    row[0] = "{:06d}".format(int(row[0])) if str(row[0]).isdigit() else str(row[0])
                    row[1] = round(float(row[1]), 4)
                    f.write(" ".join(map(str, row)) + "\n")
        if verbose:
            print("All results files saved.")

    # ---- ground truth ---------------------------------------------------------------------------------------------
    def _image_labels(self, i):
        g = self.gt_format
        labels = np.asarray(self.data_generator.labels[i], dtype=float).reshape(-1, 5)
        keep = np.ones(len(labels), dtype=bool)
        if self.ignore_under_area > 0 and len(labels):
            area = (labels[:, g["ymax"]] - labels[:, g["ymin"]]) * (labels[:, g["xmax"]] - labels[:, g["xmin"]])
            keep = area >= self.ignore_under_area
        return labels, keep

    def get_num_gt_per_class(self, ignore_neutral_boxes=True, verbose=True, ret=False):
        if self.data_generator.labels is None:
            raise ValueError("Computing the number of ground truth boxes per class not possible, no ground truth given.")
        counts = np.zeros(self.n_classes + 1, dtype=int)
        neutral_known = getattr(self.data_generator, "eval_neutral", None) is not None
        for i in range(len(self.data_generator.labels)):
            labels, keep = self._image_labels(i)
            boxes = labels[keep]
            for j in range(boxes.shape[0]):
                # (the reference indexes eval_neutral with the position in the area-filtered list, :545; kept)
                if ignore_neutral_boxes and neutral_known and self.data_generator.eval_neutral[i][j]:
                    continue
                counts[int(boxes[j, self.gt_format["class_id"]])] += 1
        self.num_gt_per_class = counts
        if ret:
            return counts

    # ---- matching -------------------------------------------------------------------------------------------------
    def match_predictions(self, ignore_neutral_boxes=True, matching_iou_threshold=0.5, border_pixels="include",
                          sorting_algorithm="quicksort", verbose=True, ret=False):
        if self.data_generator.labels is None:
            raise ValueError("Matching predictions to ground truth boxes not possible, no ground truth given.")
        if self.prediction_results is None:
            raise ValueError("There are no prediction results. You must run `predict_on_dataset()` before calling this method.")
        g = self.gt_format
        neutral_known = getattr(self.data_generator, "eval_neutral", None) is not None
        use_neutral = ignore_neutral_boxes and neutral_known
        ground_truth = {}
        for i, image_id in enumerate(self.data_generator.image_ids):
            labels, keep = self._image_labels(i)
            neutral = (np.asarray(self.data_generator.eval_neutral[i], dtype=bool) if use_neutral
                       else np.zeros(len(labels), dtype=bool))
            ground_truth[str(image_id)] = (labels[keep], neutral[:len(labels)][keep] if len(neutral) >= len(labels)
                                           else np.zeros(int(keep.sum()), dtype=bool))
        true_positives, false_positives = [[]], [[]]
        cumulative_true_positives, cumulative_false_positives = [[]], [[]]
        for class_id in range(1, self.n_classes + 1):
            preds = self.prediction_results[class_id]
            true_pos = np.zeros(len(preds), dtype=int)
            false_pos = np.zeros(len(preds), dtype=int)
            if len(preds) == 0:
                if verbose:
                    print("No predictions for class {}/{}".format(class_id, self.n_classes))
                true_positives.append(true_pos)
                false_positives.append(false_pos)
                # (reference :664-667 appends nothing to the cumulative lists here; its later indexing by class id
                # then breaks -- empty arrays are appended instead)
                cumulative_true_positives.append(np.cumsum(true_pos))
                cumulative_false_positives.append(np.cumsum(false_pos))
                continue
            conf = np.array([p[1] for p in preds], dtype=np.float32)
            boxes = np.array([p[2:6] for p in preds], dtype=np.float32).astype(float)
            order = np.argsort(-conf, kind=sorting_algorithm)
            gt_matched = {}
            for rank, idx in enumerate(order):
                image_id = str(preds[idx][0])
                gt, neutral = ground_truth[image_id]
                mask = gt[:, g["class_id"]] == class_id if len(gt) else np.zeros(0, dtype=bool)
                gt_c, neutral_c = gt[mask], neutral[mask]
                if gt_c.size == 0:
                    false_pos[rank] = 1
                    continue
                overlaps = iou(gt_c[:, [g["xmin"], g["ymin"], g["xmax"], g["ymax"]]], boxes[idx], coords="corners",
                               mode="element-wise", border_pixels=border_pixels)
                best = int(np.argmax(overlaps))
                if overlaps[best] < matching_iou_threshold:
                    false_pos[rank] = 1
                elif not (use_neutral and neutral_c[best]):
                    taken = gt_matched.setdefault(image_id, np.zeros(gt_c.shape[0], dtype=bool))
                    if not taken[best]:
                        true_pos[rank] = 1
                        taken[best] = True
                    else:
                        false_pos[rank] = 1      # duplicate detection of an already detected object
                # else: matched a neutral ('difficult') box: neither true nor false positive
            true_positives.append(true_pos)
            false_positives.append(false_pos)
            cumulative_true_positives.append(np.cumsum(true_pos))
            cumulative_false_positives.append(np.cumsum(false_pos))
        self.true_positives, self.false_positives = true_positives, false_positives
        self.cumulative_true_positives = cumulative_true_positives
        self.cumulative_false_positives = cumulative_false_positives
        if ret:
            return true_positives, false_positives, cumulative_true_positives, cumulative_false_positives

    # ---- precision / recall / AP ------------------------------------------------------------------------------------
    def compute_precision_recall(self, verbose=True, ret=False):
        if self.cumulative_true_positives is None or self.cumulative_false_positives is None:
            raise ValueError("True and false positives not available. You must run `match_predictions()` before you call this method.")
        if self.num_gt_per_class is None:
            raise ValueError("Number of ground truth boxes per class not available. You must run `get_num_gt_per_class()` before you call this method.")
        cumulative_precisions, cumulative_recalls = [[]], [[]]
        for class_id in range(1, self.n_classes + 1):
            tp = np.asarray(self.cumulative_true_positives[class_id], dtype=float)
            fp = np.asarray(self.cumulative_false_positives[class_id], dtype=float)
            with np.errstate(divide="ignore", invalid="ignore"):
                cumulative_precisions.append(np.where(tp + fp > 0, tp / (tp + fp), 0))
                cumulative_recalls.append(tp / self.num_gt_per_class[class_id])
        self.cumulative_precisions, self.cumulative_recalls = cumulative_precisions, cumulative_recalls
        if ret:
            return cumulative_precisions, cumulative_recalls

    def compute_average_precisions(self, mode="sample", num_recall_points=11, verbose=True, ret=False):
        if self.cumulative_precisions is None or self.cumulative_recalls is None:
            raise ValueError("Precisions and recalls not available. You must run `compute_precision_recall()` before you call this method.")
        if mode not in {"sample", "integrate"}:
            raise ValueError("`mode` can be either 'sample' or 'integrate', but received '{}'".format(mode))
        average_precisions = [0.0]
        for class_id in range(1, self.n_classes + 1):
            prec = np.asarray(self.cumulative_precisions[class_id], dtype=float)
            rec = np.asarray(self.cumulative_recalls[class_id], dtype=float)
            ap = 0.0
            if mode == "sample":         # VOC <= 2009: mean over t of max{precision : recall >= t}
                for t in np.linspace(0, 1, num_recall_points, endpoint=True):
                    sel = prec[rec >= t]
                    ap += np.amax(sel) if sel.size else 0.0
                ap /= num_recall_points
            elif rec.size:               # VOC >= 2010: area under the monotone envelope, over the recall values seen
                uniq, first = np.unique(rec, return_index=True)
                env = np.zeros_like(uniq)
                width = np.zeros_like(uniq)
                for i in range(len(uniq) - 2, -1, -1):
                    env[i] = max(np.amax(prec[first[i]:first[i + 1]]), env[i + 1])
                    width[i] = uniq[i + 1] - uniq[i]
                ap = float(np.sum(env * width))
            average_precisions.append(ap)
        self.average_precisions = average_precisions
        if ret:
            return average_precisions

    def compute_mean_average_precision(self, ret=True):
        if self.average_precisions is None:
            raise ValueError("Average precisions not available. You must run `compute_average_precisions()` before you call this method.")
        self.mean_average_precision = np.average(self.average_precisions[1:])
        if ret:
            return self.mean_average_precision


def apply_inverse_transforms(y_pred_decoded, inverse_transforms):
    """Undo the generator's geometric transformations on decoded boxes (data_generator/object_detection_2d_misc_utils.py:
    `apply_inverse_transforms`): each image has a list of callables applied in reverse order; None = identity."""
    if inverse_transforms is None:
        return y_pred_decoded
    out = []
    for boxes, chain in zip(y_pred_decoded, inverse_transforms):
        boxes = np.copy(boxes)
        for inverter in (chain or [])[::-1]:
            if inverter is not None:
                boxes = inverter(boxes)
        out.append(boxes)
    return out

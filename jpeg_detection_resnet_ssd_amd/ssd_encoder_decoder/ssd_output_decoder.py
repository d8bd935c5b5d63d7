"""Host-side (numpy) decoding of raw SSD predictions: offsets -> absolute boxes, per-class confidence threshold and
greedy NMS, top-k.  Same function names and arguments as localisation_part/ssd_encoder_decoder/ssd_output_decoder.py
(`greedy_nms` :27-75, `_greedy_nms` :77-92, `decode_detections` :111-226, `decode_detections_fast` :228-340)."""
import numpy as np

from ..bounding_box_utils.bounding_box_utils import convert_coordinates, iou


def _nms(rows, score_col, box_start, iou_threshold, coords, border_pixels):
    left = np.copy(rows)
    kept = []
    while left.shape[0] > 0:
        i = int(np.argmax(left[:, score_col]))
        best = np.copy(left[i])
        kept.append(best)
        left = np.delete(left, i, axis=0)
        if left.shape[0] == 0:
            break
        sim = iou(left[:, box_start:], best[box_start:], coords=coords, mode="element-wise", border_pixels=border_pixels)
        left = left[sim <= iou_threshold]
    return np.array(kept)


def greedy_nms(y_pred_decoded, iou_threshold=0.45, coords="corners", border_pixels="half"):
    """Per batch item: rows [class_id, score, 4 coords] -> the rows surviving greedy non-maximum suppression."""
    return [_nms(item, 1, 2, iou_threshold, coords, border_pixels) for item in y_pred_decoded]


def _greedy_nms(predictions, iou_threshold=0.45, coords="corners", border_pixels="half"):
    """Rows [score, 4 coords] of ONE class."""
    return _nms(predictions, 0, 1, iou_threshold, coords, border_pixels)


def decode_detections(y_pred, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, input_coords="centroids",
                      normalize_coords=True, img_height=None, img_width=None, border_pixels="half"):
    """(batch, #boxes, #classes + 12) -> list of (k_i, 6) arrays [class_id, confidence, xmin, ymin, xmax, ymax]."""
    if normalize_coords and ((img_height is None) or (img_width is None)):
        raise ValueError("If relative box coordinates are supposed to be converted to absolute coordinates, the decoder "
                         "needs the image size in order to decode the predictions, but `img_height == {}` and "
                         "`img_width == {}`".format(img_height, img_width))
    raw = np.copy(y_pred[:, :, :-8])
    anc, var = y_pred[:, :, -8:-4], y_pred[:, :, -4:]
    if input_coords == "centroids":
        raw[:, :, [-2, -1]] = np.exp(raw[:, :, [-2, -1]] * var[:, :, [2, 3]]) * anc[:, :, [2, 3]]
        raw[:, :, [-4, -3]] = raw[:, :, [-4, -3]] * var[:, :, [0, 1]] * anc[:, :, [2, 3]] + anc[:, :, [0, 1]]
        raw = convert_coordinates(raw, start_index=-4, conversion="centroids2corners")
    elif input_coords == "minmax":
        raw[:, :, -4:] *= var
        raw[:, :, [-4, -3]] *= np.expand_dims(anc[:, :, 1] - anc[:, :, 0], axis=-1)
        raw[:, :, [-2, -1]] *= np.expand_dims(anc[:, :, 3] - anc[:, :, 2], axis=-1)
        raw[:, :, -4:] += anc
        raw = convert_coordinates(raw, start_index=-4, conversion="minmax2corners")
    elif input_coords == "corners":
        raw[:, :, -4:] *= var
        raw[:, :, [-4, -2]] *= np.expand_dims(anc[:, :, 2] - anc[:, :, 0], axis=-1)
        raw[:, :, [-3, -1]] *= np.expand_dims(anc[:, :, 3] - anc[:, :, 1], axis=-1)
        raw[:, :, -4:] += anc
    else:
        raise ValueError("Unexpected value for `input_coords`. Supported input coordinate formats are 'minmax', "
                         "'corners' and 'centroids'.")
    if normalize_coords:
        raw[:, :, [-4, -2]] *= img_width
        raw[:, :, [-3, -1]] *= img_height
    n_classes = raw.shape[-1] - 4
    out = []
    for item in raw:
        pred = []
        for class_id in range(1, n_classes):
            single = item[:, [class_id, -4, -3, -2, -1]]
            met = single[single[:, 0] > confidence_thresh]
            if met.shape[0] > 0:
                maxima = _greedy_nms(met, iou_threshold=iou_threshold, coords="corners", border_pixels=border_pixels)
                rows = np.zeros((maxima.shape[0], 6))
                rows[:, 0] = class_id
                rows[:, 1:] = maxima
                pred.append(rows)
        if pred:
            pred = np.concatenate(pred, axis=0)
            if top_k != "all" and pred.shape[0] > top_k:
                keep = np.argpartition(pred[:, 1], kth=pred.shape[0] - top_k, axis=0)[pred.shape[0] - top_k:]
                pred = pred[keep]
        else:
            pred = np.array(pred)
        out.append(pred)
    return out


def decode_detections_fast(y_pred, confidence_thresh=0.5, iou_threshold=0.45, top_k="all", input_coords="centroids",
                           normalize_coords=True, img_height=None, img_width=None, border_pixels="half"):
    """The cheaper decoder: each box keeps its arg-max class only, background boxes are dropped, then threshold,
    one NMS over all classes together (if `iou_threshold` is given) and top-k.  -> list of (k_i, 6) arrays."""
    if normalize_coords and ((img_height is None) or (img_width is None)):
        raise ValueError("If relative box coordinates are supposed to be converted to absolute coordinates, the decoder "
                         "needs the image size in order to decode the predictions, but `img_height == {}` and "
                         "`img_width == {}`".format(img_height, img_width))
    conv = np.copy(y_pred[:, :, -14:-8])
    conv[:, :, 0] = np.argmax(y_pred[:, :, :-12], axis=-1)
    conv[:, :, 1] = np.amax(y_pred[:, :, :-12], axis=-1)
    anc, var = y_pred[:, :, -8:-4], y_pred[:, :, -4:]
    if input_coords == "centroids":
        conv[:, :, [4, 5]] = np.exp(conv[:, :, [4, 5]] * var[:, :, [2, 3]]) * anc[:, :, [2, 3]]
        conv[:, :, [2, 3]] = conv[:, :, [2, 3]] * var[:, :, [0, 1]] * anc[:, :, [2, 3]] + anc[:, :, [0, 1]]
        conv = convert_coordinates(conv, start_index=-4, conversion="centroids2corners")
    elif input_coords == "minmax":
        conv[:, :, 2:] *= var
        conv[:, :, [2, 3]] *= np.expand_dims(anc[:, :, 1] - anc[:, :, 0], axis=-1)
        conv[:, :, [4, 5]] *= np.expand_dims(anc[:, :, 3] - anc[:, :, 2], axis=-1)
        conv[:, :, 2:] += anc
        conv = convert_coordinates(conv, start_index=-4, conversion="minmax2corners")
    elif input_coords == "corners":
        conv[:, :, 2:] *= var
        conv[:, :, [2, 4]] *= np.expand_dims(anc[:, :, 2] - anc[:, :, 0], axis=-1)
        conv[:, :, [3, 5]] *= np.expand_dims(anc[:, :, 3] - anc[:, :, 1], axis=-1)
        conv[:, :, 2:] += anc
    else:
        raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")
    if normalize_coords:
        conv[:, :, [2, 4]] *= img_width
        conv[:, :, [3, 5]] *= img_height
    out = []
    for item in conv:
        boxes = item[np.nonzero(item[:, 0])]
        boxes = boxes[boxes[:, 1] >= confidence_thresh]
        if iou_threshold and boxes.shape[0] > 0:
            boxes = _nms(boxes, 1, 2, iou_threshold, "corners", border_pixels)
        if top_k != "all" and boxes.shape[0] > top_k:
            keep = np.argpartition(boxes[:, 1], kth=boxes.shape[0] - top_k, axis=0)[boxes.shape[0] - top_k:]
            boxes = boxes[keep]
        out.append(boxes)
    return out

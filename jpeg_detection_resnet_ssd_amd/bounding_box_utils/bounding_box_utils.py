"""Axis-aligned box helpers used by AnchorBoxes and SSDInputEncoder on the host (numpy).
Same functions, argument names and conventions as
localisation_part/bounding_box_utils/bounding_box_utils.py:24-87 (convert_coordinates), :119-281
(intersection_area) and :283-383 (iou); written broadcast-style instead of tiling."""
import numpy as np

_BORDER = {"half": 0, "include": 1, "exclude": -1}

# conversion name -> function(four input columns) -> four output columns
_CONVERSIONS = {
    "minmax2centroids": lambda a, b, c, d, e: ((a + b) / 2.0, (c + d) / 2.0, b - a + e, d - c + e),
    "centroids2minmax": lambda a, b, c, d, e: (a - c / 2.0, a + c / 2.0, b - d / 2.0, b + d / 2.0),
    "corners2centroids": lambda a, b, c, d, e: ((a + c) / 2.0, (b + d) / 2.0, c - a + e, d - b + e),
    "centroids2corners": lambda a, b, c, d, e: (a - c / 2.0, b - d / 2.0, a + c / 2.0, b + d / 2.0),
    "minmax2corners": lambda a, b, c, d, e: (a, c, b, d),
    "corners2minmax": lambda a, b, c, d, e: (a, c, b, d),
}


def convert_coordinates(tensor, start_index, conversion, border_pixels="half"):
    """Copy of `tensor` (as float) with the four coordinates starting at `start_index` of the last
    axis converted between 'minmax' (xmin,xmax,ymin,ymax), 'corners' (xmin,ymin,xmax,ymax) and
    'centroids' (cx,cy,w,h)."""
    if conversion not in _CONVERSIONS:
        raise ValueError("Unexpected conversion value. Supported values are 'minmax2centroids', 'centroids2minmax', "
                         "'corners2centroids', 'centroids2corners', 'minmax2corners', and 'corners2minmax'.")
    e = _BORDER[border_pixels]
    src = np.asarray(tensor)
    out = np.copy(src).astype(float)
    i = start_index
    cols = _CONVERSIONS[conversion](src[..., i], src[..., i + 1], src[..., i + 2], src[..., i + 3], e)
    for j, col in enumerate(cols):
        out[..., i + j] = col
    return out


def _as_corner_columns(boxes, coords):
    if coords == "centroids":
        boxes = convert_coordinates(boxes, 0, "centroids2corners")
        coords = "corners"
    if coords == "corners":
        return boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    if coords == "minmax":
        return boxes[:, 0], boxes[:, 2], boxes[:, 1], boxes[:, 3]
    raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")


def _check(boxes1, boxes2, mode):
    boxes1, boxes2 = np.asarray(boxes1), np.asarray(boxes2)
    if boxes1.ndim > 2:
        raise ValueError("boxes1 must have rank either 1 or 2, but has rank {}.".format(boxes1.ndim))
    if boxes2.ndim > 2:
        raise ValueError("boxes2 must have rank either 1 or 2, but has rank {}.".format(boxes2.ndim))
    if boxes1.ndim == 1:
        boxes1 = boxes1[None, :]
    if boxes2.ndim == 1:
        boxes2 = boxes2[None, :]
    if not (boxes1.shape[1] == boxes2.shape[1] == 4):
        raise ValueError("All boxes must consist of 4 coordinates, but the boxes in `boxes1` and `boxes2` have {} and "
                         "{} coordinates, respectively.".format(boxes1.shape[1], boxes2.shape[1]))
    if mode not in ("outer_product", "element-wise"):
        raise ValueError("`mode` must be one of 'outer_product' and 'element-wise', but got '{}'.".format(mode))
    return boxes1, boxes2


def _intersection(c1, c2, mode, d):
    x0a, y0a, x1a, y1a = c1
    x0b, y0b, x1b, y1b = c2
    if mode == "outer_product":
        x0a, y0a, x1a, y1a = (v[:, None] for v in (x0a, y0a, x1a, y1a))
        x0b, y0b, x1b, y1b = (v[None, :] for v in (x0b, y0b, x1b, y1b))
    w = np.maximum(0, np.minimum(x1a, x1b) - np.maximum(x0a, x0b) + d)
    h = np.maximum(0, np.minimum(y1a, y1b) - np.maximum(y0a, y0b) + d)
    return w * h


def intersection_area(boxes1, boxes2, coords="centroids", mode="outer_product", border_pixels="half"):
    boxes1, boxes2 = _check(boxes1, boxes2, mode)
    return _intersection(_as_corner_columns(boxes1, coords), _as_corner_columns(boxes2, coords), mode,
                         _BORDER[border_pixels])


def iou(boxes1, boxes2, coords="centroids", mode="outer_product", border_pixels="half"):
    """Jaccard similarity, (m, n) in 'outer_product' mode, (m,) in 'element-wise' mode.  As in the
    reference (bounding_box_utils.py:345) the intersection always uses the 'half' border rule; the
    areas use `border_pixels`."""
    boxes1, boxes2 = _check(boxes1, boxes2, mode)
    c1, c2 = _as_corner_columns(boxes1, coords), _as_corner_columns(boxes2, coords)
    inter = _intersection(c1, c2, mode, 0)
    d = _BORDER[border_pixels]
    a1 = (c1[2] - c1[0] + d) * (c1[3] - c1[1] + d)
    a2 = (c2[2] - c2[0] + d) * (c2[3] - c2[1] + d)
    if mode == "outer_product":
        a1, a2 = a1[:, None], a2[None, :]
    return inter / (a1 + a2 - inter)

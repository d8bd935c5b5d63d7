"""Anchor (prior) box arithmetic shared by the AnchorBoxes layer and SSDInputEncoder -- numpy, host side.
Restates localisation_part/keras_layers/keras_layer_AnchorBoxes.py:150-248 ==
localisation_part/ssd_encoder_decoder/ssd_input_encoder.py:456-543 (the two are the same algorithm)."""
import numpy as np

from .bounding_box_utils import convert_coordinates


def anchor_boxes_for_map(img_height, img_width, fm_height, fm_width, this_scale, next_scale, aspect_ratios,
                         two_boxes_for_ar1, this_steps, this_offsets, clip_boxes, variances, coords,
                         normalize_coords):
    """(fm_height, fm_width, n_boxes, 8) float64: 4 box coordinates in `coords` format + 4 variances."""
    size = min(img_height, img_width)
    wh = []
    for ar in aspect_ratios:
        if ar == 1:
            wh.append((this_scale * size, this_scale * size))
            if two_boxes_for_ar1:
                s = np.sqrt(this_scale * next_scale) * size
                wh.append((s, s))
        else:
            wh.append((this_scale * size * np.sqrt(ar), this_scale * size / np.sqrt(ar)))
    wh = np.array(wh)
    n_boxes = len(wh)

    def two(v, default):
        if v is None:
            return default
        if isinstance(v, (list, tuple)) and len(v) == 2:
            return v[0], v[1]
        return v, v

    step_h, step_w = two(this_steps, (img_height / fm_height, img_width / fm_width))
    off_h, off_w = two(this_offsets, (0.5, 0.5))
    cy = np.linspace(off_h * step_h, (off_h + fm_height - 1) * step_h, fm_height)
    cx = np.linspace(off_w * step_w, (off_w + fm_width - 1) * step_w, fm_width)
    cx_grid, cy_grid = np.meshgrid(cx, cy)
    boxes = np.zeros((fm_height, fm_width, n_boxes, 4))
    boxes[..., 0] = cx_grid[..., None]
    boxes[..., 1] = cy_grid[..., None]
    boxes[..., 2] = wh[:, 0]
    boxes[..., 3] = wh[:, 1]
    boxes = convert_coordinates(boxes, start_index=0, conversion="centroids2corners")
    if clip_boxes:
        xs = boxes[..., [0, 2]]
        xs[xs >= img_width] = img_width - 1
        xs[xs < 0] = 0
        boxes[..., [0, 2]] = xs
        ys = boxes[..., [1, 3]]
        ys[ys >= img_height] = img_height - 1
        ys[ys < 0] = 0
        boxes[..., [1, 3]] = ys
    if normalize_coords:
        boxes[..., [0, 2]] /= img_width
        boxes[..., [1, 3]] /= img_height
    if coords == "centroids":
        boxes = convert_coordinates(boxes, start_index=0, conversion="corners2centroids", border_pixels="half")
    elif coords == "minmax":
        boxes = convert_coordinates(boxes, start_index=0, conversion="corners2minmax", border_pixels="half")
    var = np.zeros_like(boxes) + np.asarray(variances, dtype=np.float64)
    return np.concatenate((boxes, var), axis=-1)

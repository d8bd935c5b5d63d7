"""SSDInputEncoder: ground-truth boxes -> `y_true (batch, #boxes, #classes + 12)`, on the host in numpy.
Same constructor, argument validation, anchor order and encoding rules as
localisation_part/ssd_encoder_decoder/ssd_input_encoder.py:36-611 (the producer of the tensor the
multibox loss consumes).  Layout of the last axis:
[one-hot classes (background first) | 4 encoded offsets | 4 anchor coords | 4 variances]."""
import numpy as np

from ..bounding_box_utils.anchor_boxes import anchor_boxes_for_map
from ..bounding_box_utils.bounding_box_utils import convert_coordinates, iou
from .matching_utils import match_bipartite_greedy, match_multi


class DegenerateBoxError(Exception):
    pass


class SSDInputEncoder:
    def __init__(self, img_height, img_width, n_classes, predictor_sizes, min_scale=0.1, max_scale=0.9, scales=None,
                 aspect_ratios_global=[0.5, 1.0, 2.0], aspect_ratios_per_layer=None, two_boxes_for_ar1=True,
                 steps=None, offsets=None, clip_boxes=False, variances=[0.1, 0.1, 0.2, 0.2], matching_type="multi",
                 pos_iou_threshold=0.5, neg_iou_limit=0.3, border_pixels="half", coords="centroids",
                 normalize_coords=True, background_id=0):
        predictor_sizes = np.array(predictor_sizes)
        if predictor_sizes.ndim == 1:
            predictor_sizes = np.expand_dims(predictor_sizes, axis=0)
        n_layers = predictor_sizes.shape[0]
        if (min_scale is None or max_scale is None) and scales is None:
            raise ValueError("Either `min_scale` and `max_scale` or `scales` need to be specified.")
        if scales:
            if len(scales) != n_layers + 1:
                raise ValueError("It must be either scales is None or len(scales) == len(predictor_sizes)+1, but "
                                 "len(scales) == {} and len(predictor_sizes)+1 == {}".format(len(scales), n_layers + 1))
            scales = np.array(scales)
            if np.any(scales <= 0):
                raise ValueError("All values in `scales` must be greater than 0, but the passed list of scales is {}"
                                 .format(scales))
        elif not 0 < min_scale <= max_scale:
            raise ValueError("It must be 0 < min_scale <= max_scale, but it is min_scale = {} and max_scale = {}"
                             .format(min_scale, max_scale))
        if aspect_ratios_per_layer is not None:
            if len(aspect_ratios_per_layer) != n_layers:
                raise ValueError("It must be either aspect_ratios_per_layer is None or len(aspect_ratios_per_layer) == "
                                 "len(predictor_sizes), but len(aspect_ratios_per_layer) == {} and len(predictor_sizes) "
                                 "== {}".format(len(aspect_ratios_per_layer), n_layers))
            for ars in aspect_ratios_per_layer:
                if np.any(np.array(ars) <= 0):
                    raise ValueError("All aspect ratios must be greater than zero.")
        else:
            if aspect_ratios_global is None:
                raise ValueError("At least one of `aspect_ratios_global` and `aspect_ratios_per_layer` must not be `None`.")
            if np.any(np.array(aspect_ratios_global) <= 0):
                raise ValueError("All aspect ratios must be greater than zero.")
        if len(variances) != 4:
            raise ValueError("4 variance values must be pased, but {} values were received.".format(len(variances)))
        variances = np.array(variances)
        if np.any(variances <= 0):
            raise ValueError("All variances must be >0, but the variances given are {}".format(variances))
        if coords not in ("minmax", "centroids", "corners"):
            raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")
        if steps is not None and len(steps) != n_layers:
            raise ValueError("You must provide at least one step value per predictor layer.")
        if offsets is not None and len(offsets) != n_layers:
            raise ValueError("You must provide at least one offset value per predictor layer.")

        self.img_height, self.img_width = img_height, img_width
        self.n_classes = n_classes + 1
        self.predictor_sizes = predictor_sizes
        self.min_scale, self.max_scale = min_scale, max_scale
        self.scales = np.linspace(min_scale, max_scale, n_layers + 1) if scales is None else scales
        self.aspect_ratios = ([aspect_ratios_global] * n_layers if aspect_ratios_per_layer is None
                              else aspect_ratios_per_layer)
        self.two_boxes_for_ar1 = two_boxes_for_ar1
        self.steps = steps if steps is not None else [None] * n_layers
        self.offsets = offsets if offsets is not None else [None] * n_layers
        self.clip_boxes = clip_boxes
        self.variances = variances
        self.matching_type = matching_type
        self.pos_iou_threshold = pos_iou_threshold
        self.neg_iou_limit = neg_iou_limit
        self.border_pixels = border_pixels
        self.coords = coords
        self.normalize_coords = normalize_coords
        self.background_id = background_id
        if aspect_ratios_per_layer is not None:
            self.n_boxes = [len(a) + 1 if (1 in a) and two_boxes_for_ar1 else len(a) for a in aspect_ratios_per_layer]
        else:
            self.n_boxes = (len(aspect_ratios_global) + 1 if (1 in aspect_ratios_global) and two_boxes_for_ar1
                            else len(aspect_ratios_global))
        self.boxes_list = [self.generate_anchor_boxes_for_layer(self.predictor_sizes[i], self.aspect_ratios[i],
                                                                self.scales[i], self.scales[i + 1], self.steps[i],
                                                                self.offsets[i]) for i in range(n_layers)]
        self._template = None

    def generate_anchor_boxes_for_layer(self, feature_map_size, aspect_ratios, this_scale, next_scale, this_steps=None,
                                        this_offsets=None, diagnostics=False):
        full = anchor_boxes_for_map(self.img_height, self.img_width, int(feature_map_size[0]), int(feature_map_size[1]),
                                    this_scale, next_scale, aspect_ratios, self.two_boxes_for_ar1, this_steps,
                                    this_offsets, self.clip_boxes, self.variances, self.coords, self.normalize_coords)
        return full[..., :4]

    def generate_encoding_template(self, batch_size, diagnostics=False):
        if self._template is None:
            boxes = np.concatenate([b.reshape(-1, 4) for b in self.boxes_list], axis=0)
            classes = np.zeros((boxes.shape[0], self.n_classes))
            var = np.zeros_like(boxes) + self.variances
            self._template = np.concatenate((classes, boxes, boxes, var), axis=1)
        return np.tile(self._template[None], (batch_size, 1, 1))

    def __call__(self, ground_truth_labels, diagnostics=False):
        class_id, xmin, ymin, xmax, ymax = 0, 1, 2, 3, 4
        batch_size = len(ground_truth_labels)
        y = self.generate_encoding_template(batch_size)
        y[:, :, self.background_id] = 1
        eye = np.eye(self.n_classes)
        for i in range(batch_size):
            gt = np.asarray(ground_truth_labels[i])
            if gt.size == 0:
                continue
            labels = gt.astype(float)
            if np.any(labels[:, xmax] - labels[:, xmin] <= 0) or np.any(labels[:, ymax] - labels[:, ymin] <= 0):
                raise DegenerateBoxError(
                    "SSDInputEncoder detected degenerate ground truth bounding boxes for batch item {} with bounding "
                    "boxes {}, i.e. bounding boxes where xmax <= xmin and/or ymax <= ymin. Degenerate ground truth "
                    "bounding boxes will lead to NaN errors during the training.".format(i, labels))
            if self.normalize_coords:
                labels[:, [ymin, ymax]] /= self.img_height
                labels[:, [xmin, xmax]] /= self.img_width
            if self.coords == "centroids":
                labels = convert_coordinates(labels, xmin, "corners2centroids", self.border_pixels)
            elif self.coords == "minmax":
                labels = convert_coordinates(labels, xmin, "corners2minmax")
            one_hot = np.concatenate([eye[labels[:, class_id].astype(int)], labels[:, [xmin, ymin, xmax, ymax]]], axis=-1)
            sim = iou(labels[:, [xmin, ymin, xmax, ymax]], y[i, :, -12:-8], coords=self.coords, mode="outer_product",
                      border_pixels=self.border_pixels)
            first = match_bipartite_greedy(sim)
            y[i, first, :-8] = one_hot
            sim[:, first] = 0
            if self.matching_type == "multi":
                gts, anchors = match_multi(sim, self.pos_iou_threshold)
                y[i, anchors, :-8] = one_hot[gts]
                sim[:, anchors] = 0
            neutral = np.nonzero(np.amax(sim, axis=0) >= self.neg_iou_limit)[0]
            y[i, neutral, self.background_id] = 0
        if self.coords == "centroids":
            y[:, :, [-12, -11]] -= y[:, :, [-8, -7]]
            y[:, :, [-12, -11]] /= y[:, :, [-6, -5]] * y[:, :, [-4, -3]]
            y[:, :, [-10, -9]] /= y[:, :, [-6, -5]]
            y[:, :, [-10, -9]] = np.log(y[:, :, [-10, -9]]) / y[:, :, [-2, -1]]
        elif self.coords == "corners":
            y[:, :, -12:-8] -= y[:, :, -8:-4]
            y[:, :, [-12, -10]] /= np.expand_dims(y[:, :, -6] - y[:, :, -8], axis=-1)
            y[:, :, [-11, -9]] /= np.expand_dims(y[:, :, -5] - y[:, :, -7], axis=-1)
            y[:, :, -12:-8] /= y[:, :, -4:]
        elif self.coords == "minmax":
            y[:, :, -12:-8] -= y[:, :, -8:-4]
            y[:, :, [-12, -11]] /= np.expand_dims(y[:, :, -7] - y[:, :, -8], axis=-1)
            y[:, :, [-10, -9]] /= np.expand_dims(y[:, :, -5] - y[:, :, -6], axis=-1)
            y[:, :, -12:-8] /= y[:, :, -4:]
        if diagnostics:
            matched = np.copy(y)
            matched[:, :, -12:-8] = 0
            return y, matched
        return y

    # ---- the same encoding on the GPU ------------------------------------------------------------------------
    def encode_on_device(self, ground_truth_labels, out=None, device=None):
        """`__call__` computed by the HIP kernel `dj_ssd_encode_targets` (float64 arithmetic in the reference's order
        of operations, so the matches are the host's): -> float32 CUDA tensor (batch, #boxes, #classes + 12), written
        into `out` when given (e.g. a plan's resident `y_true` buffer).  Only the labels (a few KB) cross PCIe."""
        import torch
        from ..engine import call
        if self.coords != "centroids":
            raise NotImplementedError("encode_on_device supports coords='centroids' (the trainer's setting) only")
        if self.matching_type not in ("multi", "bipartite"):
            raise ValueError("unknown matching_type %r" % (self.matching_type,))
        batch_size = len(ground_truth_labels)
        max_gt = max([1] + [len(g) for g in ground_truth_labels])
        labels = np.zeros((batch_size, max_gt, 5), dtype=np.float64)
        counts = np.zeros(batch_size, dtype=np.int32)
        for i, gt in enumerate(ground_truth_labels):
            gt = np.asarray(gt)
            if gt.size == 0:
                continue
            lab = gt.astype(float)
            if np.any(lab[:, 3] - lab[:, 1] <= 0) or np.any(lab[:, 4] - lab[:, 2] <= 0):
                raise DegenerateBoxError(
                    "SSDInputEncoder detected degenerate ground truth bounding boxes for batch item {} with bounding "
                    "boxes {}, i.e. bounding boxes where xmax <= xmin and/or ymax <= ymin. Degenerate ground truth "
                    "bounding boxes will lead to NaN errors during the training.".format(i, lab))
            cls = lab[:, 0].astype(int)
            if np.any(cls < 0) or np.any(cls >= self.n_classes):
                raise IndexError("class id out of range for {} classes (incl. background)".format(self.n_classes))
            labels[i, :len(lab)] = lab[:, :5]
            counts[i] = len(lab)
        if device is None:
            device = out.device if out is not None else torch.device("cuda", torch.cuda.current_device())
        key = str(device)
        cache = self.__dict__.setdefault("_device_anchors", {})
        if key not in cache:
            cache[key] = torch.from_numpy(np.ascontiguousarray(self.generate_encoding_template(1)[0, :, -8:])).to(device)
        anchors = cache[key]
        n_boxes = anchors.shape[0]
        if out is None:
            out = torch.empty(batch_size, n_boxes, self.n_classes + 12, dtype=torch.float32, device=device)
        assert out.is_contiguous() and tuple(out.shape) == (batch_size, n_boxes, self.n_classes + 12)
        lab_d = torch.from_numpy(labels).to(device, non_blocking=True)
        cnt_d = torch.from_numpy(counts).to(device, non_blocking=True)
        border = {"half": 0, "include": 1, "exclude": -1}[self.border_pixels]
        call("dj_ssd_encode_targets", lab_d, cnt_d, batch_size, max_gt, anchors, n_boxes, self.n_classes,
             int(self.img_height), int(self.img_width), int(bool(self.normalize_coords)), border,
             int(self.matching_type == "multi"), float(self.pos_iou_threshold), float(self.neg_iou_limit),
             int(self.background_id), out)
        return out



class PendingTargets(object):
    """Ground-truth boxes of one batch, to be encoded straight into the model's resident `y_true` buffer."""

    def __init__(self, encoder, ground_truth_labels):
        self.encoder = encoder
        self.ground_truth_labels = [np.asarray(g, dtype=float).reshape(-1, 5) if np.size(g) else np.zeros((0, 5))
                                    for g in ground_truth_labels]

    def __len__(self):
        return len(self.ground_truth_labels)

    @property
    def shape(self):
        return (len(self.ground_truth_labels), self.encoder.generate_encoding_template(1).shape[1],
                self.encoder.n_classes + 12)

    def encode_into(self, out):
        return self.encoder.encode_on_device(self.ground_truth_labels, out=out)

    def numpy(self):
        return self.encoder(self.ground_truth_labels)


class DeviceLabelEncoder(object):
    """Drop-in for the `label_encoder=` argument of the reference's data generators
    (localisation_part/training_dct_pascal_j2d_resnet.py:244-265): the generator thread only packs the boxes; the
    matching runs on the GPU when `Model.fit_generator / train_on_batch` uploads the batch."""

    def __init__(self, encoder):
        self.encoder = encoder

    def __call__(self, ground_truth_labels, diagnostics=False):
        if diagnostics:
            return self.encoder(ground_truth_labels, diagnostics=True)
        return PendingTargets(self.encoder, ground_truth_labels)

    def __getattr__(self, name):
        return getattr(self.encoder, name)

"""AnchorBoxes (PriorBox) layer: a constant (batch, H, W, n_boxes, 8) tensor of anchor centroids and
variances derived from the predictor map's size.  Same constructor, validation and arithmetic as
localisation_part/keras_layers/keras_layer_AnchorBoxes.py:58-255; the tensor is computed once on the
host when a plan is built and stays resident in HBM."""
import numpy as np
import torch

from ..bounding_box_utils.anchor_boxes import anchor_boxes_for_map
from ..engine import Value
from ..keras import backend as K
from ..keras.layers import InputSpec, Layer


class AnchorBoxes(Layer):
    def __init__(self, img_height, img_width, this_scale, next_scale, aspect_ratios=[0.5, 1.0, 2.0],
                 two_boxes_for_ar1=True, this_steps=None, this_offsets=None, clip_boxes=False,
                 variances=[0.1, 0.1, 0.2, 0.2], coords="centroids", normalize_coords=False, **kwargs):
        if K.backend() != "tensorflow":
            raise TypeError("This layer only supports TensorFlow at the moment, but you are using the {} backend."
                            .format(K.backend()))
        if (this_scale < 0) or (next_scale < 0) or (this_scale > 1):
            raise ValueError("`this_scale` must be in [0, 1] and `next_scale` must be >0, but `this_scale` == {}, "
                             "`next_scale` == {}".format(this_scale, next_scale))
        if len(variances) != 4:
            raise ValueError("4 variance values must be pased, but {} values were received.".format(len(variances)))
        variances = np.array(variances)
        if np.any(variances <= 0):
            raise ValueError("All variances must be >0, but the variances given are {}".format(variances))
        self.img_height = img_height
        self.img_width = img_width
        self.this_scale = this_scale
        self.next_scale = next_scale
        self.aspect_ratios = aspect_ratios
        self.two_boxes_for_ar1 = two_boxes_for_ar1
        self.this_steps = this_steps
        self.this_offsets = this_offsets
        self.clip_boxes = clip_boxes
        self.variances = variances
        self.coords = coords
        self.normalize_coords = normalize_coords
        if (1 in aspect_ratios) and two_boxes_for_ar1:
            self.n_boxes = len(aspect_ratios) + 1
        else:
            self.n_boxes = len(aspect_ratios)
        super(AnchorBoxes, self).__init__(**kwargs)

    def build(self, input_shape):
        self.input_spec = [InputSpec(shape=input_shape)]
        super(AnchorBoxes, self).build(input_shape)

    def compute_output_shape(self, input_shape):
        batch_size, feature_map_height, feature_map_width, feature_map_channels = input_shape
        return (batch_size, feature_map_height, feature_map_width, self.n_boxes, 8)

    def get_config(self):
        config = {
            "img_height": self.img_height, "img_width": self.img_width, "this_scale": self.this_scale,
            "next_scale": self.next_scale, "aspect_ratios": list(self.aspect_ratios),
            "two_boxes_for_ar1": self.two_boxes_for_ar1, "clip_boxes": self.clip_boxes,
            "variances": list(self.variances), "coords": self.coords, "normalize_coords": self.normalize_coords,
        }
        base_config = super(AnchorBoxes, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))

    def anchors(self):
        _, fm_h, fm_w, _ = self.input_shape
        return anchor_boxes_for_map(self.img_height, self.img_width, fm_h, fm_w, self.this_scale, self.next_scale,
                                    self.aspect_ratios, self.two_boxes_for_ar1, self.this_steps, self.this_offsets,
                                    self.clip_boxes, self.variances, self.coords, self.normalize_coords)

    def lower(self, plan, model, ins):
        a = torch.from_numpy(self.anchors().astype(np.float32))
        buf = plan.empty(plan.batch_size, *a.shape)
        buf.copy_(a.unsqueeze(0).expand(plan.batch_size, *a.shape))
        out = Value(buf, needs_grad=False, name=self.name)
        out.constant = True   # filled here once: Reshape / Concatenate downstream copy it at plan build, not per step
        return out

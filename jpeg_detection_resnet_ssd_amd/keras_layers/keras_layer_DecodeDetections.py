"""DecodeDetections layer (`mode='inference'`): raw SSD predictions -> (batch, top_k, 6) rows
[class_id, confidence, xmin, ymin, xmax, ymax], zero padded.  Same constructor and validation as
localisation_part/keras_layers/keras_layer_DecodeDetections.py:27-107; the decode / per-class NMS / top-k subgraph of
its `call` (:109-265) runs as three HIP kernels (dj_decode_detections)."""
import torch

from ..engine import Value, call, query
from ..keras import backend as K
from ..keras.layers import InputSpec, Layer, _materialised


class DecodeDetections(Layer):
    def __init__(self, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400, coords="centroids",
                 normalize_coords=True, img_height=None, img_width=None, fast=False, **kwargs):
        if K.backend() != "tensorflow":
            raise TypeError("This layer only supports TensorFlow at the moment, but you are using the {} backend."
                            .format(K.backend()))
        if normalize_coords and ((img_height is None) or (img_width is None)):
            raise ValueError("If relative box coordinates are supposed to be converted to absolute coordinates, the "
                             "decoder needs the image size in order to decode the predictions, but `img_height == {}` "
                             "and `img_width == {}`".format(img_height, img_width))
        if coords != "centroids":
            raise ValueError("The DetectionOutput layer currently only supports the 'centroids' coordinate format.")
        self.fast = bool(fast)   # True: the DecodeDetectionsFast algorithm (arg-max class per box, one class-agnostic NMS)
        self.confidence_thresh = confidence_thresh
        self.iou_threshold = iou_threshold
        self.top_k = top_k
        self.normalize_coords = normalize_coords
        self.img_height = img_height
        self.img_width = img_width
        self.coords = coords
        self.nms_max_output_size = nms_max_output_size
        super(DecodeDetections, self).__init__(**kwargs)

    def build(self, input_shape):
        self.input_spec = [InputSpec(shape=input_shape)]
        super(DecodeDetections, self).build(input_shape)

    def compute_output_shape(self, input_shape):
        batch_size, n_boxes, last_axis = input_shape
        return (batch_size, self.top_k, 6)

    def get_config(self):
        config = {"confidence_thresh": self.confidence_thresh, "iou_threshold": self.iou_threshold, "top_k": self.top_k,
                  "nms_max_output_size": self.nms_max_output_size, "coords": self.coords,
                  "normalize_coords": self.normalize_coords, "img_height": self.img_height, "img_width": self.img_width}
        base_config = super(DecodeDetections, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))

    def lower(self, plan, model, ins):
        yp = _materialised(ins[0], self.name, plan)
        assert yp.is_contiguous()
        b, n, width = yp.shape
        n_cls = width - 12
        if self.fast:
            ws = plan.empty(query("dj_decode_detections_fast_workspace_floats", b, n, self.nms_max_output_size))
        else:
            ws = plan.empty(query("dj_decode_detections_workspace_floats", b, n, n_cls, self.nms_max_output_size))
        out = plan.empty(b, self.top_k, 6)
        entry = "dj_decode_detections_fast" if self.fast else "dj_decode_detections"
        plan.emit(lambda: call(entry, yp, b, n, n_cls, float(self.confidence_thresh),
                               float(self.iou_threshold), int(self.top_k), int(self.nms_max_output_size),
                               int(bool(self.normalize_coords)), int(self.img_height or 0), int(self.img_width or 0), ws,
                               out))
        return Value(out, needs_grad=False, name=self.name)

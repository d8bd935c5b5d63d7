"""DecodeDetectionsFast layer (`mode='inference_fast'`): same constructor as
localisation_part/keras_layers/keras_layer_DecodeDetectionsFast.py:29-106; `call` (:108-215) -- arg-max class per box,
background dropped, confidence threshold, ONE class-agnostic NMS, top-k -- runs as `dj_decode_detections_fast`."""
from .keras_layer_DecodeDetections import DecodeDetections


class DecodeDetectionsFast(DecodeDetections):
    def __init__(self, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400, coords="centroids",
                 normalize_coords=True, img_height=None, img_width=None, **kwargs):
        kwargs.pop("fast", None)
        super(DecodeDetectionsFast, self).__init__(confidence_thresh=confidence_thresh, iou_threshold=iou_threshold,
                                                   top_k=top_k, nms_max_output_size=nms_max_output_size, coords=coords,
                                                   normalize_coords=normalize_coords, img_height=img_height,
                                                   img_width=img_width, fast=True, **kwargs)

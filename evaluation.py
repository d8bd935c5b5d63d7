#!/usr/bin/env python
"""Pascal-VOC mAP evaluation of a trained SSD300 ResNet50-DCT model on MI355X: the entry point of
localisation_part/evaluation.py (same positional `weights`, `--archi`, dataset flags; `-r/--ssd_resnet` is the only
model family built here), `mode='inference'` model with the on-device DecodeDetections layer, `Evaluator(...)` with the
reference's settings (evaluation.py:102-131: batch 8, 'resize', 11-point sampling, IoU 0.5, 'include' borders).
The VOC XML/JPEG dataset readers are out of scope; `--generator module:factory` plugs one in, the default is
the synthetic JPEG-DCT dataset."""
import importlib
import os
import sys
from argparse import ArgumentParser

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

parser = ArgumentParser()
parser.add_argument("weights", type=str)
parser.add_argument("-r", "--ssd_resnet", action="store_true", default=False)
for short, long_ in (("-s", "--ssd"), ("-so", "--ssd_other"), ("-sd", "--ssd_dct"), ("-sm", "--ssd_miisst"),
                     ("-smd", "--ssd_miisst_dct"), ("-p12", "--pascal_2012"), ("-p10", "--pascal_2010"),
                     ("-pv12", "--pascal_val_2012"), ("-p07", "--pascal_2007"), ("-mv", "--miisst_val"),
                     ("-mt", "--miisst_train")):
    parser.add_argument(short, long_, action="store_true", default=False)
parser.add_argument("-dp", "--dataset_path")
parser.add_argument("--archi", default="ssd_custom")
parser.add_argument("--generator", default=None, help="module:factory returning a DataGenerator-like evaluation dataset")
parser.add_argument("--synthetic_images", type=int, default=64)
parser.add_argument("--batch_size", type=int, default=8)
args = parser.parse_args()
if args.ssd or args.ssd_other or args.ssd_dct or args.ssd_miisst or args.ssd_miisst_dct:
    raise SystemExit("only the ResNet50-DCT SSD family (-r, --archi ...) is built here; the VGG models are out of scope")

from jpeg_detection_resnet_ssd_amd.eval_utils.average_precision_evaluator import Evaluator  # noqa: E402
from jpeg_detection_resnet_ssd_amd.keras import backend as K  # noqa: E402
from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD  # noqa: E402
from jpeg_detection_resnet_ssd_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from jpeg_detection_resnet_ssd_amd.models.keras_ssd300_dct_j2d_resnet import (  # noqa: E402
    ssd_resnet_EF_layers_custom, ssd_resnet_EF_layers_identical)

img_height, img_width, n_classes, model_mode = 300, 300, 20, "inference"
K.clear_session()
ssd_params = {"image_size": (img_height, img_width, 3), "n_classes": n_classes, "mode": model_mode,
              "l2_regularization": 0.0005, "scales": [0.1, 0.2, 0.37, 0.54, 0.71, 0.88, 1.05],
              "aspect_ratios_per_layer": [[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                          [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]],
              "two_boxes_for_ar1": True, "steps": [8, 16, 32, 64, 100, 300], "offsets": [0.5] * 6, "clip_boxes": False,
              "variances": [0.1, 0.1, 0.2, 0.2], "normalize_coords": True, "subtract_mean": [123, 117, 104],
              "swap_channels": [2, 1, 0], "confidence_thresh": 0.01, "iou_threshold": 0.45, "top_k": 200,
              "nms_max_output_size": 400, "archi": args.archi}
model = (ssd_resnet_EF_layers_custom if args.archi == "ssd_custom" else ssd_resnet_EF_layers_identical)(**ssd_params)
if args.weights != "random":
    model.load_weights(args.weights)
model.compile(optimizer=SGD(lr=0.001, momentum=0.9, decay=0.0, nesterov=False),
              loss=SSDLoss(neg_pos_ratio=3, alpha=1.0).compute_loss)

if args.generator:
    mod, fn = args.generator.split(":")
    dataset = getattr(importlib.import_module(mod), fn)()
else:
    from jpeg_detection_resnet_ssd_amd.data.generators import SyntheticDataGeneratorDCT
    print("Using the synthetic JPEG-DCT evaluation set (%d images)" % args.synthetic_images)
    dataset = SyntheticDataGeneratorDCT(n_images=args.synthetic_images, seed=4242)
if args.archi == "deconv":
    # the reference selects DataGeneratorDeconvDCT here; the stand-in switches its emission instead
    _generate = dataset.generate
    dataset.generate = lambda **kw: _generate(**dict(kw, deconv=True))

classes = ["background", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
           "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
evaluator = Evaluator(model=model, n_classes=n_classes, data_generator=dataset, model_mode=model_mode)
results = evaluator(img_height=img_height, img_width=img_width, batch_size=args.batch_size, data_generator_mode="resize",
                    round_confidences=False, matching_iou_threshold=0.5, border_pixels="include",
                    sorting_algorithm="quicksort", average_precision_mode="sample", num_recall_points=11,
                    ignore_neutral_boxes=True, return_precisions=True, return_recalls=True,
                    return_average_precisions=True, verbose=True)
mean_average_precision, average_precisions, precisions, recalls = results
for i in range(1, len(average_precisions)):
    print("{:<14}{:<6}{}".format(classes[i], "AP", round(average_precisions[i], 3)))
print()
print("{:<14}{:<6}{}".format("", "mAP", round(mean_average_precision, 3)))

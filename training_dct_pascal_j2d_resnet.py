#!/usr/bin/env python
"""Drop-in for localisation_part/training_dct_pascal_j2d_resnet.py: same flags (including the required-but-unused
ones), same environment variables, same model / optimizer / loss / encoder / callback set-up, running on MI355X.

    python3 training_dct_pascal_j2d_resnet.py -vd 0 --crop --p07p12 --reg --resnet --archi ssd_custom

Differences, all forced by what is absent offline: the Pascal-VOC DataGeneratorDCT (PIL / cv2 / jpeg2dct) is
replaced by a synthetic JPEG-DCT generator with the same emission contract unless `--generator module:factory` names
a user-supplied one; checkpoints are .npz name->array archives (h5py is not installed); `--weights` goes straight to
`load_weights(by_name=True)` (the reference's preceding `load_model` only prints a summary).
Multi-GPU (not in the reference: "no multi-GPU support for this part") = one process per GPU:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 training_dct_pascal_j2d_resnet.py ...
"""
import importlib
import os
from argparse import ArgumentParser
from math import ceil

parser = ArgumentParser(description="Script to train the SSD Resnet on the pascal voc dataset.")
parser.add_argument("--weights", default=None, help="The weights to load into the model")
parser.add_argument("-vd", "--visible_device", help="The device to use when training with the GPU", default="-1")
parser.add_argument("--restart", default=None, help="Wether the simulation starts from a previous save")
parser.add_argument("--archi", help="""The network architecture to use, value can be :\n
* cb5_only : CbCr and Y only go through the conv block 5 of Resnet50\n
* deconv : deconvolution architecture of Uber article\n
* up_sampling : up sampling architecture of Uber article\n
* y_cb4_cbcr_cb5 :Y go through the conv block 4 of Resnet50 and CbCr go through conv block 5\n
* ssd_custom : the extra-feature layers of SSD are removed to match dimension with full Late-concat-RFA architecture of Uber
""")
loading_check = parser.add_mutually_exclusive_group(required=True)
loading_check.add_argument("--ssd", action="store_true")
loading_check.add_argument("--resnet", action="store_true")
ssd_augmentation = parser.add_mutually_exclusive_group(required=True)
ssd_augmentation.add_argument("--crop", action="store_true")
ssd_augmentation.add_argument("--no_crop", action="store_true")
training_set = parser.add_mutually_exclusive_group(required=True)
training_set.add_argument("--p07", action="store_true")
training_set.add_argument("--p07p12", action="store_true")
regularizer = parser.add_mutually_exclusive_group(required=True)
regularizer.add_argument("--reg", action="store_true")
regularizer.add_argument("--no_reg", action="store_true")
# additions (all optional)
parser.add_argument("--generator", default=None, help="module:factory returning (train_gen_obj, val_gen_obj) with the "
                    "DataGeneratorDCT.generate()/get_dataset_size() surface; default: synthetic JPEG-DCT data")
parser.add_argument("--epochs", type=int, default=480)
parser.add_argument("--steps_per_epoch", type=int, default=1000)
parser.add_argument("--batch_size", type=int, default=32)
parser.add_argument("--synthetic_images", type=int, default=2048)
args = parser.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
if world == 1 and args.visible_device not in ("-1", ""):
    os.environ["HIP_VISIBLE_DEVICES"] = args.visible_device       # CUDA_VISIBLE_DEVICES of the reference
os.environ["CUDA_VISIBLE_DEVICES_REQUESTED"] = args.visible_device
if "LOCAL_WORK_DIR" not in os.environ:
    os.environ["LOCAL_WORK_DIR"] = "./" + args.visible_device
else:
    os.environ["LOCAL_WORK_DIR"] = os.path.join(os.environ["LOCAL_WORK_DIR"], args.visible_device)
os.makedirs(os.environ["LOCAL_WORK_DIR"], exist_ok=True)
os.environ.setdefault("EXPERIMENTS_OUTPUT_DIRECTORY", os.environ["LOCAL_WORK_DIR"])
os.makedirs(os.environ["EXPERIMENTS_OUTPUT_DIRECTORY"], exist_ok=True)
deconv = args.archi == "deconv"

import torch  # noqa: E402
from jpeg_detection_resnet_ssd_amd import dist as djdist  # noqa: E402
from jpeg_detection_resnet_ssd_amd.keras.callbacks import (CSVLogger, EarlyStopping, ModelCheckpoint,  # noqa: E402
                                                           ReduceLROnPlateau, TensorBoard, TerminateOnNaN)
from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD  # noqa: E402
from jpeg_detection_resnet_ssd_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from jpeg_detection_resnet_ssd_amd.models.keras_ssd300_dct_j2d_resnet import (ssd_resnet_EF_layers_custom,  # noqa: E402
                                                                             ssd_resnet_EF_layers_identical)
from jpeg_detection_resnet_ssd_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

rank, world, local = djdist.init_from_env()
if torch.cuda.is_available():
    torch.cuda.set_device(local)

img_height, img_width, img_channels = 300, 300, 3
n_classes = 20
scales = [0.1, 0.2, 0.37, 0.54, 0.71, 0.88, 1.05]
aspect_ratios = [[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                 [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]]
two_boxes_for_ar1 = True
steps = [8, 16, 32, 64, 100, 300]
offsets = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
clip_boxes = False
variances = [0.1, 0.1, 0.2, 0.2]
normalize_coords = True

ssd_args = {"image_size": (img_height, img_width, img_channels), "n_classes": n_classes, "mode": "training",
            "l2_regularization": 0.0005, "scales": scales, "aspect_ratios_per_layer": aspect_ratios,
            "two_boxes_for_ar1": two_boxes_for_ar1, "steps": steps, "offsets": offsets, "clip_boxes": clip_boxes,
            "variances": variances, "normalize_coords": normalize_coords, "archi": args.archi}
if args.archi == "ssd_custom":
    model = ssd_resnet_EF_layers_custom(**ssd_args)
else:
    model = ssd_resnet_EF_layers_identical(**ssd_args)

if args.restart:
    model.load_weights(args.restart, by_name=True)
elif args.weights:
    model.load_weights(args.weights, by_name=True)

sgd = SGD(lr=0.001, momentum=0.9, decay=0.0, nesterov=False)
ssd_loss = SSDLoss(neg_pos_ratio=3, alpha=1.0)
model.compile(optimizer=sgd, loss=ssd_loss.compute_loss)
if world > 1:
    dp = djdist.DataParallel(model)
    dp.broadcast_weights(0)

if args.generator:
    mod, fn = args.generator.split(":")
    train_dataset, val_dataset = getattr(importlib.import_module(mod), fn)(args)
else:
    from jpeg_detection_resnet_ssd_amd.data.generators import SyntheticDataGeneratorDCT
    train_dataset = SyntheticDataGeneratorDCT(n_images=args.synthetic_images, seed=1234 + 100000 * rank)
    val_dataset = SyntheticDataGeneratorDCT(n_images=max(args.batch_size, args.synthetic_images // 8), seed=987654)

batch_size = args.batch_size
predictor_sizes = [model.get_layer("%s_mbox_conf_%d" % (n, n_classes + 1)).output_shape[1:3]
                   for n in ("conv4_3_norm", "fc7", "conv6_2", "conv7_2", "conv8_2", "conv9_2")]
ssd_input_encoder = SSDInputEncoder(img_height=img_height, img_width=img_width, n_classes=n_classes,
                                    predictor_sizes=predictor_sizes, scales=scales,
                                    aspect_ratios_per_layer=aspect_ratios, two_boxes_for_ar1=two_boxes_for_ar1,
                                    steps=steps, offsets=offsets, clip_boxes=clip_boxes, variances=variances,
                                    matching_type="multi", pos_iou_threshold=0.5, neg_iou_limit=0.5,
                                    normalize_coords=normalize_coords)
label_encoder = ssd_input_encoder
if os.environ.get("DJ_DEVICE_ENCODER", "1") != "0":
    # same encodings (tests/test_encode_gpu.py), computed on the GPU at upload time instead of ~7 ms/image of host numpy
    from jpeg_detection_resnet_ssd_amd.ssd_encoder_decoder.ssd_input_encoder import DeviceLabelEncoder
    label_encoder = DeviceLabelEncoder(ssd_input_encoder)
train_generator = train_dataset.generate(batch_size=batch_size, shuffle=True, transformations=[],
                                         label_encoder=label_encoder,
                                         returns={"processed_images", "encoded_labels"},
                                         keep_images_without_gt=False, deconv=deconv)
val_generator = val_dataset.generate(batch_size=batch_size, shuffle=False, transformations=[],
                                     label_encoder=label_encoder, returns={"processed_images", "encoded_labels"},
                                     keep_images_without_gt=False, deconv=deconv)
train_dataset_size = train_dataset.get_dataset_size()
val_dataset_size = val_dataset.get_dataset_size()
if rank == 0:
    print("Number of images in the training dataset:\t{:>6}".format(train_dataset_size))
    print("Number of images in the validation dataset:\t{:>6}".format(val_dataset_size))

callbacks = [ReduceLROnPlateau(monitor="val_loss", factor=0.1, patience=7), TerminateOnNaN(),
             TensorBoard(log_dir=os.path.join(os.environ["LOCAL_WORK_DIR"], "./logs")),
             EarlyStopping(monitor="val_loss", min_delta=0, patience=10)]
if rank == 0:
    out_dir = os.environ["EXPERIMENTS_OUTPUT_DIRECTORY"]
    callbacks = [ModelCheckpoint(filepath=os.path.join(
        out_dir, "ssd300_pascal_07+12_epoch-{epoch:02d}_loss-{loss:.4f}_val_loss-{val_loss:.4f}.h5"),
        monitor="val_loss", verbose=1, save_best_only=True, save_weights_only=False, mode="auto", period=1),
        CSVLogger(filename=os.path.join(out_dir, "ssd300_pascal_07+12_training_log.csv"), separator=",",
                  append=True)] + callbacks

if args.restart:
    # "..._epoch-{epoch:02d}_loss-..." (the ModelCheckpoint pattern above).  The reference parses the whole path
    # (`args.restart.split('-')[1]`), which breaks on a directory name containing '-'; the file name alone is parsed here
    initial_epoch = int(os.path.basename(args.restart).split("-")[1].split("_")[0])
else:
    initial_epoch = 0
history = model.fit_generator(generator=train_generator, steps_per_epoch=args.steps_per_epoch, epochs=args.epochs,
                              callbacks=callbacks, validation_data=val_generator,
                              validation_steps=ceil(val_dataset_size / batch_size), initial_epoch=initial_epoch)

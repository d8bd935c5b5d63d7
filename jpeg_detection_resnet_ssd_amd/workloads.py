"""The SSD300 training workloads of BASELINE.json / SURVEY 8(d), assembled from the drop-in pieces:
model builder with the reference trainer's arguments (localisation_part/training_dct_pascal_j2d_resnet.py:92-156),
SGD(0.001, 0.9), SSDLoss(3, 1.0), synthetic JPEG-DCT inputs and encoder-made targets."""
import numpy as np

# algorithmic conv+deconv work, forward+dgrad+wgrad (BASELINE.md section 2 / SURVEY 8(d))
TRAIN_GFLOP_PER_IMAGE = {"ssd_custom": 41.60, "deconv": 74.43, "up_sampling": 74.36}

SSD_ARGS = dict(image_size=(300, 300, 3), n_classes=20, mode="training", l2_regularization=0.0005,
                scales=[0.1, 0.2, 0.37, 0.54, 0.71, 0.88, 1.05],
                aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                         [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                         [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]],
                two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=[0.5] * 6, clip_boxes=False,
                variances=[0.1, 0.1, 0.2, 0.2], normalize_coords=True)


def build_ssd(archi, weight_seed=42, compile_model=True):
    from .keras import backend as K
    from .keras.optimizers import SGD
    from .keras_loss_function.keras_ssd_loss import SSDLoss
    from .models.keras_ssd300_dct_j2d_resnet import ssd_resnet_EF_layers_custom, ssd_resnet_EF_layers_identical
    K.clear_session()
    K.set_random_seed(weight_seed)
    fn = ssd_resnet_EF_layers_custom if archi == "ssd_custom" else ssd_resnet_EF_layers_identical
    model, sizes = fn(archi=archi, return_predictor_sizes=True, **SSD_ARGS)
    if compile_model:
        model.compile(optimizer=SGD(lr=0.001, momentum=0.9, decay=0.0, nesterov=False),
                      loss=SSDLoss(neg_pos_ratio=3, alpha=1.0).compute_loss)
    return model, sizes


def make_encoder(sizes):
    from .ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    return SSDInputEncoder(300, 300, 20, [tuple(int(v) for v in s) for s in sizes], scales=SSD_ARGS["scales"],
                           aspect_ratios_per_layer=SSD_ARGS["aspect_ratios_per_layer"], two_boxes_for_ar1=True,
                           steps=SSD_ARGS["steps"], offsets=SSD_ARGS["offsets"], clip_boxes=False,
                           variances=SSD_ARGS["variances"], matching_type="multi", pos_iou_threshold=0.5,
                           neg_iou_limit=0.5, normalize_coords=True)


def synthetic_batch(archi, sizes, batch, seed=1234, fast=False):
    """([inputs...], y_true) as the reference's generator would yield them (float32 here)."""
    from .data import synthetic_dct as sd
    split = archi == "deconv"
    x = (sd.fast_dct_batch(batch, seed=seed, split_chroma=split) if fast
         else sd.dct_batch(batch, seed=seed, split_chroma=split))
    y = make_encoder(sizes)(sd.random_ground_truth(batch, seed=seed)).astype(np.float32)
    return x, y

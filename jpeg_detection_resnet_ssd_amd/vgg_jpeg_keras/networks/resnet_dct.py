"""ResNet50 classifiers over RGB pixels or JPEG DCT coefficients: `ResNet50RGB` and `ResNet50Custom(archi=...)`,
drop-in for classification_part/vgg_jpeg_keras/networks/resnet_dct.py:165-314 and :317-452 (same keyword
arguments, layer names, `archi` dispatch), built from this package's Keras-style layers for the MI355X engine.

DCT inputs for 224x224 images: Y (28, 28, 64) with CbCr (14, 14, 128), or Y, Cb (14, 14, 64), Cr (14, 14, 64) for
`archi="deconv"`.  `weights='imagenet'` would download resnet50_weights_tf_dim_ordering_tf_kernels.h5 from GitHub
(resnet_dct.py:295-308,434-448); there is no network here, so it raises and every run uses `weights=None` or a
local `.npz` written by `Model.save_weights`."""
import os
import warnings

from ...keras.layers import (Activation, BatchNormalization, Conv2D, Dense, GlobalAveragePooling2D, Input,
                             MaxPooling2D, ZeroPadding2D)
from ...keras.models import Model
from ...models import resnet_dct_blocks as blocks
from ...models.resnet_dct_blocks import conv_block, identity_block  # noqa: F401

_ARCHIS = {
    "late_concat_rfa_thinner": blocks.late_concat_rfa_thinner,
    "up_sampling": blocks.up_sampling,
    "up_sampling_rfa": blocks.up_sampling_rfa,
    "cb5_only": blocks.only_cb5,
    "late_concat_more_channels": blocks.late_concat_rfa_thinner_more_channels,
    "y_cb4_cbcr_cb5": blocks.y_in_CB4_cbcr_in_cb5,
}


def _check_weights_arg(weights, include_top, classes):
    if not (weights in {"imagenet", None} or os.path.exists(weights)):
        raise ValueError("The `weights` argument should be either `None` (random initialization), `imagenet` "
                         "(pre-training on ImageNet), or the path to the weights file to be loaded.")
    if weights == "imagenet" and include_top and classes != 1000:
        raise ValueError('If using `weights` as `"imagenet"` with `include_top` as true, `classes` should be 1000')
    if weights == "imagenet":
        raise RuntimeError("weights='imagenet' needs a download from github.com (keras_utils.get_file); no network is "
                           "available: pass weights=None or a local weight file")


def _top(x, include_top, pooling, classes):
    if include_top:
        x = GlobalAveragePooling2D(name="avg_pool")(x)
        x = Dense(classes, activation="softmax", name="fc1000")(x)
    elif pooling == "avg":
        x = GlobalAveragePooling2D()(x)
    elif pooling == "max":
        raise NotImplementedError("GlobalMaxPooling2D is not used by the reference's trainers")
    else:
        warnings.warn("The output shape of `ResNet50(include_top=False)` has been changed since Keras 2.2.0.")
    return x


def ResNet50RGB(include_top=True, weights="imagenet", input_tensor=None, input_shape=(224, 224, 3), pooling=None,
                classes=1000, **kwargs):
    """Stock ResNet50 on RGB pixels; `archi` and any other keyword is swallowed like the reference's **kwargs
    (its config/resnet passes archi= here -- SURVEY 3.2)."""
    _check_weights_arg(weights, include_top, classes)
    img_input = Input(shape=input_shape)
    x = ZeroPadding2D(padding=(3, 3), name="conv1_pad")(img_input)
    x = Conv2D(64, (7, 7), strides=(2, 2), padding="valid", kernel_initializer="he_normal", name="conv1")(x)
    x = BatchNormalization(axis=3, name="bn_conv1")(x)
    x = Activation("relu")(x)
    x = ZeroPadding2D(padding=(1, 1), name="pool1_pad")(x)
    x = MaxPooling2D((3, 3), strides=(2, 2))(x)
    x = conv_block(x, 3, [64, 64, 256], stage=2, block="a", strides=(1, 1))
    x = identity_block(x, 3, [64, 64, 256], stage=2, block="b")
    x = identity_block(x, 3, [64, 64, 256], stage=2, block="c")
    x = conv_block(x, 3, [128, 128, 512], stage=3, block="a")
    for b in "bcd":
        x = identity_block(x, 3, [128, 128, 512], stage=3, block=b)
    x = conv_block(x, 3, [256, 256, 1024], stage=4, block="a")
    for b in "bcdef":
        x = identity_block(x, 3, [256, 256, 1024], stage=4, block=b)
    x = blocks.block5(x)
    x = _top(x, include_top, pooling, classes)
    model = Model(img_input, x, name="resnet50rgb")
    if weights is not None:
        model.load_weights(weights)
    return model


def ResNet50Custom(include_top=True, weights="imagenet", input_tensor=None, input_shape=None, pooling=None,
                   classes=1000, archi="late_concat", **kwargs):
    """DCT-domain ResNet50: `archi` selects the backbone, stage 5 + GAP + `fc1000` follow."""
    _check_weights_arg(weights, include_top, classes)
    if archi == "deconv":
        x, input_shape, input_y, input_cb, input_cr = blocks.deconv(28)
        inputs = [input_y, input_cb, input_cr]
    elif archi in _ARCHIS:
        x, input_shape, input_y, input_cbcr = _ARCHIS[archi](28)
        inputs = [input_y, input_cbcr]
    else:
        # the reference falls through with `x` undefined (UnboundLocalError); fail with a message instead
        raise ValueError("Unknown network architecture %r" % (archi,))
    x = blocks.block5(x)
    x = _top(x, include_top, pooling, classes)
    model = Model(inputs=inputs, outputs=x, name="resnet50_custom")
    if weights is not None:
        model.load_weights(weights, by_name=False)
    return model

"""SSD300 over ResNet50-DCT backbones: `ssd_resnet_EF_layers_custom` and
`ssd_resnet_EF_layers_identical`, drop-in for the builders of
localisation_part/models/keras_ssd300_dct_j2d_resnet.py:167-932 and :935-1588 -- same keyword
arguments, validation errors, layer names, predictor order (so `y_pred` rows line up with
SSDInputEncoder's template) and `--archi` dispatch; the graph is built from this package's
Keras-style layers and runs on the MI355X engine.

Inputs are the de-quantised JPEG DCT coefficients jpeg2dct emits for a 300x300 image:
Y (38, 38, 64) and CbCr (19, 19, 128), or Y, Cb (19, 19, 64), Cr (19, 19, 64) for `archi="deconv"`.
Output: (batch, #boxes, n_classes + 1 + 4 + 8) = [softmax class scores | 4 box offsets | 4 anchor
coordinates | 4 variances]; #boxes = 8732 (custom) or 6716 (identical)."""
import numpy as np

from ..keras.layers import (Activation, Concatenate, Conv2D, MaxPooling2D, Reshape, ZeroPadding2D)
from ..keras.models import Model
from ..keras.regularizers import l2
from ..keras_layers.keras_layer_AnchorBoxes import AnchorBoxes
from ..keras_layers.keras_layer_L2Normalization import L2Normalization
from . import resnet_dct_blocks as blocks
from .resnet_dct_blocks import conv_block, identity_block  # noqa: F401  (same public names as the reference module)

_SOURCE_NAMES = ["conv4_3_norm", "fc7", "conv6_2", "conv7_2", "conv8_2", "conv9_2"]


def _check_ssd_arguments(n_predictor_layers, min_scale, max_scale, scales, aspect_ratios_global,
                         aspect_ratios_per_layer, two_boxes_for_ar1, steps, offsets, variances):
    """Argument validation of the reference builders (keras_ssd300_dct_j2d_resnet.py:324-401);
    returns (scales, aspect_ratios, n_boxes, steps, offsets, variances)."""
    if aspect_ratios_global is None and aspect_ratios_per_layer is None:
        raise ValueError("`aspect_ratios_global` and `aspect_ratios_per_layer` cannot both be None. At least one "
                         "needs to be specified.")
    if aspect_ratios_per_layer:
        if len(aspect_ratios_per_layer) != n_predictor_layers:
            raise ValueError("It must be either aspect_ratios_per_layer is None or len(aspect_ratios_per_layer) == {}, "
                             "but len(aspect_ratios_per_layer) == {}.".format(n_predictor_layers,
                                                                            len(aspect_ratios_per_layer)))
    if (min_scale is None or max_scale is None) and scales is None:
        raise ValueError("Either `min_scale` and `max_scale` or `scales` need to be specified.")
    if scales:
        if len(scales) != n_predictor_layers + 1:
            raise ValueError("It must be either scales is None or len(scales) == {}, but len(scales) == {}."
                             .format(n_predictor_layers + 1, len(scales)))
    else:
        scales = np.linspace(min_scale, max_scale, n_predictor_layers + 1)
    if len(variances) != 4:
        raise ValueError("4 variance values must be pased, but {} values were received.".format(len(variances)))
    variances = np.array(variances)
    if np.any(variances <= 0):
        raise ValueError("All variances must be >0, but the variances given are {}".format(variances))
    if (steps is not None) and (len(steps) != n_predictor_layers):
        raise ValueError("You must provide at least one step value per predictor layer.")
    if (offsets is not None) and (len(offsets) != n_predictor_layers):
        raise ValueError("You must provide at least one offset value per predictor layer.")
    if aspect_ratios_per_layer:
        aspect_ratios = aspect_ratios_per_layer
        n_boxes = [len(ar) + 1 if (1 in ar) and two_boxes_for_ar1 else len(ar) for ar in aspect_ratios_per_layer]
    else:
        aspect_ratios = [aspect_ratios_global] * n_predictor_layers
        nb = len(aspect_ratios_global) + 1 if (1 in aspect_ratios_global) and two_boxes_for_ar1 \
            else len(aspect_ratios_global)
        n_boxes = [nb] * n_predictor_layers
    if steps is None:
        steps = [None] * n_predictor_layers
    if offsets is None:
        offsets = [None] * n_predictor_layers
    return scales, aspect_ratios, n_boxes, steps, offsets, variances


def _head_conv(x, filters, kernel, name, l2_reg, strides=(1, 1), padding="same", dilation=(1, 1)):
    """SSD extra-feature conv: fused ReLU, he_normal, l2 kernel regulariser, no BatchNormalization."""
    return Conv2D(filters, kernel, strides=strides, dilation_rate=dilation, activation="relu", padding=padding,
                  kernel_initializer="he_normal", kernel_regularizer=l2(l2_reg), name=name)(x)


def _multibox(sources, model_inputs, n_classes, n_boxes, l2_reg, img_height, img_width, scales, aspect_ratios,
              two_boxes_for_ar1, steps, offsets, clip_boxes, variances, coords, normalize_coords, mode,
              return_predictor_sizes, decode_args):
    """Predictor convs, anchors, reshape/concat/softmax assembly (keras_ssd300_dct_j2d_resnet.py:562-932)."""
    def predictor(x, ch, name):
        return Conv2D(ch, (3, 3), padding="same", kernel_initializer="he_normal", kernel_regularizer=l2(l2_reg),
                      name=name)(x)

    conf = [predictor(s, n_boxes[i] * n_classes, "{}_mbox_conf_{}".format(_SOURCE_NAMES[i], n_classes))
            for i, s in enumerate(sources)]
    loc = [predictor(s, n_boxes[i] * 4, "{}_mbox_loc".format(_SOURCE_NAMES[i])) for i, s in enumerate(sources)]
    priors = [AnchorBoxes(img_height, img_width, this_scale=scales[i], next_scale=scales[i + 1],
                          aspect_ratios=aspect_ratios[i], two_boxes_for_ar1=two_boxes_for_ar1, this_steps=steps[i],
                          this_offsets=offsets[i], clip_boxes=clip_boxes, variances=variances, coords=coords,
                          normalize_coords=normalize_coords,
                          name="{}_mbox_priorbox".format(_SOURCE_NAMES[i]))(loc[i]) for i in range(len(sources))]
    conf_r = [Reshape((-1, n_classes), name="{}_mbox_conf_reshape".format(_SOURCE_NAMES[i]))(t)
              for i, t in enumerate(conf)]
    loc_r = [Reshape((-1, 4), name="{}_mbox_loc_reshape".format(_SOURCE_NAMES[i]))(t) for i, t in enumerate(loc)]
    pri_r = [Reshape((-1, 8), name="{}_mbox_priorbox_reshape".format(_SOURCE_NAMES[i]))(t)
             for i, t in enumerate(priors)]
    mbox_conf = Concatenate(axis=1, name="mbox_conf")(conf_r)
    mbox_loc = Concatenate(axis=1, name="mbox_loc")(loc_r)
    mbox_priorbox = Concatenate(axis=1, name="mbox_priorbox")(pri_r)
    mbox_conf_softmax = Activation("softmax", name="mbox_conf_softmax")(mbox_conf)
    predictions = Concatenate(axis=2, name="predictions_ssd")([mbox_conf_softmax, mbox_loc, mbox_priorbox])

    if mode == "training":
        model = Model(inputs=model_inputs, outputs=predictions)
    elif mode in ("inference", "inference_fast"):
        from ..keras_layers.keras_layer_DecodeDetections import DecodeDetections
        decoded = DecodeDetections(name="decoded_predictions", fast=(mode == "inference_fast"), **decode_args)(predictions)
        model = Model(inputs=model_inputs, outputs=decoded)
    else:
        raise ValueError("`mode` must be one of 'training', 'inference' or 'inference_fast', but received '{}'."
                         .format(mode))
    if return_predictor_sizes:
        return model, np.array([t._keras_shape[1:3] for t in conf])
    return model


def ssd_resnet_EF_layers_custom(image_size, n_classes, mode="training", l2_regularization=0.0005, min_scale=None,
                                max_scale=None, scales=None, aspect_ratios_global=None,
                                aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                                         [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                                         [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5],
                                                         [1.0, 2.0, 0.5]],
                                two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=None,
                                clip_boxes=False, variances=[0.1, 0.1, 0.2, 0.2], coords="centroids",
                                normalize_coords=True, subtract_mean=[123, 117, 104], divide_by_stddev=None,
                                swap_channels=[2, 1, 0], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                                nms_max_output_size=400, return_predictor_sizes=False, archi="ssd_custom"):
    """`--archi ssd_custom`: late-concat-RFA-thinner backbone; predictors on conv4_3 (38x38x384),
    conv3_3 (19x19x512), conv4_6 (10x10x1024) -- each L2-normalised -- and fc7 (5x5), conv6_2 (3x3),
    conv9_2 (1x1): 8732 boxes.  `subtract_mean`, `divide_by_stddev` and `swap_channels` are accepted
    and unused, as in the reference (the DCT inputs are never colour-normalised)."""
    n_predictor_layers = 6
    n_classes += 1
    l2_reg = l2_regularization
    img_height, img_width = image_size[0], image_size[1]
    scales, aspect_ratios, n_boxes, steps, offsets, variances = _check_ssd_arguments(
        n_predictor_layers, min_scale, max_scale, scales, aspect_ratios_global, aspect_ratios_per_layer,
        two_boxes_for_ar1, steps, offsets, variances)

    taps = {}
    x, _, input_y, input_cbcr = blocks.late_concat_rfa_thinner(38, taps)
    x = blocks.block5(x)
    pool5 = MaxPooling2D((3, 3), strides=(1, 1), padding="same", name="pool5_ssd")(x)
    fc6 = _head_conv(pool5, 1024, (3, 3), "fc6", l2_reg, dilation=(6, 6))
    fc7 = _head_conv(fc6, 1024, (1, 1), "fc7", l2_reg)
    conv6_1 = _head_conv(fc7, 256, (1, 1), "conv6_1", l2_reg)
    conv6_1 = ZeroPadding2D(padding=((1, 1), (1, 1)), name="conv6_padding")(conv6_1)
    conv6_2 = _head_conv(conv6_1, 256, (3, 3), "conv6_2", l2_reg, strides=(2, 2), padding="valid")
    conv9_1 = _head_conv(conv6_2, 128, (1, 1), "conv9_1", l2_reg)
    conv9_2 = _head_conv(conv9_1, 256, (3, 3), "conv9_2", l2_reg, padding="valid")
    conv4_3_norm = L2Normalization(gamma_init=20, name="conv4_3_norm")(taps["conv4_3"])
    conv3_3_norm = L2Normalization(gamma_init=20, name="conv3_3_norm")(taps["conv3_3"])
    conv4_6_norm = L2Normalization(gamma_init=20, name="conv4_6_norm")(taps["conv4_6"])
    sources = [conv4_3_norm, conv3_3_norm, conv4_6_norm, fc7, conv6_2, conv9_2]
    decode_args = dict(confidence_thresh=confidence_thresh, iou_threshold=iou_threshold, top_k=top_k,
                       nms_max_output_size=nms_max_output_size, coords=coords, normalize_coords=normalize_coords,
                       img_height=img_height, img_width=img_width)
    return _multibox(sources, [input_y, input_cbcr], n_classes, n_boxes, l2_reg, img_height, img_width, scales,
                     aspect_ratios, two_boxes_for_ar1, steps, offsets, clip_boxes, variances, coords,
                     normalize_coords, mode, return_predictor_sizes, decode_args)


def ssd_resnet_EF_layers_identical(image_size, n_classes, mode="training", l2_regularization=0.0005, min_scale=None,
                                   max_scale=None, scales=None, aspect_ratios_global=None,
                                   aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                                            [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                                                            [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5],
                                                            [1.0, 2.0, 0.5]],
                                   two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=None,
                                   clip_boxes=False, variances=[0.1, 0.1, 0.2, 0.2], coords="centroids",
                                   normalize_coords=True, subtract_mean=[123, 117, 104], divide_by_stddev=None,
                                   swap_channels=[2, 1, 0], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                                   nms_max_output_size=400, return_predictor_sizes=False, archi="deconv"):
    """`--archi deconv | y_cb4_cbcr_cb5 | up_sampling | cb5_only`: every backbone ends in a
    10x10x2048 stage-5 map; the stock SSD extra layers follow (conv7_2 at stride 1), and predictor 0
    reads the L2-normalised raw Y input (keras_ssd300_dct_j2d_resnet.py:1221): 6716 boxes."""
    n_predictor_layers = 6
    n_classes += 1
    l2_reg = l2_regularization
    img_height, img_width = image_size[0], image_size[1]
    scales, aspect_ratios, n_boxes, steps, offsets, variances = _check_ssd_arguments(
        n_predictor_layers, min_scale, max_scale, scales, aspect_ratios_global, aspect_ratios_per_layer,
        two_boxes_for_ar1, steps, offsets, variances)

    if archi == "deconv":
        x, _, input_y, input_cb, input_cr = blocks.deconv(38)
        model_inputs = [input_y, input_cb, input_cr]
    else:
        if archi == "y_cb4_cbcr_cb5":
            x, _, input_y, input_cbcr = blocks.y_in_CB4_cbcr_in_cb5(38)
        elif archi == "up_sampling":  # the reference dispatches this name to up_sampling_rfa()
            x, _, input_y, input_cbcr = blocks.up_sampling_rfa(38)
        elif archi == "cb5_only":
            x, _, input_y, input_cbcr = blocks.only_cb5(38)
        else:
            raise ValueError("Unknown network architecture")
        model_inputs = [input_y, input_cbcr]
    x = blocks.block5(x)

    pool5 = MaxPooling2D((3, 3), strides=(1, 1), padding="same", name="pool5_ssd")(x)
    fc6 = _head_conv(pool5, 1024, (3, 3), "fc6", l2_reg, dilation=(6, 6))
    fc7 = _head_conv(fc6, 1024, (1, 1), "fc7", l2_reg)
    conv6_1 = _head_conv(fc7, 256, (1, 1), "conv6_1", l2_reg)
    conv6_1 = ZeroPadding2D(padding=((1, 1), (1, 1)), name="conv6_padding")(conv6_1)
    conv6_2 = _head_conv(conv6_1, 512, (3, 3), "conv6_2", l2_reg, strides=(2, 2), padding="valid")
    conv7_1 = _head_conv(conv6_2, 128, (1, 1), "conv7_1", l2_reg)
    conv7_1 = ZeroPadding2D(padding=((1, 1), (1, 1)), name="conv7_padding")(conv7_1)
    conv7_2 = _head_conv(conv7_1, 256, (3, 3), "conv7_2", l2_reg, padding="valid")
    conv8_1 = _head_conv(conv7_2, 128, (1, 1), "conv8_1", l2_reg)
    conv8_2 = _head_conv(conv8_1, 256, (3, 3), "conv8_2", l2_reg, padding="valid")
    conv9_1 = _head_conv(conv8_2, 128, (1, 1), "conv9_1", l2_reg)
    conv9_2 = _head_conv(conv9_1, 256, (3, 3), "conv9_2", l2_reg, padding="valid")
    conv4_3_norm = L2Normalization(gamma_init=20, name="conv4_3_norm")(input_y)
    sources = [conv4_3_norm, fc7, conv6_2, conv7_2, conv8_2, conv9_2]
    decode_args = dict(confidence_thresh=confidence_thresh, iou_threshold=iou_threshold, top_k=top_k,
                       nms_max_output_size=nms_max_output_size, coords=coords, normalize_coords=normalize_coords,
                       img_height=img_height, img_width=img_width)
    return _multibox(sources, model_inputs, n_classes, n_boxes, l2_reg, img_height, img_width, scales, aspect_ratios,
                     two_boxes_for_ar1, steps, offsets, clip_boxes, variances, coords, normalize_coords, mode,
                     return_predictor_sizes, decode_args)

"""ORACLE (test infrastructure only).  Functional CPU restatement of the reference's model graphs
-- ResNet50-DCT backbones, SSD300 head, prediction assembly, training step -- on top of
oracle/keras_ops.py.  Weights come in as a dict keyed '<keras layer name>/<weight name>'.

Follows localisation_part/models/keras_ssd300_dct_j2d_resnet.py: identity_block :46-100, conv_block
:103-164, ssd_resnet_EF_layers_custom :440-879, ssd_resnet_EF_layers_identical :1096-1528, backbones
:1591-1771; classification_part/vgg_jpeg_keras/networks/resnet_dct.py:317-452 for the classifiers.
PARITY UNPINNED for the conv/BN/loss arithmetic (see keras_ops.py header); the anchor tensor is pinned
by tests/golden/anchors_*.npz generated from the reference's own numpy encoder."""
import math

import numpy as np
import torch

from . import keras_ops as ko

TRAIN_SSD_ARGS = dict(  # localisation_part/training_dct_pascal_j2d_resnet.py:92-125
    img_height=300, img_width=300, n_classes=20,
    scales=[0.1, 0.2, 0.37, 0.54, 0.71, 0.88, 1.05],
    aspect_ratios=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                   [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]],
    two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=[0.5] * 6, clip_boxes=False,
    variances=[0.1, 0.1, 0.2, 0.2], normalize_coords=True, l2_reg=0.0005)


class Net(object):
    """Carries the weight dict, the training flag and the BN moving-statistic updates of one pass."""

    def __init__(self, weights, training=True):
        self.w = weights
        self.training = training
        self.new_state = {}
        self.bn_counter = 0
        self.deconv_counter = 0
        self.reg_kernels = []
        self.activations = {}
        self.trace = None         # set to {} to record every conv output (and retain its gradient)

    def conv(self, x, name, strides=(1, 1), padding="valid", dilation=(1, 1), relu=False, reg=False):
        k = self.w[name + "/kernel"]
        if reg:
            self.reg_kernels.append(name + "/kernel")
        y = ko.conv2d(x, k, self.w[name + "/bias"], strides, padding, dilation)
        if self.trace is not None:
            if y.requires_grad:
                y.retain_grad()
            self.trace[name] = y
        return ko.relu(y) if relu else y

    def bn(self, x, name=None):
        if name is None:
            self.bn_counter += 1
            name = "batch_normalization_%d" % self.bn_counter
        g, b = self.w[name + "/gamma"], self.w[name + "/beta"]
        mm, mv = self.w[name + "/moving_mean"], self.w[name + "/moving_variance"]
        if not self.training:
            return ko.batch_norm_infer(x, g, b, mm, mv)
        y, mean, var = ko.batch_norm_train(x, g, b)
        count = x.shape[0] * x.shape[1] * x.shape[2]
        nm, nv = ko.batch_norm_moving_update(mm.detach(), mv.detach(), mean.detach(), var.detach(), count)
        self.new_state[name + "/moving_mean"] = nm
        self.new_state[name + "/moving_variance"] = nv
        return y

    def identity_block(self, x, k, stage, block):
        base = "res%d%s_branch" % (stage, block)
        bnb = "bn%d%s_branch" % (stage, block)
        y = ko.relu(self.bn(self.conv(x, base + "2a"), bnb + "2a"))
        y = ko.relu(self.bn(self.conv(y, base + "2b", padding="same"), bnb + "2b"))
        y = self.bn(self.conv(y, base + "2c"), bnb + "2c")
        return ko.relu(y + x)

    def conv_block(self, x, k, stage, block, strides=(2, 2)):
        base = "res%d%s_branch" % (stage, block)
        bnb = "bn%d%s_branch" % (stage, block)
        y = ko.relu(self.bn(self.conv(x, base + "2a", strides=strides), bnb + "2a"))
        y = ko.relu(self.bn(self.conv(y, base + "2b", padding="same"), bnb + "2b"))
        y = self.bn(self.conv(y, base + "2c"), bnb + "2c")
        sc = self.bn(self.conv(x, base + "1", strides=strides), bnb + "1")
        return ko.relu(y + sc)

    def deconv2x(self, x):
        self.deconv_counter += 1
        name = "conv2d_transpose_%d" % self.deconv_counter
        return ko.conv2d_transpose(x, self.w[name + "/kernel"], self.w[name + "/bias"], (2, 2))


def _block5(net, x):
    x = net.conv_block(x, 3, 5, "a")
    x = net.identity_block(x, 3, 5, "b")
    return net.identity_block(x, 3, 5, "c")


def _rfa_trunk(net, x):
    x = net.conv_block(x, 1, 4, "a2", (1, 1))
    x = net.identity_block(x, 2, 4, "b2")
    x = net.identity_block(x, 3, 4, "c2")
    x = net.conv_block(x, 3, 3, "a1", (1, 1))
    for b in "bcd":
        x = net.identity_block(x, 3, 3, b)
    x = net.conv_block(x, 3, 4, "a")
    for b in "bcdef":
        x = net.identity_block(x, 3, 4, b)
    return x


def _y_prefix(net, y):
    y = net.conv_block(y, 1, 1, "a2", (1, 1))
    y = net.identity_block(y, 2, 1, "b2")
    y = net.identity_block(y, 3, 1, "c2")
    y = net.conv_block(y, 3, 2, "a3", (1, 1))
    for b in ("b3", "c3", "d3"):
        y = net.identity_block(y, 3, 2, b)
    return y


def backbone(net, archi, inputs, taps):
    """Returns the tensor that feeds stage 5 (block5)."""
    if archi in ("ssd_custom", "late_concat_rfa_thinner"):
        y_in, cbcr_in = inputs
        y = _y_prefix(net, net.bn(y_in))
        taps["conv4_3"] = y
        y = net.conv_block(y, 3, 2, "a4")
        cbcr = net.conv_block(net.bn(cbcr_in), 1, 2, "a5", (1, 1))
        x = torch.cat([y, cbcr], dim=-1)
        for b in "bcd":
            x = net.identity_block(x, 3, 3, b)
        taps["conv3_3"] = x
        x = net.conv_block(x, 3, 4, "a")
        for b in "bcdef":
            x = net.identity_block(x, 3, 4, b)
        taps["conv4_6"] = x
        return x
    if archi == "deconv":
        y_in, cb_in, cr_in = inputs
        cb, cr = net.deconv2x(cb_in), net.deconv2x(cr_in)
        x = torch.cat([y_in, torch.cat([cb, cr], dim=-1)], dim=-1)
        return _rfa_trunk(net, net.bn(x))
    if archi in ("up_sampling", "up_sampling_rfa"):
        y_in, cbcr_in = inputs
        x = torch.cat([y_in, ko.upsampling_nearest_2x(cbcr_in)], dim=-1)
        return _rfa_trunk(net, net.bn(x))
    if archi == "late_concat_more_channels":
        # C/vgg_jpeg_keras/networks/resnet_dct.py:529-566: the late-concat graph with 768-channel Y stages (the filter
        # counts come with the weights) and stage-3 identity blocks named b1 / c1 / d1
        y_in, cbcr_in = inputs
        y = _y_prefix(net, net.bn(y_in))
        y = net.conv_block(y, 3, 2, "a4")
        cbcr = net.conv_block(net.bn(cbcr_in), 1, 2, "a5", (1, 1))
        x = torch.cat([y, cbcr], dim=-1)
        for b in ("b1", "c1", "d1"):
            x = net.identity_block(x, 3, 3, b)
        x = net.conv_block(x, 3, 4, "a")
        for b in "bcdef":
            x = net.identity_block(x, 3, 4, b)
        return x
    if archi == "up_sampling_plain":
        # C/vgg_jpeg_keras/networks/resnet_dct.py:454-487 (classifier `archi="up_sampling"`): no RFA prefix
        y_in, cbcr_in = inputs
        x = net.bn(torch.cat([y_in, ko.upsampling_nearest_2x(cbcr_in)], dim=-1))
        x = net.conv_block(x, 3, 3, "a1", (1, 1))
        for b in "bcd":
            x = net.identity_block(x, 3, 3, b)
        x = net.conv_block(x, 3, 4, "a")
        for b in "bcdef":
            x = net.identity_block(x, 3, 4, b)
        return x
    if archi == "cb5_only":
        y_in, cbcr_in = inputs
        y = _y_prefix(net, net.bn(y_in))
        y = net.conv_block(y, 3, 2, "a4")
        cbcr = net.conv_block(net.bn(cbcr_in), 1, 2, "a5", (1, 1))
        return torch.cat([y, cbcr], dim=-1)
    if archi == "y_cb4_cbcr_cb5":
        y_in, cbcr_in = inputs
        y = _y_prefix(net, net.bn(y_in))
        x = net.conv_block(y, 3, 4, "a2")
        for b in ("b2", "c2", "d2", "e2", "f2"):
            x = net.identity_block(x, 3, 4, b)
        cbcr = net.conv_block(net.bn(cbcr_in), 1, 2, "a5", (1, 1))
        return torch.cat([x, cbcr], dim=-1)
    raise ValueError("Unknown network architecture")


# ------------------------------------------------------------------------------------
# anchors (L/keras_layers/keras_layer_AnchorBoxes.py:150-248), centroids + normalised
# ------------------------------------------------------------------------------------
def anchor_boxes(fm_h, fm_w, this_scale, next_scale, aspect_ratios, step, offset, cfg=TRAIN_SSD_ARGS):
    size = min(cfg["img_height"], cfg["img_width"])
    wh = []
    for ar in aspect_ratios:
        if ar == 1:
            wh.append((this_scale * size, this_scale * size))
            if cfg["two_boxes_for_ar1"]:
                wh.append((math.sqrt(this_scale * next_scale) * size,) * 2)
        else:
            wh.append((this_scale * size * math.sqrt(ar), this_scale * size / math.sqrt(ar)))
    wh = np.array(wh)
    cy = np.linspace(offset * step, (offset + fm_h - 1) * step, fm_h)
    cx = np.linspace(offset * step, (offset + fm_w - 1) * step, fm_w)
    out = np.zeros((fm_h, fm_w, len(wh), 8))
    xmin = cx[None, :, None] - wh[None, None, :, 0] / 2.0
    xmax = cx[None, :, None] + wh[None, None, :, 0] / 2.0
    ymin = cy[:, None, None] - wh[None, None, :, 1] / 2.0
    ymax = cy[:, None, None] + wh[None, None, :, 1] / 2.0
    if cfg["clip_boxes"]:
        raise NotImplementedError
    xmin, xmax = xmin / cfg["img_width"], xmax / cfg["img_width"]
    ymin, ymax = ymin / cfg["img_height"], ymax / cfg["img_height"]
    out[..., 0] = (xmin + xmax) / 2.0
    out[..., 1] = (ymin + ymax) / 2.0
    out[..., 2] = xmax - xmin
    out[..., 3] = ymax - ymin
    out[..., 4:] = np.asarray(cfg["variances"])
    return out


_SRC = ["conv4_3_norm", "fc7", "conv6_2", "conv7_2", "conv8_2", "conv9_2"]


def ssd_forward(weights, inputs, archi, training=True, cfg=TRAIN_SSD_ARGS):
    """-> (y_pred (B, #boxes, 33), net).  `inputs`: [Y, CbCr] or [Y, Cb, Cr] NHWC tensors."""
    net = Net(weights, training)
    taps = {}
    x = _block5(net, backbone(net, archi, inputs, taps))
    pool5 = ko.max_pool_3x3_s1_same(x)
    fc6 = net.conv(pool5, "fc6", padding="same", dilation=(6, 6), relu=True, reg=True)
    fc7 = net.conv(fc6, "fc7", padding="same", relu=True, reg=True)
    c61 = net.conv(fc7, "conv6_1", padding="same", relu=True, reg=True)
    c61 = ko.zero_padding(c61, ((1, 1), (1, 1)))
    if archi == "ssd_custom":
        c62 = net.conv(c61, "conv6_2", strides=(2, 2), relu=True, reg=True)
        c91 = net.conv(c62, "conv9_1", padding="same", relu=True, reg=True)
        c92 = net.conv(c91, "conv9_2", relu=True, reg=True)
        sources = [ko.l2_normalization(taps["conv4_3"], weights["conv4_3_norm/conv4_3_norm_gamma"]),
                   ko.l2_normalization(taps["conv3_3"], weights["conv3_3_norm/conv3_3_norm_gamma"]),
                   ko.l2_normalization(taps["conv4_6"], weights["conv4_6_norm/conv4_6_norm_gamma"]),
                   fc7, c62, c92]
    else:
        c62 = net.conv(c61, "conv6_2", strides=(2, 2), relu=True, reg=True)
        c71 = net.conv(c62, "conv7_1", padding="same", relu=True, reg=True)
        c71 = ko.zero_padding(c71, ((1, 1), (1, 1)))
        c72 = net.conv(c71, "conv7_2", relu=True, reg=True)
        c81 = net.conv(c72, "conv8_1", padding="same", relu=True, reg=True)
        c82 = net.conv(c81, "conv8_2", relu=True, reg=True)
        c91 = net.conv(c82, "conv9_1", padding="same", relu=True, reg=True)
        c92 = net.conv(c91, "conv9_2", relu=True, reg=True)
        sources = [ko.l2_normalization(inputs[0], weights["conv4_3_norm/conv4_3_norm_gamma"]),
                   fc7, c62, c72, c82, c92]
    n_cls = cfg["n_classes"] + 1
    b = inputs[0].shape[0]
    confs, locs, pris = [], [], []
    for i, s in enumerate(sources):
        c = net.conv(s, "%s_mbox_conf_%d" % (_SRC[i], n_cls), padding="same", reg=True)
        l = net.conv(s, "%s_mbox_loc" % _SRC[i], padding="same", reg=True)
        confs.append(c.reshape(b, -1, n_cls))
        locs.append(l.reshape(b, -1, 4))
        a = anchor_boxes(s.shape[1], s.shape[2], cfg["scales"][i], cfg["scales"][i + 1], cfg["aspect_ratios"][i],
                         cfg["steps"][i], cfg["offsets"][i], cfg)
        a = torch.from_numpy(a).to(c.dtype).reshape(1, -1, 8).expand(b, -1, -1)
        pris.append(a)
    conf = ko.softmax(torch.cat(confs, dim=1))
    y_pred = torch.cat([conf, torch.cat(locs, dim=1), torch.cat(pris, dim=1)], dim=2)
    net.activations.update(dict(fc7=fc7, conv6_2=c62, conv9_2=c92, block5=x, **taps))
    return y_pred, net


def classifier_forward(weights, inputs, archi, training=True):
    """ResNet50Custom(archi=...) (C/vgg_jpeg_keras/networks/resnet_dct.py:392-417): -> (probs, net)."""
    net = Net(weights, training)
    if archi == "up_sampling":     # the classifier dispatches this name to the plain graph, the SSD builder to the RFA one
        archi = "up_sampling_plain"
    x = _block5(net, backbone(net, archi, inputs, {}))
    x = ko.global_average_pooling(x)
    logits = ko.dense(x, weights["fc1000/kernel"], weights["fc1000/bias"])
    return ko.softmax(logits), net


def ssd_training_step(weights, inputs, y_true, archi, lr=0.001, momentum=0.9, decay=0.0, nesterov=False,
                      velocities=None, iterations=0, cfg=TRAIN_SSD_ARGS, trainable=None):
    """One `train_on_batch` of the reference's SSD trainer (TRAIN_SSD:152-156): loss = Keras mean of
    SSDLoss.compute_loss + sum of l2 penalties; SGD update; BN moving statistics update.
    Returns dict(loss, data_loss, y_pred, grads, new_weights, new_velocities)."""
    leaf = {}
    for k, v in weights.items():
        is_state = k.endswith("moving_mean") or k.endswith("moving_variance")
        leaf[k] = v.clone().requires_grad_(not is_state and (trainable is None or k in trainable))
    y_pred, net = ssd_forward(leaf, inputs, archi, training=True, cfg=cfg)
    data_loss = ko.ssd_loss(y_true, y_pred).mean()
    reg = sum(ko.l2_penalty(leaf[k], cfg["l2_reg"]) for k in net.reg_kernels)
    loss = data_loss + reg
    loss.backward()
    new_w, new_v, grads = {}, {}, {}
    for k, v in leaf.items():
        if v.requires_grad and v.grad is not None:
            grads[k] = v.grad.detach()
            vel = velocities[k] if velocities is not None else torch.zeros_like(v)
            p, nv = ko.sgd_keras_step(v.detach(), v.grad, vel, lr, momentum, decay, iterations, nesterov)
            new_w[k], new_v[k] = p, nv
        else:
            new_w[k] = v.detach()
    new_w.update(net.new_state)
    return dict(loss=float(loss.detach()), data_loss=float(data_loss.detach()), reg_loss=float(reg.detach()), y_pred=y_pred.detach(),
                grads=grads, new_weights=new_w, new_velocities=new_v, net=net)


def resnet_rgb_forward(weights, image, training=True):
    """ResNet50RGB (C/vgg_jpeg_keras/networks/resnet_dct.py:165-314): -> (probs, net).  `image` (B,224,224,3)."""
    net = Net(weights, training)
    x = ko.zero_padding(image, ((3, 3), (3, 3)))
    x = ko.relu(net.bn(net.conv(x, "conv1", strides=(2, 2)), "bn_conv1"))
    x = ko.max_pool(ko.zero_padding(x, ((1, 1), (1, 1))), (3, 3), (2, 2), "valid")
    x = net.conv_block(x, 3, 2, "a", (1, 1))
    for b in "bc":
        x = net.identity_block(x, 3, 2, b)
    x = net.conv_block(x, 3, 3, "a")
    for b in "bcd":
        x = net.identity_block(x, 3, 3, b)
    x = net.conv_block(x, 3, 4, "a")
    for b in "bcdef":
        x = net.identity_block(x, 3, 4, b)
    x = _block5(net, x)
    x = ko.global_average_pooling(x)
    return ko.softmax(ko.dense(x, weights["fc1000/kernel"], weights["fc1000/bias"])), net


def classifier_training_step(weights, inputs, y_onehot, archi, lr=0.1, momentum=0.9, decay=1e-4, nesterov=True,
                             velocities=None, iterations=0):
    """One train_on_batch of the classification trainer (C/training.py:175-198, C/config/resnet/config_file.py:58-65):
    categorical cross-entropy (batch mean) + Keras SGD.  archi 'resnet_rgb' or a DCT archi name."""
    leaf = {}
    for k, v in weights.items():
        is_state = k.endswith("moving_mean") or k.endswith("moving_variance")
        leaf[k] = v.clone().requires_grad_(not is_state)
    if archi == "resnet_rgb":
        probs, net = resnet_rgb_forward(leaf, inputs[0], training=True)
    else:
        probs, net = classifier_forward(leaf, inputs, archi, training=True)
    loss = ko.categorical_crossentropy(y_onehot, probs).mean()
    loss.backward()
    new_w, new_v, grads = {}, {}, {}
    for k, v in leaf.items():
        if v.requires_grad and v.grad is not None:
            grads[k] = v.grad.detach()
            vel = velocities[k] if velocities is not None else torch.zeros_like(v)
            p, nv = ko.sgd_keras_step(v.detach(), v.grad, vel, lr, momentum, decay, iterations, nesterov)
            new_w[k], new_v[k] = p, nv
        else:
            new_w[k] = v.detach()
    new_w.update(net.new_state)
    return dict(loss=float(loss.detach()), probs=probs.detach(), grads=grads, new_weights=new_w, new_velocities=new_v)

"""ResNet50 bottleneck blocks and the DCT-domain backbones ("archis") shared by the SSD300 and the
classifier builders.  Restates, table-driven, identity_block / conv_block and the backbone functions of
localisation_part/models/keras_ssd300_dct_j2d_resnet.py:46-164,440-479,1591-1771 and
classification_part/vgg_jpeg_keras/networks/resnet_dct.py:59-163,454-711: same layer names
(`res{stage}{block}_branch2a`, `bn...`), same creation order (so Keras auto-names such as
batch_normalization_1 agree), same filters / kernel sizes / strides.

`grid` is the Y block grid: 38 for 300x300 SSD inputs, 28 for 224x224 classifier inputs; chroma is
grid/2 (4:2:0)."""
from ..keras.layers import (Activation, Add, BatchNormalization, Concatenate, Conv2D, Conv2DTranspose, Input,
                            UpSampling2D)


def identity_block(input_tensor, kernel_size, filters, stage, block):
    """Bottleneck without a conv on the shortcut: 1x1 -> kxk 'same' -> 1x1, each followed by BN, ReLU
    after the first two and after the residual Add."""
    f1, f2, f3 = filters
    conv_name = "res" + str(stage) + block + "_branch"
    bn_name = "bn" + str(stage) + block + "_branch"
    x = input_tensor
    for suffix, f, k, pad in (("2a", f1, (1, 1), "valid"), ("2b", f2, kernel_size, "same"), ("2c", f3, (1, 1), "valid")):
        x = Conv2D(f, k, padding=pad, kernel_initializer="he_normal", name=conv_name + suffix)(x)
        x = BatchNormalization(axis=3, name=bn_name + suffix)(x)
        if suffix != "2c":
            x = Activation("relu")(x)
    x = Add()([x, input_tensor])
    return Activation("relu")(x)


def conv_block(input_tensor, kernel_size, filters, stage, block, strides=(2, 2)):
    """Bottleneck whose shortcut is a strided 1x1 conv + BN (`..._branch1`)."""
    f1, f2, f3 = filters
    conv_name = "res" + str(stage) + block + "_branch"
    bn_name = "bn" + str(stage) + block + "_branch"
    x = Conv2D(f1, (1, 1), strides=strides, kernel_initializer="he_normal", name=conv_name + "2a")(input_tensor)
    x = BatchNormalization(axis=3, name=bn_name + "2a")(x)
    x = Activation("relu")(x)
    x = Conv2D(f2, kernel_size, padding="same", kernel_initializer="he_normal", name=conv_name + "2b")(x)
    x = BatchNormalization(axis=3, name=bn_name + "2b")(x)
    x = Activation("relu")(x)
    x = Conv2D(f3, (1, 1), kernel_initializer="he_normal", name=conv_name + "2c")(x)
    x = BatchNormalization(axis=3, name=bn_name + "2c")(x)
    shortcut = Conv2D(f3, (1, 1), strides=strides, kernel_initializer="he_normal", name=conv_name + "1")(input_tensor)
    shortcut = BatchNormalization(axis=3, name=bn_name + "1")(shortcut)
    x = Add()([x, shortcut])
    return Activation("relu")(x)


def _stack(x, rows):
    """rows: ('c', kernel, filters, stage, block[, stride]) conv_block | ('i', kernel, filters, stage, block)."""
    for row in rows:
        if row[0] == "c":
            s = row[5] if len(row) > 5 else 2
            x = conv_block(x, row[1], list(row[2]), stage=row[3], block=row[4], strides=(s, s))
        else:
            x = identity_block(x, row[1], list(row[2]), stage=row[3], block=row[4])
    return x


def _ids(kernel, filters, stage, blocks):
    return [("i", kernel, filters, stage, b) for b in blocks]


def block5(x):
    """conv5_x: the last ResNet50 stage (stride 2)."""
    f = (512, 512, 2048)
    return _stack(x, [("c", 3, f, 5, "a")] + _ids(3, f, 5, "bc"))


def _rfa_trunk(x, taps):
    """Receptive-field-aware trunk of the Uber deconv / up_sampling_rfa archis: full-resolution
    stage-4 prefix (a2,b2,c2), stride-1 stage 3, then stage 4 at half resolution."""
    f4, f3 = (256, 256, 1024), (128, 128, 512)
    x = _stack(x, [("c", 1, f4, 4, "a2", 1), ("i", 2, f4, 4, "b2"), ("i", 3, f4, 4, "c2")])
    x = _stack(x, [("c", 3, f3, 3, "a1", 1)] + _ids(3, f3, 3, "bcd"))
    x = _stack(x, [("c", 3, f4, 4, "a")] + _ids(3, f4, 4, "bc"))
    taps["conv4_3"] = x
    return _stack(x, _ids(3, f4, 4, "def"))


def _y_prefix(y, f1, f2):
    """Stage 1 (k=1 conv block, k=2 and k=3 identity blocks) and stride-1 stage 2 on the Y stream."""
    y = _stack(y, [("c", 1, f1, 1, "a2", 1), ("i", 2, f1, 1, "b2"), ("i", 3, f1, 1, "c2")])
    return _stack(y, [("c", 3, f2, 2, "a3", 1)] + _ids(3, f2, 2, ["b3", "c3", "d3"]))


def late_concat_rfa_thinner(grid, taps=None, block3_names="bcd"):
    """Two-stream late-concat archi (the backbone of `ssd_custom`): Y through stages 1-2 at full
    resolution then stride 2; CbCr through one conv block; concat; stages 3 and 4."""
    taps = {} if taps is None else taps
    input_y, input_cbcr = Input((grid, grid, 64)), Input((grid // 2, grid // 2, 128))
    y = BatchNormalization(input_shape=(grid, grid, 64))(input_y)
    y = _y_prefix(y, (256, 256, 384), (128, 128, 384))
    taps["conv4_3"] = y
    y = conv_block(y, 3, [256, 256, 384], stage=2, block="a4")
    cbcr = BatchNormalization(input_shape=(grid // 2, grid // 2, 128))(input_cbcr)
    cbcr = conv_block(cbcr, 1, [256, 256, 128], stage=2, block="a5", strides=(1, 1))
    x = Concatenate(axis=-1)([y, cbcr])
    x = _stack(x, _ids(3, (128, 128, 512), 3, block3_names))
    taps["conv3_3"] = x
    f4 = (256, 256, 1024)
    x = _stack(x, [("c", 3, f4, 4, "a")] + _ids(3, f4, 4, "bcdef"))
    taps["conv4_6"] = x
    return x, [grid, grid, 192], input_y, input_cbcr


def late_concat_rfa_thinner_more_channels(grid, taps=None):
    taps = {} if taps is None else taps
    input_y, input_cbcr = Input((grid, grid, 64)), Input((grid // 2, grid // 2, 128))
    y = BatchNormalization(input_shape=(grid, grid, 64))(input_y)
    y = _y_prefix(y, (256, 256, 768), (256, 256, 768))
    y = conv_block(y, 3, [256, 256, 384], stage=2, block="a4")
    cbcr = BatchNormalization(input_shape=(grid // 2, grid // 2, 128))(input_cbcr)
    cbcr = conv_block(cbcr, 1, [256, 256, 128], stage=2, block="a5", strides=(1, 1))
    x = Concatenate(axis=-1)([y, cbcr])
    x = _stack(x, _ids(3, (128, 128, 512), 3, ["b1", "c1", "d1"]))
    f4 = (256, 256, 1024)
    x = _stack(x, [("c", 3, f4, 4, "a")] + _ids(3, f4, 4, "bcdef"))
    return x, [grid, grid, 192], input_y, input_cbcr


def up_sampling(grid, taps=None):
    """Nearest-neighbour up-sampled chroma concatenated with Y, then stages 3 and 4 (no RFA prefix)."""
    input_y, input_cbcr = Input((grid, grid, 64)), Input((grid // 2, grid // 2, 128))
    cbcr = UpSampling2D()(input_cbcr)
    concat = Concatenate(axis=-1)([input_y, cbcr])
    x = BatchNormalization(input_shape=(grid, grid, 64))(concat)
    f3, f4 = (128, 128, 512), (256, 256, 1024)
    x = _stack(x, [("c", 3, f3, 3, "a1", 1)] + _ids(3, f3, 3, "bcd"))
    x = _stack(x, [("c", 3, f4, 4, "a")] + _ids(3, f4, 4, "bcdef"))
    return x, [grid, grid, 192], input_y, input_cbcr


def up_sampling_rfa(grid, taps=None):
    taps = {} if taps is None else taps
    input_y, input_cbcr = Input((grid, grid, 64)), Input((grid // 2, grid // 2, 128))
    cbcr = UpSampling2D()(input_cbcr)
    concat = Concatenate(axis=-1)([input_y, cbcr])
    x = BatchNormalization(input_shape=(grid, grid, 64))(concat)
    x = _rfa_trunk(x, taps)
    return x, [grid, grid, 192], input_y, input_cbcr


def deconv(grid, taps=None):
    """Learned 2x chroma up-sampling (Conv2DTranspose k2 s2 on Cb and Cr separately) + RFA trunk;
    three model inputs [Y, Cb, Cr]."""
    taps = {} if taps is None else taps
    half = grid // 2
    input_y, input_cb, input_cr = Input((grid, grid, 64)), Input((half, half, 64)), Input((half, half, 64))
    cb = Conv2DTranspose(64, 2, strides=(2, 2))(input_cb)
    cr = Conv2DTranspose(64, 2, strides=(2, 2))(input_cr)
    cbcr = Concatenate(axis=-1)([cb, cr])
    concat = Concatenate(axis=-1)([input_y, cbcr])
    x = BatchNormalization(input_shape=(grid, grid, 64))(concat)
    x = _rfa_trunk(x, taps)
    return x, [grid, grid, 192], input_y, input_cb, input_cr


def only_cb5(grid, taps=None):
    taps = {} if taps is None else taps
    input_y, input_cbcr = Input((grid, grid, 64)), Input((grid // 2, grid // 2, 128))
    y = BatchNormalization(input_shape=(grid, grid, 64))(input_y)
    y = _y_prefix(y, (256, 256, 768), (256, 256, 768))
    taps["conv4_3"] = y
    y = conv_block(y, 3, [256, 256, 768], stage=2, block="a4")
    cbcr = BatchNormalization(input_shape=(grid // 2, grid // 2, 128))(input_cbcr)
    cbcr = conv_block(cbcr, 1, [256, 256, 256], stage=2, block="a5", strides=(1, 1))
    x = Concatenate(axis=-1)([y, cbcr])
    return x, [grid, grid, 192], input_y, input_cbcr


def y_in_CB4_cbcr_in_cb5(grid, taps=None):
    """Y through stages 1, 2 and a 768-wide stage 4; CbCr joins just before stage 5.  The reference
    also builds a `res2a4` conv block whose output is never used (keras_ssd300_dct_j2d_resnet.py:1608);
    it is created here too (it consumes the same auto-name counters) and, as in Keras, does not become
    part of the Model because nothing downstream reads it."""
    taps = {} if taps is None else taps
    input_y, input_cbcr = Input((grid, grid, 64)), Input((grid // 2, grid // 2, 128))
    y = BatchNormalization(input_shape=(grid, grid, 64))(input_y)
    y = _y_prefix(y, (256, 256, 384), (128, 128, 512))
    taps["conv4_3"] = y
    conv_block(y, 3, [256, 256, 384], stage=2, block="a4", strides=(1, 1))  # dead branch, see docstring
    f = (256, 256, 768)
    x = _stack(y, [("c", 3, f, 4, "a2")] + _ids(3, f, 4, ["b2", "c2", "d2", "e2", "f2"]))
    cbcr = BatchNormalization(input_shape=(grid // 2, grid // 2, 128))(input_cbcr)
    cbcr = conv_block(cbcr, 1, [256, 256, 256], stage=2, block="a5", strides=(1, 1))
    x = Concatenate(axis=-1)([x, cbcr])
    return x, [grid, grid, 192], input_y, input_cbcr

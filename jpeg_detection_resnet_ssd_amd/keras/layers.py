"""Keras-2.2.4-style layers, limited to what the reference's ResNet50-DCT / SSD300 builders use
(localisation_part/models/keras_ssd300_dct_j2d_resnet.py:22-36,
classification_part/vgg_jpeg_keras/networks/resnet_dct.py), each with a `lower()` that turns it into
C-ABI launches inside an engine.Plan.  Same constructor keywords, defaults, auto-naming and
`_keras_shape` / `output_shape` attributes as Keras, so builder code reads like the reference's."""
import os
import re

import numpy as np
import torch

from .. import engine
from ..engine import GradRef, Value, call, launcher, query, rows_of
from .. import kernels as Kn
from . import backend as K
from . import initializers

_SERIAL = [0]


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()


def _pair(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 2
        return (int(v[0]), int(v[1]))
    return (int(v), int(v))


class KTensor(object):
    """Symbolic tensor: static shape with a None batch axis, and the layer that produces it."""

    def __init__(self, shape, layer=None, index=0):
        self.shape = tuple(shape)
        self._keras_shape = self.shape
        self.layer = layer
        self.index = index

    def __repr__(self):
        return "<KTensor %s from %s>" % (self.shape, self.layer.name if self.layer else None)


class InputSpec(object):
    def __init__(self, shape=None, ndim=None, dtype=None):
        self.shape = shape
        self.ndim = ndim
        self.dtype = dtype


class WeightSpec(object):
    def __init__(self, layer, name, shape, initializer, trainable, l2):
        self.layer = layer
        self.name = name            # Keras weight name inside the layer, e.g. 'kernel'
        self.shape = tuple(int(s) for s in shape)
        self.initializer = initializer
        self.trainable = trainable
        self.l2 = l2
        self.param = None           # device tensor (view into the model's flat store)
        self.grad = None
        self.init_value = None

    @property
    def key(self):
        return "%s/%s" % (self.layer.name, self.name)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


class Layer(object):
    """Base class following keras.engine.topology.Layer's protocol (build / call /
    compute_output_shape / get_config), cf. localisation_part/keras_layers/*.py."""

    def __init__(self, name=None, trainable=True, input_shape=None, **kwargs):
        if kwargs:
            unknown = set(kwargs) - {"dtype", "batch_input_shape", "weights"}
            if unknown:
                raise TypeError("unexpected keyword arguments %s" % sorted(unknown))
        if not name:
            prefix = _snake(self.__class__.__name__)
            name = "%s_%d" % (prefix, K.get_uid(prefix))
        self.name = name
        self.trainable = trainable
        self.built = False
        self.weight_specs = []
        self.inbound = None
        self.outbound = None
        self.input_spec = None
        self.serial = None

    # -- Keras protocol ------------------------------------------------------------
    def build(self, input_shape):
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable}

    def add_weight(self, name, shape, initializer="zeros", trainable=True, regularizer=None):
        spec = WeightSpec(self, name, shape, initializers.get(initializer), trainable and self.trainable,
                          regularizer.l2 if regularizer is not None else 0.0)
        self.weight_specs.append(spec)
        return spec

    @property
    def weights(self):
        return list(self.weight_specs)

    @property
    def trainable_weights(self):
        return [w for w in self.weight_specs if w.trainable]

    @trainable_weights.setter
    def trainable_weights(self, value):  # custom layers assign it (L2Normalization.build)
        pass

    def count_params(self):
        return sum(w.size for w in self.weight_specs)

    def __call__(self, inputs):
        if self.inbound is not None:
            raise NotImplementedError("layer %s called twice: shared layers are not on the reference's path" % self.name)
        many = isinstance(inputs, (list, tuple))
        ins = list(inputs) if many else [inputs]
        for t in ins:
            if not isinstance(t, KTensor):
                raise TypeError("layer %s called on a non-tensor %r" % (self.name, t))
        shape_arg = [t.shape for t in ins] if many else ins[0].shape
        if not self.built:
            self.build(shape_arg)
            self.built = True
        out_shape = self.compute_output_shape(shape_arg)
        self.inbound = ins
        _SERIAL[0] += 1
        self.serial = _SERIAL[0]
        out = KTensor(out_shape, self, 0)
        self.outbound = [out]
        self.input_shape = shape_arg
        self.output_shape = tuple(out_shape)
        return out

    @property
    def output(self):
        return self.outbound[0]

    @property
    def input(self):
        return self.inbound[0] if len(self.inbound) == 1 else self.inbound

    # -- lowering ---------------------------------------------------------------
    def lower(self, plan, model, ins):
        raise NotImplementedError("%s has no MI355X lowering" % self.__class__.__name__)


class InputLayer(Layer):
    def __init__(self, shape, name=None):
        super(InputLayer, self).__init__(name=name or "input_%d" % K.get_uid("input"))
        self.shape = (None,) + tuple(shape)
        self.inbound = []
        _SERIAL[0] += 1
        self.serial = _SERIAL[0]
        self.outbound = [KTensor(self.shape, self, 0)]
        self.output_shape = self.shape
        self.input_shape = self.shape
        self.built = True

    def lower(self, plan, model, ins):
        buf = plan.empty(plan.batch_size, *self.shape[1:])
        plan.inputs.append(buf)
        return Value(buf, needs_grad=False, name=self.name)


def Input(shape=None, batch_shape=None, name=None, dtype=None, tensor=None):
    if shape is None and batch_shape is not None:
        shape = batch_shape[1:]
    return InputLayer(tuple(shape), name=name).outbound[0]


# =====================================================================================
# helpers shared by the lowerings
# =====================================================================================
def _materialised(v, who, plan=None, allow_pad=False):
    """Buffer holding the value itself.  A virtual BatchNormalization(+ReLU) output is written out with one
    dj_affine_act pass (cached per Value); gradients keep being requested on `v`, whose BatchNormalization applies
    the ReLU mask itself."""
    if v.pad is not None and not allow_pad:
        raise NotImplementedError("%s cannot consume a pending ZeroPadding2D" % who)
    if not v.is_affine:
        return v.buf
    if plan is None:
        raise NotImplementedError("%s cannot consume a virtual BatchNormalization output" % who)
    src = v.alias_of if (v.alias_of is not None and v.pad is not None) else v   # ZeroPadding2D alias shares the data
    if getattr(src, "_mat", None) is None:
        rows, c, ld = rows_of(src.buf)
        y = plan.empty(*src.buf.shape)
        zbuf, sc, sh, relu = src.buf, src.scale, src.shift, int(src.relu)
        plan.emit(launcher(*Kn.affine_act_call(zbuf, ld, sc, sh, None, 0, None, None, y, c, rows, c, relu)))
        src._mat = y
    return src._mat


def _bias_grad(plan, dy, spec):
    """dbias = column sum of dy (one launch for short tensors, two-stage otherwise)."""
    rows, c, ld = rows_of(dy)
    if rows <= 8192:
        # one dj_colsum_multi launch for all of these at the end of the backward pass (dy stays as it is until then)
        plan.deferred_colsums.append((dy, rows, c, ld, spec.grad, spec))
        return
    nr = query("dj_reduce_rows", rows)
    partial = plan.empty(nr, 2, c)
    plan.emit_bwd(lambda: call("dj_colsum_partial", dy, rows, c, ld, partial))
    plan.emit_bwd(lambda: call("dj_colreduce_finalize", partial, nr, c, 0, spec.grad, 0))
    plan.note_grad(spec)


# =====================================================================================
# Conv2D / Conv2DTranspose / Dense
# =====================================================================================
def _takes16(model, t):
    """Every reader of Keras tensor `t` can work on a tensor held in 16 bits and write its bf16 gradient: Conv2D layers
    whose three GEMMs take the branch-free reduced-precision kernels, BatchNormalization behind a convolution, Add, and
    Activation('relu') in front of such layers (BASELINE config 5: the conv -> BN -> ReLU -> conv chains and the residual
    sums of the bottleneck blocks, localisation_part/models/keras_ssd300_dct_j2d_resnet.py:46-164).  Everything else
    (L2Normalization, pooling, Concatenate, the predictor heads, model outputs) keeps its input in fp32."""
    users = model.consumers_of(t)
    if not users or any(t is o for o in model.outputs):
        return False
    for u in users:
        if isinstance(u, Activation):
            if u.activation != "relu" or not _takes16(model, u.outbound[0]):
                return False
        elif isinstance(u, BatchNormalization):
            if type(t.layer) is not Conv2D:      # its statistics come from the producing convolution's epilogue
                return False
        elif isinstance(u, Add):
            continue
        elif type(u) is Conv2D:
            if not u.accepts16(t.shape[-1]):
                return False
        else:
            return False
    return True


def _bn_backward_fusable(plan, model, layer, x):
    """-> the BatchNormalization Value whose backward statistics the input-gradient GEMM of `layer` may take (its input `x`
    is that layer's output, optionally behind Activation('relu'), and `layer` is the only reader), else None.
    DJ_FUSE_BNBWD=0 switches it off.  Same box, deconv SSD300 B=32: fp16 step 13.62 -> 13.26 ms, fp32 step 25.15 -> 25.00 ms
    (with fp32 MFMAs the separate pass hid behind the GEMMs of the other stream but took HBM bandwidth from them; the
    extra vector work of the epilogue costs less than that)."""
    if os.environ.get("DJ_FUSE_BNBWD", "1") == "0" or not plan.training:
        return None
    bnv = x if getattr(x, "bn_saved", None) is not None else getattr(x, "bn_parent", None)
    if bnv is None or getattr(bnv, "bn_saved", None) is None or x.alias_of is not None or x.pad is not None:
        return None
    if x is bnv and bnv.relu_child is not None:
        return None
    if len(model.consumers_of(layer.inbound[0])) != 1:
        return None
    return bnv


class Conv2D(Layer):
    """keras.layers.Conv2D(filters, kernel_size, strides=(1,1), padding='valid', dilation_rate=(1,1),
    activation=None, use_bias=True, kernel_initializer='glorot_uniform', kernel_regularizer=None)."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", data_format=None,
                 dilation_rate=(1, 1), activation=None, use_bias=True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, **kwargs):
        super(Conv2D, self).__init__(**kwargs)
        if activation not in (None, "relu", "linear"):
            raise NotImplementedError("Conv2D activation %r" % (activation,))
        if padding not in ("valid", "same"):
            raise ValueError("Invalid border mode for Conv2D: %r" % (padding,))
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding
        self.dilation_rate = _pair(dilation_rate)
        self.activation = None if activation == "linear" else activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel_regularizer = kernel_regularizer

    def build(self, input_shape):
        cin = input_shape[-1]
        self.kernel = self.add_weight("kernel", self.kernel_size + (cin, self.filters), self.kernel_initializer,
                                      regularizer=self.kernel_regularizer)
        self.bias = self.add_weight("bias", (self.filters,), self.bias_initializer) if self.use_bias else None

    def compute_output_shape(self, input_shape):
        _, h, w, _ = input_shape
        _, _, oh, ow = Kn.conv_geometry(h, w, self.kernel_size, self.strides, self.padding, self.dilation_rate)
        return (input_shape[0], oh, ow, self.filters)

    def accepts16(self, in_c):
        """This layer can read a fp16 input with `in_c` channels and write its bf16 gradient: forward, weight gradient and
        input gradient all take the branch-free reduced-precision kernels (csrc/dj_conv.hip: channels that are multiples of
        32, and an input gradient that is stride 1 or the strided 1x1 scatter form)."""
        if in_c % 32 or self.filters % 32:
            return False
        if self.strides != (1, 1) and (self.kernel_size != (1, 1) or self.strides[0] != self.strides[1]):
            return False
        return True

    def _fusable_siblings(self, model, x):
        """The Conv2D layers that read the same tensor with the same geometry and feed neither an activation nor a
        BatchNormalization -- the conf / loc predictor pair of an SSD source
        (localisation_part/models/keras_ssd300_dct_j2d_resnet.py:562-675): candidates for ONE GEMM over the
        concatenated, zero-padded filter bank."""
        if os.environ.get("DJ_FUSE_HEADS", "1") == "0" or x.is_affine or x.pad is not None:
            return [self]
        if getattr(x, "pending_add", None) is not None:
            return [self]
        sibs = []
        for lyr in model.consumers_of(self.inbound[0]):
            if type(lyr) is not Conv2D or lyr.activation is not None or lyr.bias is None:
                continue
            if (lyr.kernel_size, lyr.strides, lyr.padding, lyr.dilation_rate) != (
                    self.kernel_size, self.strides, self.padding, self.dilation_rate):
                continue
            users = model.consumers_of(lyr.outbound[0])
            if any(isinstance(u, (BatchNormalization, Activation)) for u in users):
                continue
            sibs.append(lyr)
        sibs.sort(key=lambda l: l.serial)
        # worth it when a member cannot take the branch-free kernel by itself (rows of 126 / 84 floats) or is too narrow
        # to fill a column tile
        if len(sibs) < 2 or self not in sibs or not any(l.filters % 32 for l in sibs):
            return [self]
        return sibs

    def runs_beside(self, plan, model, ins):
        """True when this layer's forward launches go to the side stream (fused predictor heads whose input carries a
        `ready_event`): Model._plan then does not make the main stream wait for side-stream producers of its inputs."""
        if id(self) in plan.fused_outputs:
            return True
        x = ins[0]
        return plan.forward_side_ok(x) and len(self._fusable_siblings(model, x)) > 1

    def _lower_fused(self, plan, model, x, sibs):
        """One forward GEMM, one input-gradient GEMM and one weight-gradient GEMM for all of `sibs`: their kernels are
        packed side by side into Wp[kh, kw, Cin, Np] (Np = total filters rounded up to 32, padding columns zero) by one
        multi-part copy per step, the outputs / output gradients travel through (B, OH, OW, Np) buffers, and the
        Keras-named kernels, biases and their gradients stay where the optimizer and the checkpoints expect them."""
        b, h, w, cin = x.buf.shape
        n_tot = sum(l.filters for l in sibs)
        n_pad = (n_tot + 31) // 32 * 32
        desc = Kn.make_conv_desc(b, h, w, cin, n_pad, self.kernel_size, self.strides, self.padding, self.dilation_rate)
        desc.algorithmic_out_c = n_tot       # what FLOP accounting may count (bench.py): padding columns are not work
        kh, kw = self.kernel_size
        krows = kh * kw * cin
        rows = b * desc.out_h * desc.out_w
        wp = plan.zeros(kh, kw, cin, n_pad)
        bias_p = plan.zeros(n_pad)
        wp2, offs, o = wp.view(krows, n_pad), [], 0
        for l in sibs:
            offs.append(o)
            o += l.filters
        pack_parts = []
        for l, off in zip(sibs, offs):
            pack_parts.append((l.kernel.param.view(krows, l.filters), l.filters, wp2[:, off:off + l.filters], n_pad, krows,
                               l.filters, 0))
            pack_parts.append((l.bias.param.view(1, l.filters), l.filters, bias_p.view(1, n_pad)[:, off:off + l.filters],
                               n_pad, 1, l.filters, 0))
        # beside the main chain when the input's readiness is known (engine.Plan.emit_side): the heads only feed the
        # prediction assembly at the end of the forward pass
        after = x.ready_event if plan.forward_side_ok(x) else None
        emit = (lambda fn: plan.emit_side(fn, after)) if after is not None else plan.emit
        emit(engine.copy2d_multi(pack_parts))
        y_p = plan.empty(b, desc.out_h, desc.out_w, n_pad)
        xbuf = x.buf
        ws = plan.conv_workspace(desc, side=after is not None)     # split-K through slabs: no atomics, no cleared y
        plan.emit_conv(0, desc, Kn.bound("conv2d_fwd", desc, xbuf, wp, bias_p, y_p, None, None, False, False, None, ws=ws),
                       fwd_after=after)
        y_p2 = y_p.view(rows, n_pad)
        outs, unpack = [], []
        for l, off in zip(sibs, offs):
            y = plan.empty(b, desc.out_h, desc.out_w, l.filters)
            unpack.append((y_p2[:, off:off + l.filters], n_pad, y.view(rows, l.filters), l.filters, rows, l.filters, 0))
            v = Value(y, needs_grad=True, name=l.name)
            outs.append(v)
            plan.fused_outputs[id(l)] = v
        emit(engine.copy2d_multi(unpack))
        if after is not None:
            plan.side_results(outs)

        def build_backward():
            live = [(l, off, v) for l, off, v in zip(sibs, offs, outs) if v.grad is not None]
            if not live:
                return
            dy_p = plan.zeros(b, desc.out_h, desc.out_w, n_pad)     # columns nobody writes (padding, dead members) stay zero
            dy_p2, pack = dy_p.view(rows, n_pad), []
            for l, off, v in live:
                assert v.grad.mask_y is None
                dy = v.grad.buf
                pack.append((dy.view(rows, l.filters), l.filters, dy_p2[:, off:off + l.filters], n_pad, rows, l.filters, 0))
                if l.bias.trainable:
                    _bias_grad(plan, dy, l.bias)
            plan.emit_bwd(engine.copy2d_multi(pack))
            train = [(l, off) for l, off, _ in live if l.kernel.trainable]
            if train:
                dwp = plan.zeroed_each_step(kh, kw, cin, n_pad)
                dwp2 = dwp.view(krows, n_pad)
                split = engine.copy2d_multi([(dwp2[:, off:off + l.filters], n_pad, l.kernel.grad.view(krows, l.filters),
                                              l.filters, krows, l.filters, 0) for l, off in train])

                wg = Kn.bound("conv2d_wgrad", desc, xbuf, dy_p, dwp, dw_zeroed=True)

                def wgrad():
                    wg()
                    split()          # same stream, right behind the GEMM
                plan.emit_conv(2, desc, wgrad, backward=True, side=True)
                for l, _ in train:
                    plan.note_grad(l.kernel)
            if x.needs_grad:
                own_memset = (engine.tuned_splits(1, desc) or 1) > 1
                dx, beta = plan.grad_of(x, zeroed=own_memset and os.environ.get("DJ_ZERO_ARENA", "1") != "0")
                plan.emit_conv(1, desc, Kn.bound("conv2d_dgrad", desc, dy_p, wp, dx, None, bool(beta)), backward=True)

        plan.on_backward(build_backward)
        return plan.fused_outputs[id(self)]

    def lower(self, plan, model, ins):
        x = ins[0]
        if id(self) in plan.fused_outputs:          # lowered together with an earlier sibling
            return plan.fused_outputs[id(self)]
        sibs = self._fusable_siblings(model, x)
        if len(sibs) > 1:
            return self._lower_fused(plan, model, x, sibs)
        b, h, w, cin = x.buf.shape
        padding = self.padding
        if x.pad is not None:
            if padding != "valid":
                raise NotImplementedError("ZeroPadding2D followed by a 'same' Conv2D")
            padding = x.pad
        desc = Kn.make_conv_desc(b, h, w, cin, self.filters, self.kernel_size, self.strides, padding,
                                 self.dilation_rate)
        relu = self.activation == "relu"
        wgt, bias = self.kernel.param, (self.bias.param if self.bias is not None else None)
        # float16 mode: the GEMMs read the per-step 16-bit shadows of the weights (fp16 forward, bf16 input gradient) where
        # their launch takes the branch-free reduced-precision kernel (a 16-bit operand is an error anywhere else)
        w_fwd = w_bwd = wgt
        if plan.store16 and os.environ.get("DJ_WSHADOW", "1") != "0" and x.pad is None:
            unit = self.strides == (1, 1) or (self.kernel_size == (1, 1) and self.strides[0] == self.strides[1])
            if cin % 32 == 0 and self.filters % 4 == 0:
                model.weight_shadows(plan)
                w_fwd = self.kernel.param16
            if self.filters % 32 == 0 and cin % 4 == 0 and unit:
                model.weight_shadows(plan)
                w_bwd = self.kernel.parambf
        pro = (x.scale, x.shift, x.relu) if x.is_affine else (None, None, False)
        stats = fused_bn = None
        consumers = model.consumers_of(self.outbound[0])
        bn_consumer = (plan.training and not relu and len(consumers) == 1 and isinstance(consumers[0], BatchNormalization))
        # Fewer than two 64x64 output tiles per CU (the 5x5 / 10x10 stages at batch 32): such a GEMM only fills the chip
        # evenly when its reduction is split over workgroups (400 equal tiles on 256 CUs: 144 CUs carry two, the launch
        # takes two tile times for 1.56 of work), and a split launch cannot take the BatchNormalization statistics in its
        # epilogue (every workgroup holds a partial sum).  There the statistics come from the fixed-order reduction of
        # the split launch's slabs instead (dj_conv2d_nhwc_fwd_ws with DJ_CONV_STATS_MAY_SPLIT).
        # Same box, deconv B=32: limit 256 -> 25.08 ms, 512..700 -> 24.94, 1500 -> 25.12.
        tile_limit = int(os.environ.get("DJ_SPLIT_SMALL_BN_TILES", "512"))
        few_tiles = (-(-(b * desc.out_h * desc.out_w) // 64)) * (-(-self.filters // 64)) < tile_limit
        split_instead = bn_consumer and few_tiles and os.environ.get("DJ_SPLIT_SMALL_BN", "1") != "0"
        if bn_consumer:
            if os.environ.get("DJ_FUSE_BNFIN", "0") == "1" and not split_instead:
                # opt-in: the conv's last workgroup turns the column sums into the BatchNormalization coefficients itself
                # (fp64 accumulators + a ticket, both left zero by that workgroup).  Saves the finalize launch but every
                # workgroup pays a ticket round trip: 0.6 % SLOWER on the SSD300 step, hence off
                bnl, c = consumers[0], self.filters
                fused_bn = dict(scale=plan.empty(c), shift=plan.empty(c), mean=plan.empty(c), invstd=plan.empty(c))
                acc = torch.zeros(Kn.BN_ACC_REPLICAS * 2 * c, dtype=torch.float64, device=plan.device)
                ticket = torch.zeros(1, dtype=torch.int32, device=plan.device)
                bn_arg = Kn.make_bn_train(acc, ticket, bnl.gamma.param, bnl.beta.param, bnl.moving_mean.param,
                                          bnl.moving_variance.param, fused_bn["scale"], fused_bn["shift"],
                                          fused_bn["mean"], fused_bn["invstd"], bnl.epsilon, bnl.momentum)
            else:
                nrows = Kn.conv2d_stats_rows(desc)
                stats = plan.empty(nrows, 2, self.filters)
        # a split-K forward (the small-M layers) leaves its partial tiles in the plan's workspace and a fixed-order
        # reduction writes y (and takes the statistics of a `split_instead` launch): bit-reproducible, no cleared y
        # Raw output in fp16 (config 5 proper) when it feeds nothing but a BatchNormalization whose readers take 16-bit
        # tensors and this layer's own gradient GEMMs can read the bf16 gradient that comes back
        y_dtype = torch.float32
        if (plan.store16 and len(consumers) == 1 and isinstance(consumers[0], BatchNormalization) and not relu
                and not split_instead and fused_bn is None and self.accepts16(cin)
                and _takes16(model, consumers[0].outbound[0])):
            y_dtype = plan.act_dtype(b * desc.out_h * desc.out_w, self.filters)
        if x.buf.dtype != torch.float32 and not self.accepts16(cin):
            raise NotImplementedError("%s cannot read a 16-bit input (lowering bug: _takes16 said it could)" % self.name)
        y = plan.empty(b, desc.out_h, desc.out_w, self.filters, dtype=y_dtype)
        ws = plan.conv_workspace(desc, stats_may_split=split_instead)
        xbuf = x.buf
        pend = getattr(x, "pending_add", None)
        if pend is not None:
            # x = relu(bn(z) + shortcut) has not been computed yet: this conv evaluates it while staging its A tile and
            # writes it to xbuf for everybody else (Add.lower picked this layer because it runs first)
            assert pend["consumer"] is self, "a residual sum must be materialised by its first consumer"
            zb, zs, zt, rb, rs, rt = pend["z"], pend["z_scale"], pend["z_shift"], pend["res"], pend["res_scale"], pend["res_shift"]
            if fused_bn is not None:
                plan.emit_conv(4, desc, Kn.bound("conv2d_fwd_bn", desc, zb, wgt, bias, y, bn_arg, zs, zt, True, rb, rs, rt, xbuf))
            else:
                plan.emit_conv(4 if stats is not None else 0, desc,
                               Kn.bound("conv2d_fwd_addrelu", desc, zb, w_fwd, bias, y, zs, zt, rb, rs, rt, xbuf, relu, stats,
                                        ws=ws))
            x.pending_add = None
            plan.mark_ready(x)     # the sum exists from here on: its other readers may run beside the main chain
        elif fused_bn is not None:
            plan.emit_conv(4, desc, Kn.bound("conv2d_fwd_bn", desc, xbuf, wgt, bias, y, bn_arg, pro[0], pro[1], pro[2]))
        else:
            plan.emit_conv(4 if (stats is not None and not split_instead) else 0, desc,
                           Kn.bound("conv2d_fwd", desc, xbuf, w_fwd if fused_bn is None else wgt, bias, y, pro[0], pro[1], pro[2],
                                    relu, stats, ws=ws, stats_may_split=split_instead))
        out = Value(y, needs_grad=True, name=self.name)
        if stats is not None:
            out.conv_stats = (stats, stats.shape[0], bias)
        out.bn_done = fused_bn

        def build_backward():
            if out.grad is None:
                return
            assert out.grad.mask_y is None
            dy = out.grad.buf
            if relu:
                rows, c, ld = rows_of(dy)
                plan.emit_bwd(launcher("dj_relu_bwd", dy, ld, y, c, dy, ld, rows, c, 0))
            if self.bias is not None and self.bias.trainable:
                if bn_consumer:
                    # the only consumer is a training-mode BatchNormalization: it subtracts the batch mean, so
                    # d loss / d bias = sum(dz) is identically zero (TF's autodiff returns rounding noise);
                    # the gradient buffer is zero-initialised and simply left untouched
                    plan.note_grad(self.bias)
                else:
                    _bias_grad(plan, dy, self.bias)
            if self.kernel.trainable:
                dw = self.kernel.grad
                plan.emit_conv(2, desc, Kn.bound("conv2d_wgrad", desc, xbuf, dy, dw, pro[0], pro[1], pro[2],
                                                 dw_zeroed=plan.grads_cleared), backward=True, side=True)
                plan.note_grad(self.kernel)
            if x.needs_grad:
                # a first-writer dgrad that accumulates with atomics (split-K) or scatters (stride-2 1x1) clears dx with
                # a memset of its own; hand it a buffer from the arena that one memset clears per step instead
                strided_1x1 = self.kernel_size == (1, 1) and self.strides != (1, 1)
                own_memset = strided_1x1 or (engine.tuned_splits(1, desc) or 1) > 1
                bnv = _bn_backward_fusable(plan, model, self, x)
                if bnv is not None and not own_memset and x.grad is None:
                    # x = [relu](bn(z)) and this convolution is its only reader: the input-gradient GEMM takes that
                    # BatchNormalization's backward statistics in its epilogue (dj_conv2d_nhwc_dgrad_bnbwd), the layer's
                    # own pass over (gradient, z) -- dj_bn_bwd_reduce -- is not launched
                    dx, beta = plan.grad_of(x)
                    assert beta == 0
                    mean, invstd = bnv.bn_saved
                    nr = (b * h * w + 63) // 64
                    part = plan.empty(nr, 2, cin)
                    msc, msh = (x.scale, x.shift) if x.relu else (None, None)
                    zbuf = x.buf
                    fused_dgrad = Kn.bound("conv2d_dgrad_bnbwd", desc, dy, w_bwd, dx, zbuf, mean, invstd, msc, msh, part)
                    fused_dgrad.no_split = True      # for the tuners: a registered split-K factor is ignored here
                    plan.emit_conv(9, desc, fused_dgrad, backward=True)   # tuned and recorded apart from plain dgrads
                    x.grad.bwd_partial = (part, nr)
                else:
                    dx, beta = plan.grad_of(x, zeroed=own_memset and os.environ.get("DJ_ZERO_ARENA", "1") != "0")
                    plan.emit_conv(1, desc, Kn.bound("conv2d_dgrad", desc, dy, w_bwd, dx, None, bool(beta)), backward=True)

        plan.on_backward(build_backward)
        return out


class Conv2DTranspose(Layer):
    """keras.layers.Conv2DTranspose(filters, kernel_size, strides), padding 'valid', kernel layout
    (kh, kw, out, in) (localisation_part/models/keras_ssd300_dct_j2d_resnet.py:1709-1711).
    Forward = input-gradient kernel of the k x k / stride-s convolution it transposes."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", kernel_regularizer=None, **kwargs):
        super(Conv2DTranspose, self).__init__(**kwargs)
        if padding != "valid" or activation not in (None, "linear"):
            raise NotImplementedError("Conv2DTranspose: only padding='valid', no activation")
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel_regularizer = kernel_regularizer

    def build(self, input_shape):
        cin = input_shape[-1]
        self.kernel = self.add_weight("kernel", self.kernel_size + (self.filters, cin), self.kernel_initializer,
                                      regularizer=self.kernel_regularizer)
        self.bias = self.add_weight("bias", (self.filters,), self.bias_initializer) if self.use_bias else None

    def compute_output_shape(self, input_shape):
        _, h, w, _ = input_shape
        oh = (h - 1) * self.strides[0] + self.kernel_size[0]
        ow = (w - 1) * self.strides[1] + self.kernel_size[1]
        return (input_shape[0], oh, ow, self.filters)

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan)
        b, h, w, cin = xbuf.shape
        oh = (h - 1) * self.strides[0] + self.kernel_size[0]
        ow = (w - 1) * self.strides[1] + self.kernel_size[1]
        # the transposed layer's output is the "input" of the underlying convolution
        desc = Kn.make_conv_desc(b, oh, ow, self.filters, cin, self.kernel_size, self.strides, "valid", (1, 1))
        assert (desc.out_h, desc.out_w) == (h, w)
        y = plan.empty(b, oh, ow, self.filters)
        wgt, bias = self.kernel.param, (self.bias.param if self.bias is not None else None)
        plan.emit_conv(1, desc, Kn.bound("conv2d_dgrad", desc, xbuf, wgt, y, bias, False, no_split=True))
        out = Value(y, needs_grad=True, name=self.name)

        def build_backward():
            if out.grad is None:
                return
            assert out.grad.mask_y is None
            dy = out.grad.buf
            if self.bias is not None and self.bias.trainable:
                _bias_grad(plan, dy, self.bias)
            if self.kernel.trainable:
                dw = self.kernel.grad
                plan.emit_conv(2, desc, Kn.bound("conv2d_wgrad", desc, dy, xbuf, dw, dw_zeroed=plan.grads_cleared),
                               backward=True)
                plan.note_grad(self.kernel)
            if x.needs_grad:
                dx, beta = plan.grad_of(x)
                if beta:
                    raise NotImplementedError("accumulating Conv2DTranspose input gradient")
                plan.emit_conv(0, desc, Kn.bound("conv2d_fwd", desc, dy, wgt, None, dx), backward=True)

        plan.on_backward(build_backward)
        return out


class Dense(Layer):
    """keras.layers.Dense(units, activation) -- `fc1000`
    (classification_part/vgg_jpeg_keras/networks/resnet_dct.py:417).  Runs as a 1x1 convolution."""

    def __init__(self, units, activation=None, use_bias=True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", kernel_regularizer=None, **kwargs):
        super(Dense, self).__init__(**kwargs)
        if activation not in (None, "linear", "softmax", "relu"):
            raise NotImplementedError("Dense activation %r" % (activation,))
        self.units = int(units)
        self.activation = None if activation == "linear" else activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel_regularizer = kernel_regularizer

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", (input_shape[-1], self.units), self.kernel_initializer,
                                      regularizer=self.kernel_regularizer)
        self.bias = self.add_weight("bias", (self.units,), self.bias_initializer) if self.use_bias else None

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan)
        assert xbuf.dim() == 2
        b, cin = xbuf.shape
        desc = Kn.make_conv_desc(b, 1, 1, cin, self.units, (1, 1))
        x4 = xbuf.view(b, 1, 1, cin)
        z = plan.empty(b, 1, 1, self.units)
        wgt = self.kernel.param.view(1, 1, cin, self.units)
        bias = self.bias.param if self.bias is not None else None
        relu = self.activation == "relu"
        ws = plan.conv_workspace(desc)
        plan.emit(Kn.bound("conv2d_fwd", desc, x4, wgt, bias, z, relu=relu, ws=ws))
        zv = Value(z.view(b, self.units), needs_grad=True, name=self.name)
        out = zv
        if self.activation == "softmax":
            out = _lower_softmax(plan, zv, self.name + "/softmax")

        def build_backward():
            if zv.grad is None:
                return
            dy2 = zv.grad.buf
            dy = dy2.view(b, 1, 1, self.units)
            if relu:
                plan.emit_bwd(lambda: call("dj_relu_bwd", dy2, self.units, zv.buf, self.units, dy2, self.units, b,
                                           self.units, 0))
            if self.bias is not None and self.bias.trainable:
                _bias_grad(plan, dy2, self.bias)
            if self.kernel.trainable:
                dw = self.kernel.grad.view(1, 1, cin, self.units)
                plan.emit_bwd(Kn.bound("conv2d_wgrad", desc, x4, dy, dw))
                plan.note_grad(self.kernel)
            if x.needs_grad:
                dx, beta = plan.grad_of(x)
                dx4 = dx.view(b, 1, 1, cin)
                plan.emit_bwd(Kn.bound("conv2d_dgrad", desc, dy, wgt, dx4, None, bool(beta)))

        # registered before the softmax's builder would be wrong: softmax must run first in backward,
        # so the Dense builder is registered first (builders run in reverse registration order)
        plan._bwd_builders.insert(len(plan._bwd_builders) - (1 if self.activation == "softmax" else 0),
                                  build_backward)
        return out


def _lower_softmax(plan, x, name):
    xbuf = _materialised(x, name, plan)
    c = xbuf.shape[-1]
    rows = xbuf.numel() // c
    assert xbuf.is_contiguous()
    y = plan.empty(*xbuf.shape)
    plan.emit(lambda: call("dj_softmax_fwd", xbuf, y, rows, c))
    out = Value(y, needs_grad=x.needs_grad, name=name)

    def build_backward():
        if out.grad is None or not x.needs_grad:
            return
        assert out.grad.mask_y is None
        dp = out.grad.buf
        assert dp.is_contiguous()
        dx, beta = plan.grad_of(x)
        plan.emit_bwd(lambda: call("dj_softmax_bwd", y, dp, c, dx, rows, c, beta))

    plan.on_backward(build_backward)
    return out


# =====================================================================================
# BatchNormalization / Activation / Add
# =====================================================================================
class BatchNormalization(Layer):
    """keras.layers.BatchNormalization(axis=-1, momentum=0.99, epsilon=1e-3): in training mode batch
    statistics over (N,H,W); inference uses the moving statistics.  No launch of its own in the
    forward pass beyond the per-channel finalize: the result is a virtual affine Value."""

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, **kwargs):
        super(BatchNormalization, self).__init__(**kwargs)
        if axis not in (-1, 3):
            raise NotImplementedError("BatchNormalization axis %r (the reference uses bn_axis = 3)" % (axis,))
        if not (center and scale):
            raise NotImplementedError("BatchNormalization without gamma/beta")
        self.axis = axis
        self.momentum = float(momentum)
        self.epsilon = float(epsilon)

    def build(self, input_shape):
        c = input_shape[-1]
        self.gamma = self.add_weight("gamma", (c,), "ones")
        self.beta = self.add_weight("beta", (c,), "zeros")
        self.moving_mean = self.add_weight("moving_mean", (c,), "zeros", trainable=False)
        self.moving_variance = self.add_weight("moving_variance", (c,), "ones", trainable=False)

    def lower(self, plan, model, ins):
        x = ins[0]
        z = _materialised(x, self.name, plan)
        rows, c, ld = rows_of(z)
        done = getattr(x, "bn_done", None) if plan.training else None
        scale, shift = (done["scale"], done["shift"]) if done else (plan.empty(c), plan.empty(c))
        gamma, beta = self.gamma.param, self.beta.param
        mm, mv = self.moving_mean.param, self.moving_variance.param
        out = Value(z, scale=scale, shift=shift, relu=False, needs_grad=True, name=self.name)
        out.bn = self
        out.bn_applies_mask = bool(plan.training)   # its backward consumes GradRef.mask_y / .also (see Add)
        if not plan.training:
            plan.emit(lambda: call("dj_bn_infer_coeffs", gamma, beta, mm, mv, self.epsilon, scale, shift, c))
            return out
        if done:
            mean, invstd = done["mean"], done["invstd"]   # written by the producing convolution's last workgroup
        else:
            mean, invstd = plan.empty(c), plan.empty(c)
            if x.conv_stats is not None:
                partial, nrows, conv_bias = x.conv_stats
            else:
                assert z.dtype == torch.float32, "BatchNormalization statistics of a 16-bit tensor come from its producer"
                nrows = query("dj_reduce_rows", rows)
                partial, conv_bias = plan.empty(nrows, 2, c), None
                plan.emit(lambda: call("dj_colstats_partial", z, rows, c, ld, partial))
            plan.emit(launcher("dj_bn_train_finalize", partial, nrows, rows, conv_bias, gamma, beta, self.epsilon,
                               self.momentum, mm, mv, scale, shift, mean, invstd, c))
        out.bn_saved = (mean, invstd)   # a consumer convolution may take this layer's backward statistics itself

        def build_backward():
            src = out.relu_child if out.relu_child is not None else out
            if out.relu_child is not None and out.grad is not None:
                raise NotImplementedError("BatchNormalization output used both with and without its ReLU")
            if src.grad is None:
                return
            dy, mask_y = src.grad.buf, src.grad.mask_y
            if out.relu_child is not None:
                assert mask_y is None
                mode = 2
            else:
                mode = 1 if mask_y is not None else 0
            r2, c2, ld_dy = rows_of(dy)
            assert (r2, c2) == (rows, c)
            ld_y = rows_of(mask_y)[2] if mask_y is not None else 0
            k0, k1, k2 = plan.empty(c), plan.empty(c), plan.empty(c)
            dgamma, dbeta = self.gamma.grad, self.beta.grad
            if dgamma is None:  # frozen layer: scratch
                dgamma, dbeta = plan.empty(c), plan.empty(c)
            fused = getattr(src.grad, "bwd_partial", None)
            if fused is not None:
                # the convolution that produced dy took the statistics in its epilogue (Conv2D.build_backward)
                assert mode in (0, 2)
                part, nr = fused
            else:
                nr = query("dj_reduce_rows", rows)
                part = plan.empty(nr, 2, c)
                plan.emit_bwd(launcher(*Kn.bn_bwd_reduce_call(dy, ld_dy, z, ld, mask_y, ld_y, mean, invstd, scale, shift, mode,
                                                              rows, c, part)))
            plan.emit_bwd(lambda: call("dj_bn_bwd_finalize", part, nr, rows, gamma, mean, invstd, dgamma, dbeta, k0,
                                       k1, k2, c))
            plan.note_grad(self.gamma)
            plan.note_grad(self.beta)
            also = src.grad.also
            if x.needs_grad:
                dz, beta_acc = plan.grad_of(x)
                if beta_acc:
                    raise NotImplementedError("BatchNormalization input with several gradient writers")
                ld_dz = rows_of(dz)[2]
                dm, dm_beta = also if also is not None else (None, 0)
                ld_dm = rows_of(dm)[2] if dm is not None else 0
                app = Kn.bn_bwd_apply_call(dy, ld_dy, z, ld, mask_y, ld_y, scale, shift, mode, k0, k1, k2, dz, ld_dz, rows, c,
                                           dm, ld_dm, int(dm_beta))
                plan.emit_bwd(lambda: call(*app))
            elif also is not None:   # nothing to apply here: the shortcut still needs its masked gradient
                dm, dm_beta = also
                plan.emit_bwd(launcher(*Kn.relu_bwd_call(dy, ld_dy, mask_y, ld_y, dm, rows_of(dm)[2], rows, c, int(dm_beta))))

        plan.on_backward(build_backward)
        return out


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super(Activation, self).__init__(**kwargs)
        if activation not in ("relu", "softmax", "linear"):
            raise NotImplementedError("Activation %r" % (activation,))
        self.activation = activation
        self.absorbed = False  # set by Add when it fuses this ReLU

    def lower(self, plan, model, ins):
        x = ins[0]
        if self.activation == "linear" or self.absorbed:
            return x
        if self.activation == "softmax":
            return _lower_softmax(plan, x, self.name)
        if x.is_affine and not x.relu:
            if x.relu_child is not None:
                raise NotImplementedError("two ReLUs on one BatchNormalization output")
            v = Value(x.buf, scale=x.scale, shift=x.shift, relu=True, needs_grad=x.needs_grad, name=self.name)
            x.relu_child = v
            v.bn_parent = x
            return v
        xbuf = _materialised(x, self.name, plan)
        rows, c, ld = rows_of(xbuf)
        y = plan.empty(*xbuf.shape)
        plan.emit(launcher(*Kn.affine_act_call(xbuf, ld, None, None, None, 0, None, None, y, c, rows, c, 1)))
        out = Value(y, needs_grad=x.needs_grad, name=self.name)

        def build_backward():
            if out.grad is None or not x.needs_grad:
                return
            dy = out.grad.buf
            dx, beta = plan.grad_of(x)
            plan.emit_bwd(launcher(*Kn.relu_bwd_call(dy, rows_of(dy)[2], y, c, dx, rows_of(dx)[2], rows, c, beta)))

        plan.on_backward(build_backward)
        return out


class Add(Layer):
    """keras.layers.Add()([x, shortcut]) followed by Activation('relu')
    (localisation_part/models/keras_ssd300_dct_j2d_resnet.py:98-99,162-163): one pass that applies
    the BatchNormalization affines of both branches, adds, and clamps."""

    def compute_output_shape(self, input_shape):
        return input_shape[0]

    @staticmethod
    def _first_consumer_conv(plan, model, relu_layer, a, y, b=None):
        """The layer that runs first among the consumers of relu(Add) if it can take the sum as a fused prologue."""
        if os.environ.get("DJ_FUSE_ADD", "1") == "0" or relu_layer is None or not a.is_affine:
            return None
        if not (a.buf.is_contiguous() and y.dim() == 4):
            return None
        if b is not None and a.buf.dtype != b.buf.dtype:      # the fused prologue reads both operands in one storage type
            return None
        users = model.consumers_of(relu_layer.outbound[0])
        if not users:
            return None
        first = min(users, key=model.layers.index)
        if not (isinstance(first, Conv2D) and not isinstance(first, Conv2DTranspose) and first.kernel_size == (1, 1)
                and first.strides == (1, 1) and first.dilation_rate == (1, 1)):
            return None
        b_, h, w, cin = y.shape
        desc = Kn.make_conv_desc(b_, h, w, cin, first.filters, (1, 1), (1, 1), "valid", (1, 1))
        return first if Kn.conv2d_fwd_addrelu_supported(desc) else None

    def lower(self, plan, model, ins):
        a, b = ins
        if not a.is_affine and b.is_affine:
            a, b = b, a
        for v in (a, b):
            if v.relu or v.pad is not None:
                raise NotImplementedError("Add over a ReLU-ed virtual tensor")
        consumers = model.consumers_of(self.outbound[0])
        relu = len(consumers) == 1 and isinstance(consumers[0], Activation) and consumers[0].activation == "relu"
        if relu:
            consumers[0].absorbed = True
        rows, c, lda = rows_of(a.buf)
        ldb = rows_of(b.buf)[2]
        # the block sum in fp16 (config 5 proper) when everybody who reads it takes 16-bit tensors
        y_dtype = plan.act_dtype(rows, c) if (plan.store16 and _takes16(model, self.outbound[0])) else torch.float32
        y = plan.empty(*a.buf.shape, dtype=y_dtype)
        abuf, bbuf = a.buf, b.buf
        out = Value(y, needs_grad=a.needs_grad or b.needs_grad, name=self.name)
        first = self._first_consumer_conv(plan, model, consumers[0] if relu else None, a, y, b) if relu else None
        if first is not None:
            # no launch here: `first` (the next block's 1x1 conv) computes relu(bn(a) + b) in its A-tile prologue and
            # stores it to y (dj_conv2d_nhwc_fwd_addrelu) -- one elementwise pass and one read of y less per block
            out.pending_add = dict(consumer=first, z=abuf, z_scale=a.scale, z_shift=a.shift, res=bbuf, res_scale=b.scale,
                                   res_shift=b.shift)
        else:
            plan.emit(launcher(*Kn.affine_act_call(abuf, lda, a.scale, a.shift, bbuf, ldb, b.scale, b.shift, y, c, rows, c,
                                                   int(relu))))

        def build_backward():
            if out.grad is None:
                return
            assert out.grad.mask_y is None
            dy = out.grad.buf
            # identity block: the BatchNormalization branch applies the ReLU mask anyway (dj_bn_bwd_apply); it writes the
            # masked gradient for the identity shortcut in the same pass instead of a separate dj_relu_bwd sweep
            fuse = (relu and a.is_affine and a.needs_grad and not b.is_affine and b.needs_grad
                    and getattr(a, "bn_applies_mask", False))
            for v in (a, b):
                if not v.needs_grad:
                    continue
                if v.is_affine:
                    also = plan.grad_of(b) if (fuse and v is a) else None
                    plan.set_grad_ref(v, GradRef(dy, y if relu else None, also=also))
                elif fuse:
                    continue
                else:
                    dv, beta = plan.grad_of(v)
                    if relu:
                        bw = Kn.relu_bwd_call(dy, c, y, c, dv, rows_of(dv)[2], rows, c, beta)
                    else:
                        bw = Kn.copy2d_call(dy, c, dv, rows_of(dv)[2], rows, c, beta)
                    plan.emit_bwd(launcher(*bw))

        plan.on_backward(build_backward)
        return out


# =====================================================================================
# shape / routing layers
# =====================================================================================
class Concatenate(Layer):
    def __init__(self, axis=-1, **kwargs):
        super(Concatenate, self).__init__(**kwargs)
        self.axis = axis

    def compute_output_shape(self, input_shape):
        nd = len(input_shape[0])
        ax = self.axis % nd
        out = list(input_shape[0])
        out[ax] = sum(s[ax] for s in input_shape)
        for s in input_shape:
            if [d for i, d in enumerate(s) if i != ax] != [d for i, d in enumerate(out) if i != ax]:
                raise ValueError("A `Concatenate` layer requires inputs with matching shapes except for the concat "
                                 "axis. Got inputs shapes: %s" % (input_shape,))
        return tuple(out)

    def lower(self, plan, model, ins):
        bufs = [_materialised(v, self.name, plan) for v in ins]
        v_of = {id(t): v for t, v in zip(bufs, ins)}
        nd = bufs[0].dim()
        ax = self.axis % nd
        assert ax >= 1
        outer = 1
        for d in bufs[0].shape[:ax]:
            outer *= d
        inner = [t.numel() // outer for t in bufs]
        total = sum(inner)
        shape = list(bufs[0].shape)
        shape[ax] = sum(t.shape[ax] for t in bufs)
        y = plan.empty(*shape)
        yflat = y.view(outer, total)
        offs, o = [], 0
        for n in inner:
            offs.append(o)
            o += n
        live_parts = []
        for t, n, off in zip(bufs, inner, offs):
            if ax == nd - 1:
                r, c, ld = rows_of(t)     # may be a channel slice
                src, lds, rws = t, ld, r
            else:
                assert t.is_contiguous()
                src, lds, rws = t, n, outer
            dst = yflat[:, off:off + n]
            if getattr(v_of[id(t)], "constant", False):
                # constant input (the anchor boxes): its slice of y is written once, now (not in a structure-only
                # lowering on the host, which launches nothing)
                if plan.device.type == "cuda":
                    call("dj_copy2d", src, lds, dst, total, rws, n, 0)
            else:
                live_parts.append((src, lds, dst, total, rws, n, 0))
        if len(live_parts) == 1:
            src, lds, dst, _, rws, n, _ = live_parts[0]
            plan.emit(lambda: call("dj_copy2d", src, lds, dst, total, rws, n, 0))
        elif live_parts:
            plan.emit(engine.copy2d_multi(live_parts))     # one launch for all members
        out = Value(y, needs_grad=any(v.needs_grad for v in ins), name=self.name)
        out.constant = all(getattr(v, "constant", False) for v in ins)

        def build_backward():
            if out.grad is None:
                return
            assert out.grad.mask_y is None
            gflat = out.grad.buf.view(outer, total)
            parts = []
            for v, n, off in zip(ins, inner, offs):
                if not v.needs_grad:
                    continue
                dv, beta = plan.grad_of(v)
                src = gflat[:, off:off + n]
                if ax == nd - 1:
                    r, c, ld = rows_of(dv)
                    parts.append((src, total, dv, ld, r, n, beta))
                else:
                    assert dv.is_contiguous()
                    parts.append((src, total, dv, n, outer, n, beta))
            if len(parts) == 1:
                src, _, dv, ld, r, n, beta = parts[0]
                plan.emit_bwd(lambda: call("dj_copy2d", src, total, dv, ld, r, n, beta))
            elif parts:
                plan.emit_bwd(engine.copy2d_multi(parts))

        plan.on_backward(build_backward)
        return out


class Reshape(Layer):
    def __init__(self, target_shape, **kwargs):
        super(Reshape, self).__init__(**kwargs)
        self.target_shape = tuple(target_shape)

    def compute_output_shape(self, input_shape):
        known = 1
        for d in input_shape[1:]:
            known *= d
        tgt = list(self.target_shape)
        if -1 in tgt:
            i = tgt.index(-1)
            rest = 1
            for j, d in enumerate(tgt):
                if j != i:
                    rest *= d
            tgt[i] = known // rest
        return (input_shape[0],) + tuple(tgt)

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan)
        assert xbuf.is_contiguous()
        shape = (xbuf.shape[0],) + tuple(self.output_shape[1:])
        out = Value(xbuf.view(*shape), needs_grad=x.needs_grad, name=self.name)
        out.alias_of = x
        out.alias_view = lambda g: g.view(*shape)
        out.constant = getattr(x, "constant", False)
        return out


class Flatten(Reshape):
    def __init__(self, **kwargs):
        super(Flatten, self).__init__((-1,), **kwargs)


class ZeroPadding2D(Layer):
    """Folded into the following `valid` Conv2D's padding
    (localisation_part/models/keras_ssd300_dct_j2d_resnet.py:514,1143,1166)."""

    def __init__(self, padding=(1, 1), **kwargs):
        super(ZeroPadding2D, self).__init__(**kwargs)
        if isinstance(padding, int):
            padding = ((padding, padding), (padding, padding))
        elif isinstance(padding[0], int):
            padding = ((padding[0], padding[0]), (padding[1], padding[1]))
        self.padding = (tuple(padding[0]), tuple(padding[1]))

    def compute_output_shape(self, input_shape):
        b, h, w, c = input_shape
        return (b, h + sum(self.padding[0]), w + sum(self.padding[1]), c)

    def lower(self, plan, model, ins):
        x = ins[0]
        if x.pad is not None:
            raise NotImplementedError("stacked ZeroPadding2D")
        out = Value(x.buf, scale=x.scale, shift=x.shift, relu=x.relu, needs_grad=x.needs_grad, name=self.name)
        out.pad = self.padding
        out.alias_of = x
        out.alias_view = lambda g: g
        return out


class MaxPooling2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", **kwargs):
        super(MaxPooling2D, self).__init__(**kwargs)
        self.pool_size = _pair(pool_size)
        self.strides = _pair(strides) if strides is not None else self.pool_size
        self.padding = padding

    def compute_output_shape(self, input_shape):
        b, h, w, c = input_shape
        _, _, oh, ow = Kn.conv_geometry(h, w, self.pool_size, self.strides, self.padding, (1, 1))
        return (b, oh, ow, c)

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan, allow_pad=True)
        assert xbuf.is_contiguous()
        b, h, w, c = xbuf.shape
        padding, pad_zero = self.padding, 0
        if x.pad is not None:   # ZeroPadding2D in front: zeros take part in the max
            if padding != "valid":
                raise NotImplementedError("ZeroPadding2D followed by a 'same' MaxPooling2D")
            padding, pad_zero = x.pad, 1
        pt, pl, oh, ow = Kn.conv_geometry(h, w, self.pool_size, self.strides, padding, (1, 1))
        kh, kw = self.pool_size
        sh, sw = self.strides
        y = plan.empty(b, oh, ow, c)
        amax = None
        if plan.training and x.needs_grad:
            amax = torch.empty(b * oh * ow * c, dtype=torch.uint8, device=plan.device)
        plan.emit(lambda: call("dj_maxpool2d_fwd", xbuf, y, b, h, w, c, oh, ow, kh, kw, sh, sw, pt, pl, pad_zero, amax))
        out = Value(y, needs_grad=x.needs_grad, name=self.name)

        def build_backward():
            if out.grad is None or not x.needs_grad:
                return
            dy = out.grad.buf
            dx, beta = plan.grad_of(x)
            assert dx.is_contiguous() and dy.is_contiguous()
            plan.emit_bwd(lambda: call("dj_maxpool2d_bwd", xbuf, dy, dx, b, h, w, c, oh, ow, kh, kw, sh, sw, pt, pl,
                                       pad_zero, beta, amax))

        plan.on_backward(build_backward)
        return out


class UpSampling2D(Layer):
    def __init__(self, size=(2, 2), **kwargs):
        super(UpSampling2D, self).__init__(**kwargs)
        self.size = _pair(size)
        if self.size != (2, 2):
            raise NotImplementedError("UpSampling2D size %r" % (self.size,))

    def compute_output_shape(self, input_shape):
        b, h, w, c = input_shape
        return (b, 2 * h, 2 * w, c)

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan)
        if x.needs_grad:
            raise NotImplementedError("UpSampling2D gradient (the reference only up-samples model inputs)")
        b, h, w, c = xbuf.shape
        y = plan.empty(b, 2 * h, 2 * w, c)
        plan.emit(lambda: call("dj_upsample2x", xbuf, rows_of(xbuf)[2], y, c, b, h, w, c))
        return Value(y, needs_grad=False, name=self.name)


class GlobalAveragePooling2D(Layer):
    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[3])

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan)
        assert xbuf.is_contiguous()
        b, h, w, c = xbuf.shape
        y = plan.empty(b, c)
        plan.emit(lambda: call("dj_global_avg_pool_fwd", xbuf, y, b, h * w, c))
        out = Value(y, needs_grad=x.needs_grad, name=self.name)

        def build_backward():
            if out.grad is None or not x.needs_grad:
                return
            dy = out.grad.buf
            dx, beta = plan.grad_of(x)
            plan.emit_bwd(lambda: call("dj_global_avg_pool_bwd", dy, dx, b, h * w, c, beta))

        plan.on_backward(build_backward)
        return out


class Lambda(Layer):
    """Only the identity is supported: the reference defines input-normalisation lambdas but never
    applies them (localisation_part/models/keras_ssd300_dct_j2d_resnet.py:407-435)."""

    def __init__(self, function, output_shape=None, **kwargs):
        super(Lambda, self).__init__(**kwargs)
        self.function = function

    def lower(self, plan, model, ins):
        raise NotImplementedError("Lambda layers have no MI355X lowering")

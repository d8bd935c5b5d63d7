"""keras.models.Model facade over engine.Plan: the methods the reference's entry points call --
compile / fit_generator / predict / train_on_batch / load_weights(by_name) / save_weights /
get_layer(name).output_shape / layers / summary / count_params
(localisation_part/training_dct_pascal_j2d_resnet.py:134-156,244-249,330-336;
classification_part/training.py:159-198; localisation_part/eval_utils/average_precision_evaluator.py:381)."""
import math
import os
import time

import numpy as np
import torch

from .. import engine
from ..engine import GradRef, Plan, Value, call, query
from . import callbacks as cbks
from . import layers as L
from . import optimizers


def _align4(n):
    return (n + 3) // 4 * 4


class Model(object):
    def __init__(self, inputs, outputs, name=None):
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        self.name = name or "model_1"
        # collect the layers reachable from the outputs; creation order is a topological order
        seen, stack = {}, [t.layer for t in self.outputs]
        while stack:
            lyr = stack.pop()
            if id(lyr) in seen:
                continue
            seen[id(lyr)] = lyr
            for t in lyr.inbound:
                stack.append(t.layer)
        for t in self.inputs:
            if id(t.layer) not in seen:
                raise ValueError("Graph disconnected: input %r does not reach the outputs" % (t,))
        self.layers = sorted(seen.values(), key=lambda l: l.serial)
        names = {}
        for lyr in self.layers:
            if lyr.name in names:
                raise ValueError('The name "%s" is used %d times in the model. All layer names should be unique.'
                                 % (lyr.name, 2))
            names[lyr.name] = lyr
        self._by_name = names
        self._consumers = {}
        for lyr in self.layers:
            for t in lyr.inbound:
                self._consumers.setdefault(id(t), []).append(lyr)
        self.optimizer = None
        self.loss = None
        self.metrics = []
        self.stop_training = False
        self._plans = {}
        self._store = None
        self._device = None
        self.dist = None          # set by dist.DataParallel
        self._last_train_plan = None
        self.last_step_info = {}

    # ---- graph queries ------------------------------------------------------------
    def consumers_of(self, ktensor):
        return self._consumers.get(id(ktensor), [])

    def get_layer(self, name=None, index=None):
        if index is not None:
            return self.layers[index]
        if name not in self._by_name:
            raise ValueError("No such layer: " + str(name))
        return self._by_name[name]

    @property
    def weight_specs(self):
        out = []
        for lyr in self.layers:
            out.extend(lyr.weight_specs)
        return out

    def count_params(self):
        return sum(w.size for w in self.weight_specs)

    def summary(self, print_fn=print):
        print_fn("_" * 98)
        print_fn("%-45s %-28s %12s" % ("Layer (type)", "Output Shape", "Param #"))
        print_fn("=" * 98)
        for lyr in self.layers:
            print_fn("%-45s %-28s %12d" % ("%s (%s)" % (lyr.name, lyr.__class__.__name__), str(lyr.output_shape),
                                            lyr.count_params()))
        tr = sum(w.size for w in self.weight_specs if w.trainable)
        tot = self.count_params()
        print_fn("=" * 98)
        print_fn("Total params: {:,}".format(tot))
        print_fn("Trainable params: {:,}".format(tr))
        print_fn("Non-trainable params: {:,}".format(tot - tr))

    # ---- parameters -----------------------------------------------------------------
    def _ensure_params(self, device=None):
        """Allocate every weight in ONE flat HBM buffer (plus one for gradients and one for the SGD
        velocity): [un-regularised trainables | l2-regularised trainables (grouped by l2) | state]."""
        if self._store is not None:
            return
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("no MI355X visible: the compute path has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self._device = device
        specs = self.weight_specs
        plain = [w for w in specs if w.trainable and not w.l2]
        reg = sorted([w for w in specs if w.trainable and w.l2], key=lambda w: w.l2)
        state = [w for w in specs if not w.trainable]
        order = plain + reg + state
        offs, o = {}, 0
        for w in order:
            offs[id(w)] = o
            o += _align4(w.size)
        total = o
        n_train = offs[id(state[0])] if state else total
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grads = torch.zeros(n_train, dtype=torch.float32, device=device)
        vel = torch.zeros(n_train, dtype=torch.float32, device=device)
        host = torch.zeros(total, dtype=torch.float32)
        for w in order:
            a = offs[id(w)]
            val = w.init_value if w.init_value is not None else w.initializer(w.shape)
            host[a:a + w.size] = torch.as_tensor(val, dtype=torch.float32).reshape(-1)
            w.init_value = None
            w.param = flat[a:a + w.size].view(*w.shape)
            if w.trainable:
                w.grad = grads[a:a + w.size].view(*w.shape)
        flat.copy_(host)
        # SGD segments: (begin, end, l2)
        segs = []
        if plain:
            segs.append((0, offs[id(reg[0])] if reg else n_train, 0.0))
        i = 0
        while i < len(reg):
            j = i
            while j < len(reg) and reg[j].l2 == reg[i].l2:
                j += 1
            end = offs[id(reg[j])] if j < len(reg) else n_train
            segs.append((offs[id(reg[i])], end, reg[i].l2))
            i = j
        self._store = dict(flat=flat, grads=grads, vel=vel, n_train=n_train, segments=segs, offsets=offs,
                           sumsq=torch.zeros(len(segs), dtype=torch.float32, device=device))

    def weight_shadows(self, plan):
        """fp16 and bf16 copies of the flat weight buffer (K.set_floatx('float16')): what the reduced-precision GEMMs read as
        their B operand -- fp16 in the forward pass, bf16 in the input gradient -- instead of converting the fp32 master
        weights again in every tile.  `spec.param16` / `spec.parambf` are views like `spec.param`; one dj_shadow_weights
        launch at the head of `plan`'s forward list refreshes them every step, whoever changed the master copy."""
        self._ensure_params()
        st = self._store
        if "flat16" not in st:
            n = st["flat"].numel()
            st["flat16"] = torch.empty(n, dtype=torch.float16, device=self._device)
            st["flatbf"] = torch.empty(n, dtype=torch.bfloat16, device=self._device)
            for w in self.weight_specs:
                a = st["offsets"][id(w)]
                w.param16 = st["flat16"][a:a + w.size].view(*w.shape)
                w.parambf = st["flatbf"][a:a + w.size].view(*w.shape)
            if self._device.type == "cuda":
                call("dj_shadow_weights", st["flat"], st["flat16"], st["flatbf"], n)
        if not getattr(plan, "_shadows_refreshed", False):
            flat, f16, fbf, n = st["flat"], st["flat16"], st["flatbf"], st["flat"].numel()
            plan.fwd.insert(0, lambda: call("dj_shadow_weights", flat, f16, fbf, n))
            plan._shadows_refreshed = True

    @property
    def flat_gradients(self):
        self._ensure_params()
        return self._store["grads"]

    @property
    def flat_trainable(self):
        self._ensure_params()
        return self._store["flat"][: self._store["n_train"]]

    @property
    def flat_all(self):
        self._ensure_params()
        return self._store["flat"]

    def get_weights_dict(self):
        """{'<layer>/<weight>': numpy array} -- Keras-named, as `load_weights(by_name=True)` matches them."""
        self._ensure_params()
        torch.cuda.synchronize()
        return {w.key: w.param.detach().cpu().numpy().copy() for w in self.weight_specs}

    def set_weights_dict(self, d, strict=False):
        self._ensure_params()
        n = 0
        for w in self.weight_specs:
            if w.key in d:
                v = np.asarray(d[w.key], dtype=np.float32)
                if tuple(v.shape) != w.shape:
                    raise ValueError("Layer weight shape %s not compatible with provided weight shape %s (%s)"
                                     % (w.shape, v.shape, w.key))
                w.param.copy_(torch.from_numpy(v).to(w.param.device))
                n += 1
            elif strict:
                raise ValueError("missing weight " + w.key)
        return n

    def get_weights(self):
        d = self.get_weights_dict()
        return [d[w.key] for w in self.weight_specs]

    def set_weights(self, weights):
        specs = self.weight_specs
        if len(weights) != len(specs):
            raise ValueError("You called `set_weights(weights)` with a weight list of length %d, but the model was "
                             "expecting %d weights." % (len(weights), len(specs)))
        self.set_weights_dict({w.key: v for w, v in zip(specs, weights)})

    def save_weights(self, filepath, overwrite=True):
        """Keras writes HDF5; h5py is not available here, so the same name->array mapping is stored as
        an .npz archive under the given path (the path is used verbatim, suffix included)."""
        d = self.get_weights_dict()
        with open(filepath, "wb") as f:
            np.savez(f, **d)

    save = save_weights

    def load_weights(self, filepath, by_name=False):
        with np.load(filepath, allow_pickle=False) as z:
            d = {k: z[k] for k in z.files}
        if by_name:
            return self.set_weights_dict(d)
        return self.set_weights_dict(d, strict=True)

    # ---- compile ----------------------------------------------------------------------
    def compile(self, optimizer, loss=None, metrics=None, **kwargs):
        if isinstance(optimizer, str):
            if optimizer.lower() != "sgd":
                raise NotImplementedError("optimizer %r" % optimizer)
            optimizer = optimizers.SGD()
        if not isinstance(optimizer, optimizers.SGD):
            raise NotImplementedError("only keras.optimizers.SGD is on the reference's hot path")
        self.optimizer = optimizer
        self.metrics = list(metrics or [])
        owner = getattr(loss, "__self__", None)
        if owner is not None and hasattr(owner, "_dj_loss"):
            self.loss = ("ssd", owner)
        elif loss == "categorical_crossentropy" or getattr(loss, "_dj_loss", None) == "categorical_crossentropy":
            self.loss = ("cce", None)
        elif loss is None:
            self.loss = None
        elif callable(loss):
            # Keras loss protocol (localisation_part/training_dct_pascal_j2d_resnet.py:154-156): any
            # loss(y_true, y_pred) -> (batch,) written in torch ops.  It is evaluated on the device-resident tensors and
            # differentiated by torch autograd with respect to y_pred only; the model's own backward pass (the HIP
            # launch list) starts from that gradient.  Checked here, at compile time, on a tiny probe so that a callable
            # that cannot take device tensors fails now and not in the middle of fit_generator.
            self._probe_loss(loss)
            self.loss = ("callable", loss)
        else:
            raise NotImplementedError("loss %r has no MI355X lowering: pass SSDLoss(...).compute_loss, "
                                      "'categorical_crossentropy', or a callable loss(y_true, y_pred) written in torch "
                                      "ops on CUDA tensors" % (loss,))
        self._plans = {}

    def _probe_loss(self, loss):
        if not torch.cuda.is_available():
            return      # graph building without a GPU (structure tests): the probe runs on the first GPU compile
        shape = (2,) + tuple(int(d) for d in self.outputs[0].shape[1:])
        dev = torch.device("cuda", torch.cuda.current_device())
        yp = torch.rand(shape, device=dev).requires_grad_(True)
        yt = torch.rand(shape, device=dev)
        try:
            val = loss(yt, yp)
            if not (isinstance(val, torch.Tensor) and val.requires_grad):
                raise TypeError("it returned %r, not a torch tensor that depends on y_pred" % (type(val).__name__,))
            val.mean().backward()
        except Exception as e:
            raise TypeError("loss %r cannot be compiled: a custom loss must be loss(y_true, y_pred) -> (batch,) in "
                            "differentiable torch ops on CUDA tensors (probe on shape %s failed: %s: %s)"
                            % (loss, shape, type(e).__name__, e))

    # ---- plans ------------------------------------------------------------------------
    def _plan(self, batch_size, training, with_loss, external_grad=False):
        """`external_grad`: training plan without a loss whose backward starts from a caller-filled
        gradient of the first output (`plan.external_grad`) -- used for layer-level parity tests."""
        key = (batch_size, training, with_loss, external_grad)
        if key in self._plans:
            return self._plans[key]
        self._ensure_params()
        plan = Plan(self._device, batch_size, training)
        if training:
            plan.clear_gradients_first(self._store["grads"])
        for lyr in self.layers:
            lyr_ins = [plan.values[id(t)] for t in lyr.inbound]
            if isinstance(lyr, L.InputLayer):
                if not any(lyr is t.layer for t in self.inputs):
                    raise ValueError("Input layer %s is not a model input" % lyr.name)
            if not (hasattr(lyr, "runs_beside") and lyr.runs_beside(plan, self, lyr_ins)):
                plan.wait_side_inputs(lyr_ins)
            out = lyr.lower(plan, self, lyr_ins)
            plan.values[id(lyr.outbound[0])] = out
            # a tensor with several readers may have some that only feed the end of the forward pass (SSD predictor
            # heads): note the point where it is final, so that they can run beside the main chain (Plan.emit_side)
            if (training and isinstance(out, Value) and len(self.consumers_of(lyr.outbound[0])) > 1
                    and getattr(out, "pending_add", None) is None and getattr(out, "side_done", None) is None):
                plan.mark_ready(out)
        # model inputs in the order given to Model(...)
        order = []
        for t in self.inputs:
            order.append(plan.values[id(t)].buf)
        plan.inputs = order
        plan.outputs = [plan.values[id(t)] for t in self.outputs]
        if with_loss:
            self._lower_loss(plan)
        if external_grad:
            def seed_gradient():
                plan.external_grad, _ = plan.grad_of(plan.outputs[0])
            plan.on_backward(seed_gradient)
        if training:
            plan.build_backward()
        if os.environ.get("DJ_AUTOTUNE", "1") != "0":
            # timing an untuned geometry runs its launch, and a conv that finalizes its BatchNormalization advances
            # that layer's moving statistics: put the non-trainable state back afterwards
            state = self.flat_all[self._store["n_train"]:]
            saved = state.clone()
            if plan.autotune(reps=int(os.environ.get("DJ_AUTOTUNE_REPS", "2")),
                             verbose=os.environ.get("DJ_AUTOTUNE_VERBOSE", "0") == "1",
                             measure=os.environ.get("DJ_AUTOTUNE", "1") != "table"):
                state.copy_(saved)
            plan.finalize_workspaces()
            if os.environ.get("DJ_TUNE_SAVE"):
                from ..engine import save_tune_db
                save_tune_db(os.environ["DJ_TUNE_SAVE"])
        if training:
            if self.dist is not None and with_loss:
                self.dist.attach(plan)
        self._plans[key] = plan
        return plan

    def _lower_loss(self, plan):
        if self.loss is None:
            raise RuntimeError("You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.")
        kind, obj = self.loss
        pred = plan.outputs[0]
        yp = pred.buf
        assert not pred.is_affine and yp.is_contiguous()
        plan.y_true = plan.empty(*yp.shape)
        plan.loss_out = plan.zeros(8)
        yt = plan.y_true
        if kind == "ssd":
            n_cls = yp.shape[-1] - 12
            nbox = yp.numel() // yp.shape[-1]
            ws = plan.empty(query("dj_ssd_loss_workspace_floats", nbox))
            out5 = plan.loss_out
            ratio, nmin, alpha = int(obj.neg_pos_ratio), int(obj.n_neg_min), float(obj.alpha)

            def loss_fwd():
                if plan.targets_event is not None:   # y_true is being encoded on the side stream (see _upload)
                    torch.cuda.current_stream().wait_event(plan.targets_event)
                    plan.targets_event = None
                call("dj_ssd_loss_fwd", yt, yp, nbox, n_cls, ratio, nmin, alpha, ws, out5)
            plan.emit(loss_fwd)

            def build_backward():
                d, beta = plan.grad_of(pred)
                assert beta == 0
                plan.emit_bwd(lambda: call("dj_ssd_loss_bwd", yt, yp, nbox, n_cls, alpha, 1.0, ws, out5, d))

            plan.on_backward(build_backward)
        elif kind == "callable":
            out = plan.loss_out
            dpred = plan.empty(*yp.shape) if plan.training else None

            def loss_fwd():
                if plan.targets_event is not None:   # y_true is being encoded on the side stream (see _upload)
                    torch.cuda.current_stream().wait_event(plan.targets_event)
                    plan.targets_event = None
                # Keras: total loss = mean over the batch of loss(y_true, y_pred); torch autograd supplies d/d y_pred
                leaf = yp.detach().requires_grad_(plan.training)
                with torch.enable_grad():
                    val = obj(yt, leaf).mean()
                    if plan.training:
                        dpred.copy_(torch.autograd.grad(val, leaf)[0])
                out[0:1].copy_(val.detach().reshape(1))
            plan.emit(loss_fwd)

            def build_backward():
                plan.set_grad_ref(pred, GradRef(dpred))

            plan.on_backward(build_backward)
        else:
            rows, c = yp.shape[0], yp.shape[1]
            loss_rows = plan.empty(rows)
            dprobs = plan.empty(rows, c) if plan.training else None
            out = plan.loss_out
            plan.emit(lambda: call("dj_categorical_crossentropy", yt, yp, rows, c, 1.0, loss_rows, dprobs, out))

            def build_backward():
                plan.set_grad_ref(pred, GradRef(dprobs))

            plan.on_backward(build_backward)

    # ---- steps ----------------------------------------------------------------------
    @staticmethod
    def _as_list(x):
        return list(x) if isinstance(x, (list, tuple)) else [x]

    def _upload(self, plan, x, y):
        xs = self._as_list(x)
        if len(xs) != len(plan.inputs):
            raise ValueError("Error when checking model input: expected %d arrays but got %d" % (len(plan.inputs), len(xs)))
        for buf, arr in zip(plan.inputs, xs):
            t = arr if isinstance(arr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(arr))
            if tuple(t.shape) != tuple(buf.shape):
                raise ValueError("Error when checking input: expected shape %s but got array with shape %s"
                                 % (tuple(buf.shape), tuple(t.shape)))
            buf.copy_(t.to(torch.float32), non_blocking=True)
        if y is not None and hasattr(y, "encode_into"):      # PendingTargets: SSDInputEncoder runs on the device
            if tuple(y.shape) != tuple(plan.y_true.shape):
                raise ValueError("Error when checking target: expected shape %s but got array with shape %s"
                                 % (tuple(plan.y_true.shape), tuple(y.shape)))
            side = plan.side_stream
            if side is None:
                y.encode_into(plan.y_true)
            else:
                # the matching kernel runs beside the forward pass: it may start once the previous step (whose loss
                # still reads y_true) has drained, and the loss of this step waits for it
                drained = torch.cuda.Event()
                drained.record()
                side.wait_event(drained)
                with torch.cuda.stream(side):
                    y.encode_into(plan.y_true)
                    plan.targets_event = torch.cuda.Event()
                    plan.targets_event.record(side)
        elif y is not None:
            t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y))
            if tuple(t.shape) != tuple(plan.y_true.shape):
                raise ValueError("Error when checking target: expected shape %s but got array with shape %s"
                                 % (tuple(plan.y_true.shape), tuple(t.shape)))
            plan.y_true.copy_(t.to(torch.float32), non_blocking=True)

    def _apply_optimizer(self, plan=None):
        """`plan`: the training plan whose backward pass just ran (data parallel: it must carry the gradient exchange;
        None = the plan of the last run_train_step)."""
        st, opt = self._store, self.optimizer
        lr_t = opt.current_lr()
        scale = 1.0
        if self.dist is not None:
            scale = self.dist.finish_gradients(plan if plan is not None else self._last_train_plan)
        st["sumsq"].zero_()
        for i, (a, b, l2) in enumerate(st["segments"]):
            if b <= a:
                continue
            ssq = st["sumsq"][i:i + 1] if l2 else None
            call("dj_sgd_momentum_update", st["flat"][a:b], st["grads"][a:b], st["vel"][a:b], b - a, lr_t,
                 opt.momentum, int(opt.nesterov), l2, scale, ssq)
        opt.iterations += 1

    def run_train_step(self, plan):
        """Forward, backward and optimizer step on buffers already resident in HBM (no host sync)."""
        self._last_train_plan = plan
        plan.run_forward()
        plan.run_backward()
        self._apply_optimizer(plan)

    def _loss_value(self, plan, with_reg=True):
        vals = plan.loss_out.detach().cpu()
        data = float(vals[0])
        reg = 0.0
        if with_reg:
            ss = self._store["sumsq"].detach().cpu()
            for i, (_, _, l2) in enumerate(self._store["segments"]):
                reg += l2 * float(ss[i])
        self.last_step_info = dict(data_loss=data, reg_loss=reg, n_positive=float(vals[1]), n_negative=float(vals[2]))
        return data + reg

    def _metric_values(self, plan):
        """Compile-time `metrics` callables, evaluated as metric(y_true, y_pred) on the device-resident batch;
        named like Keras names anonymous metric functions (`_func`, `_func_1`, ...)."""
        out, seen = {}, {}
        for m in self.metrics:
            name = getattr(m, "__name__", str(m)) if callable(m) else str(m)
            if not callable(m):
                if name in ("accuracy", "acc"):
                    from .metrics import categorical_accuracy as m
                    name = "acc"
                else:
                    continue
            n = seen.get(name, 0)
            seen[name] = n + 1
            out[name if n == 0 else "%s_%d" % (name, n)] = float(m(plan.y_true, plan.outputs[0].buf))
        return out

    def train_on_batch(self, x, y):
        b = self._as_list(x)[0].shape[0]
        plan = self._plan(b, True, True)
        self._upload(plan, x, y)
        self.run_train_step(plan)
        loss = self._loss_value(plan)
        if self.metrics:
            self.last_step_info["metrics"] = self._metric_values(plan)
        return loss

    def test_on_batch(self, x, y):
        b = self._as_list(x)[0].shape[0]
        plan = self._plan(b, False, True)
        self._upload(plan, x, y)
        plan.run_forward()
        return self._loss_value(plan, with_reg=False) + self._reg_penalty()

    def _reg_penalty(self):
        tot = 0.0
        for (a, b, l2) in self._store["segments"]:
            if l2 and b > a:
                tot += l2 * float((self._store["flat"][a:b] ** 2).sum())
        return tot

    def predict_on_batch(self, x):
        b = self._as_list(x)[0].shape[0]
        plan = self._plan(b, False, False)
        self._upload(plan, x, None)
        plan.run_forward()
        outs = [v.buf.detach().cpu().numpy() for v in plan.outputs]
        return outs[0] if len(outs) == 1 else outs

    def predict(self, x, batch_size=32, verbose=0):
        xs = self._as_list(x)
        n = xs[0].shape[0]
        chunks = []
        for i in range(0, n, batch_size):
            chunks.append(self.predict_on_batch([a[i:i + batch_size] for a in xs]))
        if isinstance(chunks[0], list):
            return [np.concatenate([c[j] for c in chunks], axis=0) for j in range(len(chunks[0]))]
        return np.concatenate(chunks, axis=0)

    # ---- training loop ----------------------------------------------------------------
    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None,
                      validation_data=None, validation_steps=None, class_weight=None, max_queue_size=10, workers=1,
                      use_multiprocessing=False, shuffle=True, initial_epoch=0):
        """Keras `fit_generator` contract: `generator` yields (inputs, targets) batches forever;
        one epoch = `steps_per_epoch` train_on_batch calls, then an optional validation sweep."""
        if steps_per_epoch is None:
            steps_per_epoch = len(generator)
        history = cbks.History()
        cb_list = [history] + list(callbacks or [])
        for cb in cb_list:
            cb.set_model(self)
            cb.params = {"steps": steps_per_epoch, "epochs": epochs}
            cb.on_train_begin()
        it = iter(generator)
        val_it = iter(validation_data) if validation_data is not None and not isinstance(validation_data, tuple) else None
        if val_it is not None and validation_steps is None:
            if hasattr(validation_data, "__len__"):     # keras.utils.Sequence
                validation_steps = len(validation_data)
            else:
                raise ValueError("`validation_steps=None` is only valid for a generator based on the `keras.utils.Sequence`"
                                 " class. Please specify `validation_steps` or use the `keras.utils.Sequence` class.")
        self.stop_training = False
        rank0 = self.dist is None or self.dist.rank == 0
        # Keras runs the generator in a background enqueuer (`workers`, `max_queue_size`); so does this loop, with one
        # thread (a plain generator is not thread-safe): batch production overlaps the GPU step.  (Reading a step's loss
        # one step late, to spare the per-step host sync, was measured at 1057 vs 1053 img/s and is not done: callbacks
        # see every loss before the next batch starts, as in Keras.)
        feeder = _Prefetcher(it, max_queue_size) if workers and workers > 0 else None
        for epoch in range(initial_epoch, epochs):
            for cb in cb_list:
                cb.on_epoch_begin(epoch)
            t0, run, run_metrics = time.time(), 0.0, {}
            for step in range(steps_per_epoch):
                for cb in cb_list:
                    cb.on_batch_begin(step)
                batch = feeder.get() if feeder is not None else next(it)
                x, y = batch[0], batch[1]
                loss = self.train_on_batch(x, y)
                run += loss
                logs = {"loss": loss, "batch": step, "size": self._as_list(x)[0].shape[0]}
                for mk, mv in self.last_step_info.get("metrics", {}).items():
                    logs[mk] = mv
                    run_metrics[mk] = run_metrics.get(mk, 0.0) + mv
                for cb in cb_list:
                    cb.on_batch_end(step, logs)
                if self.stop_training:
                    break
            logs = {"loss": run / max(1, step + 1), "lr": self.optimizer.current_lr()}
            for mk, mv in run_metrics.items():
                logs[mk] = mv / max(1, step + 1)
            if validation_data is not None:
                vl, nv = 0.0, 0
                if val_it is not None:
                    for _ in range(int(validation_steps)):
                        vb = next(val_it)
                        vl += self.test_on_batch(vb[0], vb[1])
                        nv += 1
                else:
                    vl, nv = self.test_on_batch(validation_data[0], validation_data[1]), 1
                logs["val_loss"] = vl / max(1, nv)
            if self.dist is not None:
                logs = self.dist.average_metrics(logs)
            if verbose and rank0:
                print("Epoch %d/%d - %.0fs - %s" % (epoch + 1, epochs, time.time() - t0,
                                                    " - ".join("%s: %.4f" % kv for kv in sorted(logs.items()))))
            for cb in cb_list:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        if feeder is not None:
            feeder.close()
        for cb in cb_list:
            cb.on_train_end()
        return history


class _Prefetcher(object):
    """One background thread draining a (non thread-safe) generator into a bounded queue, like Keras'
    GeneratorEnqueuer with workers=1."""

    def __init__(self, iterator, max_queue_size=10):
        import queue
        import threading
        self.q = queue.Queue(maxsize=max(1, int(max_queue_size)))
        self.stop = threading.Event()

        def work():
            try:
                while not self.stop.is_set():
                    item = next(iterator)
                    while not self.stop.is_set():
                        try:
                            self.q.put(("item", item), timeout=0.1)
                            break
                        except queue.Full:
                            pass
            except StopIteration:
                self.q.put(("end", None))
            except BaseException as e:   # surfaces in the training thread
                self.q.put(("error", e))

        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def get(self):
        kind, item = self.q.get()
        if kind == "error":
            raise item
        if kind == "end":
            raise StopIteration
        return item

    def close(self):
        """Stop the thread and wait for it (a generator must not be resumed by two threads); batches it had already
        produced are dropped, as Keras' enqueuer does on stop()."""
        import queue
        self.stop.set()
        try:                      # a producer blocked on the full queue would only notice `stop` after its put timeout
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join()


def load_model(filepath, custom_objects=None, compile=True):
    raise NotImplementedError(
        "load_model needs the HDF5 architecture blob Keras writes; rebuild the model with its builder and call "
        "load_weights(path, by_name=True) (what the reference does right after load_model, "
        "localisation_part/training_dct_pascal_j2d_resnet.py:137-149)")

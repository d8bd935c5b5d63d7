"""Static execution plans for Keras-style graphs on one MI355X.

A `Plan` is what `Model.compile()/fit_generator()/predict()` run: the layer graph is lowered ONCE
(per batch size and mode) into a fixed list of C-ABI launches over preallocated HBM buffers --
forward list, backward list -- so a training step is launch-only (no allocation, no host sync, no
Python autograd) and can be captured into a hipGraph.  This takes the place of the Keras 2.2.4
`fit_generator -> train_on_batch -> K.function -> tf.Session.run` stack the reference relies on
(localisation_part/training_dct_pascal_j2d_resnet.py:330-336).

Tensors inside a plan are `Value`s.  A Value is either materialised (`buf`) or a *virtual*
per-channel affine(+ReLU) of a raw buffer: BatchNormalization and Activation('relu') do not launch
anything in the forward pass -- the consumer conv applies `relu(z*scale+shift)` while staging its A
tile (kernel prologue), and Add+ReLU applies both branches' affines in its single pass.

Backward lists are built in a second pass, in exactly the order they will execute; `Plan.grad_of`
hands out the gradient buffer of a Value and tells the caller whether it is the first writer
(store) or a later one (accumulate, `beta=1` epilogues), so fan-out needs no extra add kernels.
"""
import os

import torch

from . import _lib
from ._lib import check

_SIDE = {}


def _side_stream(device):
    """One extra HIP stream per device, shared by all plans of the process."""
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class _DrySideStream(object):
    """Stands in for the side stream of a plan lowered on the host: nothing can be enqueued on it."""

    def __getattr__(self, name):
        raise RuntimeError("this plan was lowered on the CPU for structure checks only: it cannot run")


def _conv(a):
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


def call(name, *args):
    """Launch one C-ABI entry point on torch's current HIP stream (tensors -> device pointers)."""
    fn = getattr(_lib.load(), name)
    rc = fn(*[_conv(a) for a in args], _lib.current_stream())
    check(rc, name)


def launcher(name, *args):
    """-> a closure that launches C-ABI entry point `name` with `args` on the current launch stream; tensors are converted
    to device pointers once, here (plan buffers never move), and kept alive by the closure."""
    fn = getattr(_lib.load(), name)
    conv = [_conv(a) for a in args]
    keep = [a for a in args if isinstance(a, torch.Tensor)]

    def run():
        check(fn(*conv, _lib.current_stream()), name)
    run._keep = keep
    return run


def copy2d_multi(parts):
    """parts: list of (src, ld_src, dst, ld_dst, rows, cols, beta) -> a launcher issuing as few dj_copy2d_multi
    launches as possible (descriptor arrays are built once, here)."""
    lib = _lib.load()
    chunks = []
    for i in range(0, len(parts), 8):
        chunk = parts[i:i + 8]
        arr = (_lib.CopyPart * len(chunk))()
        for k, (src, lds, dst, ldd, rows, cols, beta) in enumerate(chunk):
            arr[k] = _lib.CopyPart(src.data_ptr(), dst.data_ptr(), int(lds), int(ldd), int(rows), int(cols), int(beta))
        chunks.append((arr, len(chunk)))
    keep = [(p[0], p[2]) for p in parts]   # the tensors stay alive as long as the launcher does

    def run():
        stream = _lib.current_stream()
        for arr, n in chunks:
            check(lib.dj_copy2d_multi(arr, n, stream), "dj_copy2d_multi")
    run._keep = keep
    return run


def colsum_multi(parts):
    """parts: list of (x, rows, c, ld, out) -> a launcher issuing dj_colsum_multi (32 tensors per launch)."""
    lib = _lib.load()
    chunks = []
    for i in range(0, len(parts), 32):
        chunk = parts[i:i + 32]
        arr = (_lib.ColsumPart * len(chunk))()
        for k, (x, rows, c, ld, out) in enumerate(chunk):
            arr[k] = _lib.ColsumPart(x.data_ptr(), out.data_ptr(), int(rows), int(c), int(ld), 0)
        chunks.append((arr, len(chunk)))
    keep = [(p[0], p[4]) for p in parts]

    def run():
        stream = _lib.current_stream()
        for arr, n in chunks:
            check(lib.dj_colsum_multi(arr, n, stream), "dj_colsum_multi")
    run._keep = keep
    return run


def _db_lookup(key):
    """Table entry of a tuner key (direction, geometry...).  Direction 9 -- the input gradient that also takes the
    BatchNormalization backward statistics, never split -- falls back to the plain input gradient's tile variant."""
    db = _tune_db()
    known = db.get(",".join(str(int(v)) for v in key))
    if known is None and key[0] == 9:
        plain = db.get(",".join(str(int(v)) for v in (1,) + tuple(key[1:])))
        if plain is not None:
            known = [plain[0], 1, plain[2]]
    return known


def tuned_splits(direction, desc):
    """Split-K factor the in-tree table registers for this geometry (None: not in the table)."""
    names = [n for n, _ in _lib.ConvDesc._fields_][:15]
    known = _db_lookup((direction,) + tuple(getattr(desc, n) for n in names))
    return None if known is None else int(known[1])


def query(name, *args):
    return check(getattr(_lib.load(), name)(*[int(a) for a in args]), name)


def query_long(name, desc, *args):
    return check(getattr(_lib.load(), name)(desc, *[int(a) for a in args]), name)


class _TunedByMode(dict):
    """conv geometry key -> (ms, cfg, splits) of the choices registered with the library, one table per arithmetic mode
    (the library keys its overrides by mode too): `_TUNED` behaves like the dict of the calling thread's current mode."""

    def _cur(self):
        return dict.setdefault(self, int(_lib.load().dj_get_compute_mode()), {})

    def __contains__(self, key):
        return key in self._cur()

    def __getitem__(self, key):
        return self._cur()[key]

    def __setitem__(self, key, value):
        self._cur()[key] = value

    def __iter__(self):
        return iter(self._cur())

    def __len__(self):
        return len(self._cur())

    def items(self):
        return self._cur().items()

    def clear(self):
        self._cur().clear()


_TUNED = _TunedByMode()
_TUNE_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_conv.json")
_TUNE_DB_LOWP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_conv_f16.json")
_TUNE_DB_X3 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_conv_x3.json")
_TUNE_DB_MFMA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_conv_mfma.json")
_TUNE_DB_X6 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_conv_x6.json")
_DB = {}      # compute mode -> {"dir,geometry...": [cfg, splits, ms]}: tile choices measured on an MI355X, shipped in-tree


def _tune_db():
    """The in-tree table of measured tile choices (like a find-db: geometry -> kernel variant): [cfg, splits, ms alone]
    per geometry, plus a trailing 1 where tools/tune_in_step.py replaced the fastest-alone choice by the one that makes
    the whole training step fastest.  The reduced-precision modes (K.set_floatx('float16' / 'bfloat16')) have a table
    of their own: with an 8-deep MFMA the balance between staging and arithmetic, and with it the best tile, differs
    (+5.8 % on the fp16 deconv SSD300 step, +3.1 % on ssd_custom); 'float32x3' has a third (two LDS images per operand:
    the large tiles lose a block per CU).  Geometries a table does not hold are timed at plan
    time; DJ_TUNE_DB=path selects another file, DJ_TUNE_DB=0 ignores the tables."""
    mode = int(_lib.load().dj_get_compute_mode())
    if mode not in _DB:
        _DB[mode] = {}
        default = _TUNE_DB
        # float32x3 / float32x6: tables of their own, else the 16-bit-tile one
        # (mode 5 = fp32 MFMA kernels only: the table measured over those; mode 0 chooses among both fp32 kernel families)
        for m, path in ((5, _TUNE_DB_MFMA), (3, _TUNE_DB_X3), (4, _TUNE_DB_X6), (4, _TUNE_DB_X3), (mode, _TUNE_DB_LOWP)):
            if mode == m and mode != 0 and os.path.exists(path):
                default = path
                break
        path = os.environ.get("DJ_TUNE_DB", default)
        if path != "0" and os.path.exists(path):
            import json
            with open(path) as f:
                table = json.load(f)
            _DB[mode] = table.get("entries", {})
            if mode == 4 and path == _TUNE_DB_X3 and os.path.exists(_TUNE_DB):
                # 'float32x6' everywhere: no table of its own ships -- the default mode's table holds the measured x6
                # choice wherever an x6 kernel won (upper half of its index range); the float32x3 table covers the rest
                with open(_TUNE_DB) as f:
                    both = json.load(f)
                half = int(both.get("n_configs", 0)) // 2
                for key, v in both.get("entries", {}).items():
                    if half and v[0] >= half:
                        _DB[mode][key] = [v[0] - half] + list(v[1:])
    return _DB[mode]


def reset_tuning():
    """Forget the choices registered so far for the current arithmetic mode (tests; a mode switch does not need it: the
    library and this module keep one table per mode)."""
    lib = _lib.load()
    names = [n for n, _ in _lib.ConvDesc._fields_][:15]
    for key in list(_TUNED):
        desc = _lib.ConvDesc()
        for n, v in zip(names, key[1:]):
            setattr(desc, n, int(v))
        desc.ld_x, desc.ld_y = desc.in_c, desc.out_c
        lib.dj_conv2d_tune_set(int(key[0]), desc, -1, 1)
    _TUNED.clear()


def save_tune_db(path=None):
    """Write every choice made in this process (measured or loaded) to `path`."""
    import json
    entries = dict(_tune_db())
    for key, (ms, cfg, sp) in _TUNED.items():
        name = ",".join(str(int(v)) for v in key)
        if len(entries.get(name, ())) > 3:
            continue     # an in-step choice (4th element): keep it as it is
        entries[name] = [int(cfg), int(sp), round(float(ms), 5)]
    path = path or _TUNE_DB
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"arch": "gfx950", "abi": int(_lib.load().dj_abi_version()), "n_configs":
                   int(_lib.load().dj_conv2d_tune_configs()), "entries": entries}, f, indent=0, sort_keys=True)
    return len(entries)


class GradRef(object):
    """Gradient of a Value: a buffer, optionally to be masked by `mask_y > 0` by whoever reads it
    (lets Add+ReLU hand its upstream gradient to both branches without materialising the mask)."""

    def __init__(self, buf, mask_y=None, also=None):
        self.buf = buf
        self.mask_y = mask_y
        # (buffer, beta): whoever applies the mask must also store (beta 0) / accumulate (beta 1) the masked gradient
        # there -- the identity shortcut of a residual block shares it with the BatchNormalization branch
        self.also = also
        # (partial rows, count): the BatchNormalization backward statistics of this gradient, taken by the GEMM that wrote it
        self.bwd_partial = None


class Value(object):
    def __init__(self, buf, scale=None, shift=None, relu=False, needs_grad=False, name=""):
        self.buf = buf
        self.scale = scale
        self.shift = shift
        self.relu = relu
        self.needs_grad = needs_grad
        self.name = name
        self.grad = None          # GradRef, set during the backward-building pass
        self.alias_of = None      # gradient requests are redirected (ZeroPadding2D, Reshape)
        self.alias_view = None    # callable: grad buffer of alias_of -> view shaped like self
        self.pad = None           # pending ZeroPadding2D ((t,b),(l,r)) consumed by the next Conv2D
        self.conv_stats = None    # (partial, nrows, conv_bias) when the producing conv took BN statistics
        self.relu_child = None    # Value created by Activation('relu') on a virtual affine
        self.bn = None            # BatchNormalization record that produced this virtual affine

    @property
    def is_affine(self):
        return self.scale is not None

    @property
    def shape(self):
        return tuple(self.buf.shape)


class _Workspace(object):
    """One float buffer per (plan, stream), grown to the largest split-K slab set its convolutions need (capped: beyond
    the cap a launch falls back to fp32 atomics, see include/dj_hip.h)."""
    CAP_FLOATS = 64 << 20

    def __init__(self, device):
        self.device = device
        self.users = []
        # DJ_FWD_SLABS=0: no workspace -- split-K forward launches accumulate with fp32 atomics again (A/B runs)
        self.off = os.environ.get("DJ_FWD_SLABS", "1") == "0"
        self.buf = None if self.off else torch.empty(4, dtype=torch.float32, device=device)

    def register(self, desc, stats_may_split):
        if self.off:
            return
        self.users.append((desc, bool(stats_may_split)))
        self._fit(desc, stats_may_split)

    def _fit(self, desc, may):
        need = min(int(query_long("dj_conv2d_fwd_workspace_floats", desc, int(may))), self.CAP_FLOATS)
        if need > self.buf.numel():
            self.buf = torch.empty(need, dtype=torch.float32, device=self.device)

    def fit_splits(self, splits):
        """Room for every registered geometry at split-K factor `splits` (the tuner's trial launches must run the slab
        path they are being timed for, not the atomics fallback of a workspace that is too small); geometries whose
        slabs would exceed the cap are left to `splits_fit`."""
        if self.off:
            return
        need = 4
        for desc, _ in self.users:
            n = int(splits) * desc.batch * desc.out_h * desc.out_w * desc.out_c
            if n <= self.CAP_FLOATS:
                need = max(need, n)
        if need > self.buf.numel():
            self.buf = torch.empty(need, dtype=torch.float32, device=self.device)

    def resize(self):
        """Size for the CURRENT tuning choice of every registered geometry (shrinks after a tuning pass)."""
        if self.off:
            return
        need = 4
        for desc, may in self.users:
            need = max(need, min(int(query_long("dj_conv2d_fwd_workspace_floats", desc, int(may))), self.CAP_FLOATS))
        if need != self.buf.numel():
            self.buf = torch.empty(need, dtype=torch.float32, device=self.device)


class Plan(object):
    def __init__(self, device, batch_size, training):
        self.device = device
        self.batch_size = batch_size
        self.training = training
        self.fwd = []
        self.bwd = []
        self._bwd_builders = []
        self.values = {}       # KTensor id -> Value
        self.inputs = []       # device buffers the host copies each batch into
        self.outputs = []
        self.y_true = None
        self.loss_out = None   # device float[5]
        self.bytes_allocated = 0
        self.hooks_after_backward = []
        self.grad_ready = {}   # weight key -> index in self.bwd after which its gradient is final
        self.deferred_colsums = []   # (dy, rows, c, ld, grad buffer, weight spec): see build_backward
        self.conv_calls = []   # (direction, ConvDesc, launch closure) of every implicit-GEMM call, for autotune()
        self.fused_outputs = {}   # id(Conv2D layer) -> Value, for layers lowered inside a sibling's fused GEMM
        # the arithmetic mode this plan was lowered under (K.set_floatx at that time): its launches run in it whatever the
        # process default or another plan's mode is by then (per-thread override of the C library)
        self.compute_mode = int(_lib.load().dj_get_compute_mode()) if device.type == "cuda" else 0
        # BASELINE config 5 proper (K.set_floatx('float16')): conv outputs and block sums of the backbone are held as fp16
        # in HBM, their gradients as bf16 (the layers decide per tensor, keras/layers.py `_takes16`); DJ_STORE16=0 keeps
        # every tensor fp32 as in round 2.  Tensors with fewer rows than DJ_STORE16_MIN_ROWS stay fp32: they carry no
        # traffic worth halving, and their GEMMs are the ones that want to split their reduction over workgroups, which a
        # 16-bit result cannot take (no atomics, no slabs)
        self.store16 = self.compute_mode == 1 and os.environ.get("DJ_STORE16", "1") != "0"
        self.store16_min_rows = int(os.environ.get("DJ_STORE16_MIN_ROWS", "8192"))
        self.grads_cleared = False   # True: the first backward launch zeroes the model's whole flat gradient buffer
        # weight-gradient GEMMs only feed the optimizer, so they run on a second HIP stream and fill the CUs the
        # data-gradient chain leaves idle at its tile-quantisation tails (DJ_SIDE_WGRAD=0 keeps one stream)
        self._arena, self._arena_used, self._arena_high = [], 0, {}
        self.targets_event = None    # set while the side stream encodes y_true (Model._upload), cleared by the loss
        self.side_stream = None
        self.side_enabled = True     # cleared while kernels are timed one by one (bench.py)
        self._side_dirty = False
        self._workspaces = {}        # side? -> _Workspace (split-K slabs of the forward convolutions)
        self._side_joined = {}       # completion events of side-stream forward work the main stream already waits for
        if training and device.type == "cuda" and os.environ.get("DJ_SIDE_WGRAD", "1") != "0":
            self.side_stream = _side_stream(device)
            self._join_event = torch.cuda.Event()
        elif training and device.type != "cuda" and os.environ.get("DJ_SIDE_WGRAD", "1") != "0":
            # structure-only lowering on the host (tests/test_dist_gloo.py walks launch lists, gradient-ready indices and
            # buckets without a GPU): the same lists as on the card -- side-stream heads, fused launches -- whose closures
            # are never run; Model._ensure_params only accepts a CPU device when it is asked for explicitly
            self.side_stream = _DrySideStream()
            self._join_event = None

    # ---- allocation -------------------------------------------------------------
    def empty(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        self.bytes_allocated += t.numel() * t.element_size()
        return t

    def act_dtype(self, rows, channels):
        """Storage type of a backbone activation with this many rows / channels (the caller has checked that every layer
        that touches it can work on 16-bit tensors)."""
        if self.store16 and rows >= self.store16_min_rows and channels % 32 == 0:
            return torch.float16
        return torch.float32

    @staticmethod
    def grad_dtype(buf):
        """Gradients of 16-bit activations are held as bf16 (they need the exponent range), everything else as fp32."""
        return torch.float32 if buf.dtype == torch.float32 else torch.bfloat16

    def zeroed_each_step(self, *shape, dtype=torch.float32):
        """A buffer out of an arena that ONE memset clears at the start of every step: for gradient tensors whose first
        writer accumulates with atomics (split-K) or scatters (stride-2 1x1 dgrad) and would otherwise need its own
        memset launch in the middle of the backward chain."""
        n_el = 1
        for d in shape:
            n_el *= int(d)
        es = torch.empty(0, dtype=dtype).element_size()
        n = (n_el * es + 3) // 4          # floats of the arena that hold the tensor
        n_pad = (n + 63) // 64 * 64
        chunk = 64 << 20     # floats per arena chunk (256 MB)
        if not self._arena or self._arena_used + n_pad > self._arena[-1].numel():
            self._arena.append(torch.zeros(max(chunk, n_pad), dtype=torch.float32, device=self.device))
            self._arena_used = 0
            self.bytes_allocated += self._arena[-1].numel() * 4
            if len(self._arena) == 1:
                self.fwd.insert(0, self._clear_arena)
        t = self._arena[-1][self._arena_used:self._arena_used + n]
        t = t.view(*shape) if dtype == torch.float32 else t.view(dtype)[:n_el].view(*shape)
        self._arena_used += n_pad
        self._arena_high[len(self._arena) - 1] = self._arena_used
        return t

    def _clear_arena(self):
        for i, a in enumerate(self._arena):
            a[:self._arena_high[i]].zero_()

    def zeros(self, *shape):
        t = torch.zeros(*shape, dtype=torch.float32, device=self.device)
        self.bytes_allocated += t.numel() * 4
        return t

    # ---- split-K workspaces of the forward convolutions ---------------------------------------------------------------
    def conv_workspace(self, desc, side=False, stats_may_split=False):
        """Holder of the workspace through which the forward convolutions issued on one stream (main / side) run their
        split-K launches WITHOUT atomics (dj_conv2d_nhwc_fwd_ws: slabs + fixed-order reduction, bit-reproducible).
        Sized for the current tuning choice of every registered geometry; `finalize_workspaces` sizes it again once the
        tuner has had its say.  Launches on one stream are serial, so one buffer per stream serves them all."""
        ws = self._workspaces.setdefault(bool(side), _Workspace(self.device))
        ws.register(desc, stats_may_split)
        return ws

    def finalize_workspaces(self):
        for ws in self._workspaces.values():
            ws.resize()

    # ---- recording -------------------------------------------------------------
    def emit(self, fn):
        self.fwd.append(fn)

    def emit_bwd(self, fn):
        self.bwd.append(fn)

    # ---- forward work beside the main chain ------------------------------------------------------------------------
    # The predictor heads of an SSD source (pack copy, one GEMM, unpack copies) and the L2Normalization in front of the
    # first one only feed the prediction assembly at the very end of the forward pass, while the main stream goes on
    # with the next feature layers -- small, latency-bound launches that leave most CUs idle.  Such work is issued on
    # the side stream behind an event recorded when its input became final (`mark_ready`), and whoever consumes its
    # results on the main stream waits for `side_done` first (`wait_side_inputs`, called for every layer).
    def forward_side_ok(self, v):
        return (self.side_stream is not None and getattr(v, "ready_event", None) is not None
                and os.environ.get("DJ_SIDE_HEADS", "1") != "0")

    def mark_ready(self, v):
        """Record, at this point of the forward list, an event that says `v.buf` is final."""
        if self.side_stream is None or getattr(v, "ready_event", None) is not None:
            return
        ev = torch.cuda.Event()
        v.ready_event = ev
        self.fwd.append(lambda: ev.record())

    def emit_side(self, fn, after):
        """Forward launch(es) `fn` on the side stream once event `after` has passed."""
        side = self.side_stream

        def run():
            if not self.side_enabled:
                return fn()
            side.wait_event(after)
            self._launch_on_side(fn)
        self.fwd.append(run)

    def side_results(self, values):
        """The Values in `values` were written by emit_side launches issued so far: record their completion."""
        done = torch.cuda.Event()
        side = self.side_stream

        def rec():
            if self.side_enabled:
                done.record(side)
        self.fwd.append(rec)
        for v in values:
            v.side_done = done

    def wait_side_inputs(self, values):
        """Main stream: wait for side-stream producers of `values` (once per producer event)."""
        for v in values:
            done = getattr(v, "side_done", None)
            if done is None or id(done) in self._side_joined:
                continue
            self._side_joined[id(done)] = done

            def wait(done=done):
                if self.side_enabled:
                    torch.cuda.current_stream().wait_event(done)
            self.fwd.append(wait)

    def emit_conv(self, direction, desc, fn, backward=False, side=False, fwd_after=None):
        """Record one implicit-GEMM launch (direction 0 fwd / 1 dgrad / 2 wgrad, +4 = forward with BN statistics, 9 =
        input gradient that takes BatchNormalization backward statistics: the tuner keys of include/dj_hip.h).
        side=True: the launch has no consumer before the optimizer / gradient exchange and every buffer it reads is
        final when it is issued, so it may run on the plan's side stream."""
        self.conv_calls.append((direction, desc, fn))
        if side and backward and self.side_stream is not None:
            # DJ_SIDE_MIN_MFLOP (A/B switch, default 0 = every weight gradient): launches below that many MFLOP stay on
            # the main stream -- a hand-over to the side stream is an event record + a cross-queue wait, 20-40 us during
            # which a 10 us kernel could have run where it stands
            mflop = 2e-6 * desc.batch * desc.out_h * desc.out_w * desc.out_c * desc.kernel_h * desc.kernel_w * desc.in_c
            if mflop < float(os.environ.get("DJ_SIDE_MIN_MFLOP", "0")):
                side = False
        if side and backward and self.side_stream is not None:
            fn = self._on_side(fn)
        if fwd_after is not None and not backward:
            return self.emit_side(fn, fwd_after)
        (self.bwd if backward else self.fwd).append(fn)

    def _on_side(self, fn):
        side, ready = self.side_stream, torch.cuda.Event()

        def run():
            if not self.side_enabled:
                return fn()
            ready.record()                      # everything issued so far on the main stream (dy, the memset)
            side.wait_event(ready)
            self._launch_on_side(fn)
            self._side_dirty = True
        return run

    def _launch_on_side(self, fn):
        """Run the C-ABI launches of `fn` on the side stream: they take their stream from `_lib.current_stream()`, set here
        (per thread) rather than from torch's current stream, whose context manager costs ~25 us of host time per use.  (`fn` issues
        nothing but C-ABI launches; work that goes through torch -- collectives -- uses `after_both_streams`.)"""
        prev = _lib.set_launch_stream(self.side_stream.cuda_stream)
        try:
            fn()
        finally:
            _lib.set_launch_stream(prev)

    def after_both_streams(self, fn):
        """Run `fn` with the side stream current, after everything issued so far on BOTH streams: what `fn` enqueues
        (a collective: torch.distributed orders its communication stream after the current stream) then depends on the
        weight gradients of the side stream and on the main stream's work without the main stream waiting for anybody."""
        if self.side_stream is None or not self.side_enabled:
            return fn()
        ev = torch.cuda.Event()
        ev.record()
        self.side_stream.wait_event(ev)
        with torch.cuda.stream(self.side_stream):
            return fn()

    def join_side(self):
        """Make the main stream wait for the side-stream launches issued so far."""
        if self._side_dirty:
            self._join_event.record(self.side_stream)
            torch.cuda.current_stream().wait_event(self._join_event)
            self._side_dirty = False

    def autotune(self, reps=2, verbose=False, measure=True):
        """Time every distinct conv geometry of this plan under each tile configuration / split-K factor on the
        plan's own buffers and register the fastest with the launcher (dj_conv2d_tune_set).  Results only depend on
        the geometry, so they are shared by all plans of the process.  measure=False: only apply what the in-tree table
        holds; other geometries keep the launcher's default variant (deterministic, nothing is run)."""
        import ctypes
        lib = _lib.load()
        ncfg = lib.dj_conv2d_tune_configs()
        names = [n for n, _ in _lib.ConvDesc._fields_][:15]
        done = 0
        grown = False
        for direction, desc, fn in self.conv_calls:
            key = (direction,) + tuple(getattr(desc, n) for n in names)
            if key in _TUNED:
                continue
            known = _db_lookup(key)
            if known is not None and known[0] < ncfg:
                sp = int(known[1])
                if len(known) > 3 and known[3]:
                    pass    # chosen inside the training step (tools/tune_in_step.py): variant and split factor as they are
                elif (direction & 3) == 2 and self.side_stream is not None:
                    # the table holds the split-K factor that is fastest for the weight-gradient GEMM ALONE; beside the
                    # data-gradient chain half as many workgroups (and half the atomic traffic) disturb the HBM-bound
                    # kernels of that chain less: +0.7 % on the step (1134 vs 1126 img/s, same box)
                    sp = max(1, (sp + 1) // 2)
                check(lib.dj_conv2d_tune_set(direction, desc, int(known[0]), sp), "tune_set")
                _TUNED[key] = (float(known[2]), int(known[0]), int(known[1]))
                continue
            if not measure:
                continue
            c0, s0 = ctypes.c_int(0), ctypes.c_int(1)
            check(lib.dj_conv2d_default_config(direction, desc, ctypes.byref(c0), ctypes.byref(s0)), "default_config")
            base = direction & 3
            if direction & 4 or direction == 9 or getattr(fn, "no_split", False):
                split_opts = [1]
            elif base == 2:
                kk = desc.batch * desc.out_h * desc.out_w
                split_opts = sorted({1, max(1, s0.value // 2), s0.value, min(max(1, kk // 128), s0.value * 2)})
            else:
                split_opts = sorted({1, s0.value, 2, 4})
            if base == 0 and not (direction & 4):
                # a split forward launch goes through slabs in the plan's workspace (sized so far for the pre-tuning
                # choice): make room for the largest factor tried, and do not time a factor whose slabs exceed the cap --
                # it would run the atomics fallback, not what the plan runs afterwards
                mn = desc.batch * desc.out_h * desc.out_w * desc.out_c
                split_opts = [sp for sp in split_opts if sp == 1 or sp * mn <= _Workspace.CAP_FLOATS]
                if not grown:
                    for ws in self._workspaces.values():
                        ws.fit_splits(max(8, s0.value))
                    grown = True
                if max(split_opts) > 8:
                    for ws in self._workspaces.values():
                        ws.fit_splits(max(split_opts))
            best = (float("inf"), c0.value, s0.value)
            # 'float32': the upper half of the indices names the split-bf16 kernels.  Where a geometry cannot run them the
            # launcher falls back to the fp32 variant of the same index: such a twin must not win on timing noise
            n_mfma = ncfg // 2 if self.compute_mode == 0 else ncfg
            for cfg in range(ncfg):
                for sp in split_opts:
                    check(lib.dj_conv2d_tune_set(direction, desc, cfg, sp), "tune_set")
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    e1.synchronize()
                    t = e0.elapsed_time(e1) / reps
                    if t < best[0] * (0.97 if cfg >= n_mfma and best[1] < n_mfma else 1.0):
                        best = (t, cfg, sp)
            check(lib.dj_conv2d_tune_set(direction, desc, best[1], best[2]), "tune_set")
            _TUNED[key] = best
            done += 1
            if verbose:
                import sys
                print("tuned dir=%d %s -> cfg %d splits %d (%.3f ms; default cfg %d splits %d)"
                      % (direction, key[1:], best[1], best[2], best[0], c0.value, s0.value), file=sys.stderr)
        return done

    def note_grad(self, spec):
        """Record that every launch writing `spec.grad` has been emitted (data-parallel bucketing)."""
        self.grad_ready[spec.key] = len(self.bwd)

    def on_backward(self, builder):
        """Register a function that appends this op's backward launches; builders run in reverse
        registration order once the forward lowering is complete."""
        self._bwd_builders.append(builder)

    def clear_gradients_first(self, flat_grads):
        """One memset of the flat gradient buffer at the start of backward replaces the per-layer split-K memsets."""
        assert not self.bwd
        self.bwd.append(lambda: flat_grads.zero_())
        self.grads_cleared = True

    def build_backward(self):
        for b in reversed(self._bwd_builders):
            b()
        self._bwd_builders = []
        if self.deferred_colsums:
            # short column sums (bias gradients) nobody reads before the optimizer: one launch for all of them
            self.emit_bwd(colsum_multi([p[:5] for p in self.deferred_colsums]))
            for p in self.deferred_colsums:
                self.note_grad(p[5])
            self.deferred_colsums = []

    # ---- gradients ------------------------------------------------------------
    def grad_of(self, v, zeroed=False):
        """-> (buffer, beta).  beta = 0 for the first writer in backward execution order.  zeroed=True: a first writer
        that would have to clear the buffer itself gets one that is already zero (see zeroed_each_step) and beta = 1."""
        if v.alias_of is not None:
            buf, beta = self.grad_of(v.alias_of, zeroed)
            return v.alias_view(buf), beta
        if v.grad is None:
            gdt = self.grad_dtype(v.buf)
            if zeroed:
                v.grad = GradRef(self.zeroed_each_step(*v.buf.shape, dtype=gdt))
                return v.grad.buf, 1
            v.grad = GradRef(self.empty(*v.buf.shape, dtype=gdt))
            return v.grad.buf, 0
        assert v.grad.mask_y is None, "cannot accumulate into a masked gradient reference"
        return v.grad.buf, 1

    def set_grad_ref(self, v, ref):
        assert v.alias_of is None and v.grad is None, "gradient of %s already has a writer" % v.name
        v.grad = ref

    # ---- execution -------------------------------------------------------------
    def run_forward(self):
        lib = _lib.load()
        prev = lib.dj_set_thread_compute_mode(self.compute_mode)
        prev_stream = _lib.set_launch_stream(torch.cuda.current_stream().cuda_stream)   # once per pass, not per launch
        try:
            for f in self.fwd:
                f()
        finally:
            _lib.set_launch_stream(prev_stream)
            lib.dj_set_thread_compute_mode(prev)

    def run_backward(self):
        lib = _lib.load()
        prev = lib.dj_set_thread_compute_mode(self.compute_mode)
        prev_stream = _lib.set_launch_stream(torch.cuda.current_stream().cuda_stream)
        try:
            for f in self.bwd:
                f()
        finally:
            _lib.set_launch_stream(prev_stream)
            lib.dj_set_thread_compute_mode(prev)
        self.join_side()
        for h in self.hooks_after_backward:
            h()


def rows_of(buf):
    """(rows, C, ld) of an NHWC / [.., C] buffer that may be a channel slice."""
    c = buf.shape[-1]
    rows = buf.numel() // c
    if buf.dim() >= 2 and buf.shape[-2] > 1:
        ld = buf.stride(-2)
    else:
        ld = c
        for d in range(buf.dim() - 2, -1, -1):
            if buf.shape[d] > 1:
                ld = buf.stride(d)
                break
    return rows, c, ld

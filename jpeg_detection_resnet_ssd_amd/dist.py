"""Data-parallel training across the MI355X GPUs of one node: one process per GPU, RCCL over xGMI
through torch.distributed (backend "nccl" is RCCL on ROCm).

Semantics reproduced from the reference's Horovod wrapper (classification_part/config/resnet/
config_file.py:121-150, classification_part/training.py:61-66): every rank holds a full weight replica,
runs forward/backward on its own shard of the global batch (BatchNormalization statistics, hard-negative
mining and the n_positive normalisation stay PER RANK, as Horovod leaves them), gradients are averaged
over ranks every step (hvd.DistributedOptimizer), rank 0's weights are broadcast at start
(BroadcastGlobalVariablesCallback(0)) and epoch metrics are averaged (MetricAverageCallback).

MI355X specifics: all gradients live in ONE flat HBM buffer, so a bucket is a contiguous slice (no
flatten/unflatten copies); buckets are formed in the order the backward pass finishes them and each
all-reduce is issued from inside the plan's launch sequence right after the bucket's last wgrad (ordered
after both of the plan's streams, without stalling the main one), so RCCL (on its own stream) overlaps the
rest of the backward.  xGMI is point-to-point (ring per link),
so buckets are large (default 16 MiB, `DJ_BUCKET_MB`) to stay bandwidth- rather than latency-bound; the 1/world_size
is folded into the SGD kernel instead of a separate scaling pass."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("DJ_FORCE_DIST", "0") == "1"   # exercise the RCCL path on a 1-rank communicator
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def plan_buckets(ready, sizes_offsets, bucket_bytes):
    """Partition gradient tensors into buckets.

    ready: {key: index in the backward launch list after which the gradient is final}
    sizes_offsets: {key: (offset, padded_size)} in floats inside the flat gradient buffer
    -> list of (launch_index, [(lo, hi), ...]) sorted by launch_index; ranges are merged when adjacent."""
    order = sorted(ready, key=lambda k: (ready[k], -sizes_offsets[k][0]))
    buckets, cur, cur_bytes, cur_idx = [], [], 0, 0
    for k in order:
        off, n = sizes_offsets[k]
        cur.append((off, off + n))
        cur_bytes += 4 * n
        cur_idx = max(cur_idx, ready[k])
        if cur_bytes >= bucket_bytes:
            buckets.append((cur_idx, _merge(cur)))
            cur, cur_bytes, cur_idx = [], 0, 0
    if cur:
        buckets.append((cur_idx, _merge(cur)))
    return buckets


def _merge(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


class GradientExchange(object):
    """All-reduce(sum) of slices of one flat buffer, issued bucket by bucket; `finish()` waits for all of
    them and returns the factor (1/world) the optimizer applies."""

    def __init__(self, flat, world):
        self.flat = flat
        self.world = world
        self.pending = []

    def launch(self, ranges):
        for lo, hi in ranges:
            self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        return 1.0 / self.world


class DataParallel(object):
    def __init__(self, model, bucket_mb=None):
        # 16 MiB: of the SSD300 (deconv) backward's 358 launches, 32 MiB buckets close after launch 85 (97 MB: fc6 and
        # the heads), 116, 171 and only then 331 and 358 -- the fourth one's exchange starts when 8 % of the backward is
        # left.  Half the size starts that tail earlier at the price of a few more (overlapped) collective launches.
        if bucket_mb is None:
            bucket_mb = float(os.environ.get("DJ_BUCKET_MB", "16"))
        self.model = model
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        model.dist = self
        self.exchange = None
        # training plans lowered BEFORE the model became data parallel (train_on_batch, or the trainer's warm-up, called
        # first) have no all-reduce in their launch lists: splice it in now, or the replicas would silently diverge
        for key, plan in list(model._plans.items()):
            if plan.training and key[2]:
                self.attach(plan)

    def broadcast_weights(self, src=0):
        """Rank `src`'s weights (trainable and BatchNormalization state) to every rank."""
        if dist.is_initialized():
            dist.broadcast(self.model.flat_all, src=src)

    def attach(self, plan):
        """Insert the bucketed all-reduces into the plan's backward launch list."""
        # the plan itself carries the mark (an id() in a set could be reused by a plan lowered after this one was freed:
        # Model.compile drops its plans -- the early-out below and the guard of finish_gradients would then both pass
        # for a plan without an exchange)
        if not dist.is_initialized() or getattr(plan, "_dp_attached", None) is self:
            return
        m = self.model
        if self.exchange is None:
            self.exchange = GradientExchange(m.flat_gradients, self.world)
        offs = m._store["offsets"]
        so, ready = {}, {}
        for w in m.weight_specs:
            if w.trainable:
                so[w.key] = (offs[id(w)], (w.size + 3) // 4 * 4)
                ready[w.key] = plan.grad_ready.get(w.key, len(plan.bwd))
        buckets = plan_buckets(ready, so, self.bucket_bytes)
        ex = self.exchange
        for idx, ranges in sorted(buckets, key=lambda b: -b[0]):
            # issued from the side stream: the collective waits for the weight gradients there (and, through an event,
            # for the main stream's bias / BatchNormalization gradients) while the main stream runs on -- a join here
            # would stall the data-gradient chain once per bucket (27.54 vs 27.89 ms/step on a 1-rank communicator)
            plan.bwd.insert(idx, (lambda r=ranges: plan.after_both_streams(lambda: ex.launch(r))))
        plan._dp_attached = self
        self.n_buckets = len(buckets)

    def finish_gradients(self, plan=None):
        """Wait for the step's all-reduces; -> the factor (1/world) the optimizer applies to the summed gradients."""
        if not dist.is_initialized():
            return 1.0
        if plan is not None and getattr(plan, "_dp_attached", None) is not self:
            raise RuntimeError("data parallel: the training plan that just ran has no gradient exchange attached "
                               "(its gradients are per-rank): replicas would diverge")
        if self.exchange is None:
            return 1.0
        return self.exchange.finish()

    def average_metrics(self, logs):
        if self.world == 1:
            return logs
        keys = sorted(k for k, v in logs.items() if isinstance(v, (int, float)))
        t = torch.tensor([float(logs[k]) for k in keys], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t)
        t = (t / self.world).cpu()
        out = dict(logs)
        for k, v in zip(keys, t.tolist()):
            out[k] = v
        return out

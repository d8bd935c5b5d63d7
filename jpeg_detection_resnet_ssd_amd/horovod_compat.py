"""`hvd`-shaped adapter over dist.py for the classification trainer: the calls the reference makes on horovod.keras
(classification_part/training.py:43-66,137-156; config/resnet/config_file.py:121-150) -- init / rank / size /
local_rank, DistributedOptimizer, BroadcastGlobalVariablesCallback, MetricAverageCallback,
LearningRateWarmupCallback -- mapped onto one-process-per-GPU RCCL data parallelism."""
import torch

from . import dist as djdist
from .keras.callbacks import Callback

_state = {"rank": 0, "size": 1, "local": 0}


def init():
    r, w, l = djdist.init_from_env()
    _state.update(rank=r, size=w, local=l)
    if torch.cuda.is_available():
        torch.cuda.set_device(l)


def rank():
    return _state["rank"]


def size():
    return _state["size"]


def local_rank():
    return _state["local"]


def DistributedOptimizer(optimizer):
    """Gradient averaging happens inside the plan (bucketed all-reduce + 1/size in the SGD kernel) once the model is
    wrapped by dist.DataParallel, which BroadcastGlobalVariablesCallback does; the optimizer object is unchanged."""
    optimizer._dj_distributed = True
    return optimizer


def _ensure_dp(model):
    if model.dist is None:
        djdist.DataParallel(model)
    return model.dist


class _BroadcastGlobalVariablesCallback(Callback):
    def __init__(self, root_rank=0):
        self.root_rank = root_rank

    def on_train_begin(self, logs=None):
        _ensure_dp(self.model).broadcast_weights(self.root_rank)


class _MetricAverageCallback(Callback):
    """fit_generator averages the epoch logs over ranks whenever the model is data parallel."""

    def on_train_begin(self, logs=None):
        _ensure_dp(self.model)


class _LearningRateWarmupCallback(Callback):
    """lr = initial_lr / size * (epoch * (size - 1) / warmup_epochs + 1) for epoch < warmup_epochs, adjusted every
    batch (Goyal et al. gradual warm-up, as horovod.keras implements it)."""

    def __init__(self, warmup_epochs=5, momentum_correction=True, steps_per_epoch=None, verbose=0):
        self.warmup_epochs, self.steps_per_epoch, self.verbose = warmup_epochs, steps_per_epoch, verbose
        self.initial_lr = None
        self.epoch = 0

    def on_train_begin(self, logs=None):
        self.initial_lr = self.model.optimizer.lr

    def on_epoch_begin(self, epoch, logs=None):
        self.epoch = epoch

    def on_batch_begin(self, batch, logs=None):
        steps = self.steps_per_epoch or getattr(self, "params", {}).get("steps") or 1
        e = self.epoch + batch / float(steps)
        if e < self.warmup_epochs and size() > 1:
            self.model.optimizer.lr = self.initial_lr / size() * (e * (size() - 1) / self.warmup_epochs + 1)

    def on_epoch_end(self, epoch, logs=None):
        if epoch == self.warmup_epochs - 1:
            self.model.optimizer.lr = self.initial_lr
            if self.verbose and rank() == 0:
                print("Epoch %d: finished gradual learning rate warmup to %g." % (epoch + 1, self.initial_lr))


class callbacks(object):
    BroadcastGlobalVariablesCallback = _BroadcastGlobalVariablesCallback
    MetricAverageCallback = _MetricAverageCallback
    LearningRateWarmupCallback = _LearningRateWarmupCallback

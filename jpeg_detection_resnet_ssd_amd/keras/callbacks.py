"""Minimal host-side versions of the Keras callbacks the reference's training scripts instantiate
(localisation_part/training_dct_pascal_j2d_resnet.py:295-320).  Control plane only: none of them
touches the compute path."""
import csv
import math
import os


class Callback(object):
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class TerminateOnNaN(Callback):
    def on_batch_end(self, batch, logs=None):
        loss = (logs or {}).get("loss")
        if loss is not None and (math.isnan(loss) or math.isinf(loss)):
            print("Batch %d: Invalid loss, terminating training" % batch)
            self.model.stop_training = True


class CSVLogger(Callback):
    def __init__(self, filename, separator=",", append=False):
        self.filename = filename
        self.sep = separator
        self.append = append
        self.keys = None

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs.keys())
        new = not (self.append and os.path.exists(self.filename) and os.path.getsize(self.filename) > 0)
        mode = "a" if (self.append or not new) else "w"
        with open(self.filename, mode, newline="") as f:
            w = csv.writer(f, delimiter=self.sep)
            if new:
                w.writerow(["epoch"] + self.keys)
            w.writerow([epoch] + [logs.get(k) for k in self.keys])
        self.append = True


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False,
                 mode="auto", period=1):
        self.filepath = filepath
        self.monitor = monitor
        self.verbose = verbose
        self.save_best_only = save_best_only
        self.save_weights_only = save_weights_only
        self.period = period
        # Keras 2.2.4 ModelCheckpoint: explicit 'min' / 'max'; 'auto' (and unknown modes, with a warning there) means
        # max when the monitored name contains 'acc' or starts with 'fmeasure', min otherwise
        if mode == "min":
            self.greater = False
        elif mode == "max":
            self.greater = True
        else:
            self.greater = ("acc" in monitor) or monitor.startswith("fmeasure")
        self.best = -math.inf if self.greater else math.inf
        self.since = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.since += 1
        if self.since < self.period:
            return
        self.since = 0
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None:
                return
            better = cur > self.best if self.greater else cur < self.best
            if not better:
                return
            self.best = cur
        if self.verbose:
            print("Epoch %05d: saving model to %s" % (epoch + 1, path))
        self.model.save_weights(path)


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto", min_delta=1e-4,
                 cooldown=0, min_lr=0, **kwargs):
        self.monitor, self.factor, self.patience = monitor, factor, patience
        self.verbose, self.min_delta, self.cooldown, self.min_lr = verbose, min_delta, cooldown, min_lr
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        # Keras: 'min', or 'auto' with a monitor that does not contain 'acc' -> improvement = decrease
        self.greater = mode == "max" or (mode not in ("min", "max") and "acc" in monitor)
        self.best = -math.inf if self.greater else math.inf
        self.wait = 0
        self.cooldown_counter = 0

    def _improved(self, cur):
        return cur > self.best + self.min_delta if self.greater else cur < self.best - self.min_delta

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._improved(cur):
            self.best = cur
            self.wait = 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                opt = self.model.optimizer
                if opt.lr > self.min_lr:
                    opt.lr = max(opt.lr * self.factor, self.min_lr)
                    if self.verbose:
                        print("Epoch %05d: ReduceLROnPlateau reducing learning rate to %s." % (epoch + 1, opt.lr))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto", **kwargs):
        self.monitor, self.min_delta, self.patience, self.verbose = monitor, abs(min_delta), patience, verbose
        self.greater = mode == "max" or (mode not in ("min", "max") and "acc" in monitor)
        self.best = -math.inf if self.greater else math.inf
        self.wait = 0
        self.stopped_epoch = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        better = cur > self.best + self.min_delta if self.greater else cur < self.best - self.min_delta
        if better:
            self.best = cur
            self.wait = 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True


class LearningRateScheduler(Callback):
    def __init__(self, schedule, verbose=0):
        self.schedule = schedule

    def on_epoch_begin(self, epoch, logs=None):
        try:
            lr = self.schedule(epoch, self.model.optimizer.lr)
        except TypeError:
            lr = self.schedule(epoch)
        self.model.optimizer.lr = float(lr)


class TensorBoard(Callback):
    """Accepted for drop-in compatibility; writes nothing (no TensorFlow here)."""

    def __init__(self, log_dir="./logs", **kwargs):
        self.log_dir = log_dir

"""Host-side callbacks against Keras 2.2.4's documented behaviour: mode resolution of ModelCheckpoint /
ReduceLROnPlateau / EarlyStopping, CSVLogger append, and the reference trainer's checkpoint-name round trip
(localisation_part/training_dct_pascal_j2d_resnet.py:295-336).  No GPU: a stub model records what the callbacks do."""
import csv

import pytest

from jpeg_detection_resnet_ssd_amd.keras import callbacks as C


class StubOptimizer(object):
    lr = 0.1


class StubModel(object):
    def __init__(self):
        self.saved = []
        self.stop_training = False
        self.optimizer = StubOptimizer()

    def save_weights(self, path):
        self.saved.append(path)


def _run(cb, series, key):
    m = StubModel()
    cb.set_model(m)
    cb.on_train_begin()
    for e, v in enumerate(series):
        cb.on_epoch_end(e, {key: v, "loss": 1.0})
        if m.stop_training:
            break
    return m


@pytest.mark.parametrize("monitor,mode,series,expect_epochs", [
    ("val_loss", "auto", [3.0, 2.0, 2.5, 1.0], [1, 2, 4]),        # auto + loss -> min
    ("val_acc", "auto", [0.1, 0.3, 0.2, 0.4], [1, 2, 4]),         # auto + 'acc' -> max
    ("fmeasure", "auto", [0.1, 0.3, 0.2, 0.4], [1, 2, 4]),        # auto + 'fmeasure...' -> max
    ("val_loss", "max", [3.0, 2.0, 2.5, 4.0], [1, 4]),            # explicit max on a name without 'acc' (was min)
    ("val_acc", "min", [0.5, 0.3, 0.4, 0.1], [1, 2, 4]),          # explicit min on an accuracy
    ("val_loss", "bogus", [3.0, 2.0, 2.5, 1.0], [1, 2, 4]),       # unknown mode falls back to auto
])
def test_model_checkpoint_mode(monitor, mode, series, expect_epochs, tmp_path):
    cb = C.ModelCheckpoint(str(tmp_path / "w-{epoch:02d}.h5"), monitor=monitor, save_best_only=True, mode=mode)
    m = _run(cb, series, monitor)
    assert m.saved == [str(tmp_path / ("w-%02d.h5" % e)) for e in expect_epochs]


def test_model_checkpoint_period_and_name_round_trip(tmp_path):
    pattern = str(tmp_path / "ssd300_pascal_07+12_epoch-{epoch:02d}_loss-{loss:.4f}_val_loss-{val_loss:.4f}.h5")
    cb = C.ModelCheckpoint(pattern, monitor="val_loss", save_best_only=False, period=2)
    m = StubModel()
    cb.set_model(m)
    for e in range(5):
        cb.on_epoch_end(e, {"loss": 2.0 - 0.1 * e, "val_loss": 3.0 - 0.1 * e})
    assert len(m.saved) == 2 and m.saved[-1].endswith("epoch-04_loss-1.7000_val_loss-2.7000.h5")
    # the trainer's --restart parsing of such a name (TRAIN_SSD:327-330)
    import os
    assert int(os.path.basename(m.saved[-1]).split("-")[1].split("_")[0]) == 4


def test_reduce_lr_on_plateau_and_early_stopping_modes():
    cb = C.ReduceLROnPlateau(monitor="val_acc", factor=0.5, patience=2, min_delta=0.0)    # auto + acc -> max
    m = _run(cb, [0.5, 0.6, 0.6, 0.6, 0.7], "val_acc")
    assert m.optimizer.lr == pytest.approx(0.05)          # epochs 2 and 3 did not improve -> one reduction
    cb = C.ReduceLROnPlateau(monitor="val_loss", factor=0.1, patience=1, cooldown=1, min_lr=0.005)
    m = _run(cb, [1.0, 1.0, 1.0, 1.0, 1.0, 1.0], "val_loss")
    assert m.optimizer.lr == pytest.approx(0.005)         # 0.1 -> 0.01 -> (cooldown) -> clipped at min_lr
    with pytest.raises(ValueError):
        C.ReduceLROnPlateau(factor=1.0)
    stop = C.EarlyStopping(monitor="val_acc", patience=2)
    m = _run(stop, [0.2, 0.3, 0.3, 0.25, 0.9], "val_acc")
    assert m.stop_training and stop.stopped_epoch == 3
    stop = C.EarlyStopping(monitor="val_loss", patience=1, mode="max")
    m = _run(stop, [1.0, 2.0, 1.5], "val_loss")
    assert m.stop_training and stop.stopped_epoch == 2


def test_csv_logger_appends_across_runs(tmp_path):
    path = str(tmp_path / "log.csv")
    for first_epoch in (0, 2):                              # a run, then a --restart run appending to the same file
        cb = C.CSVLogger(path, separator=",", append=True)
        cb.set_model(StubModel())
        cb.on_train_begin()
        for e in range(first_epoch, first_epoch + 2):
            cb.on_epoch_end(e, {"loss": 1.0 / (e + 1), "val_loss": 2.0 / (e + 1), "lr": 0.001})
    rows = list(csv.reader(open(path)))
    assert rows[0] == ["epoch", "loss", "lr", "val_loss"] and [r[0] for r in rows[1:]] == ["0", "1", "2", "3"]


def test_compile_rejects_what_cannot_be_lowered():
    """Without a GPU the probe of a custom loss is deferred, but a non-callable / unknown loss name fails at compile."""
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.keras.layers import Conv2D, Input
    from jpeg_detection_resnet_ssd_amd.keras.models import Model
    from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
    K.clear_session()
    inp = Input((8, 8, 32))
    model = Model(inp, Conv2D(8, 1, name="c")(inp))
    with pytest.raises(NotImplementedError, match="no MI355X lowering"):
        model.compile(optimizer=SGD(), loss="mean_squared_error")
    with pytest.raises(NotImplementedError):
        model.compile(optimizer="adam", loss="categorical_crossentropy")
    model.compile(optimizer="sgd", loss="categorical_crossentropy")
    assert model.loss == ("cce", None)

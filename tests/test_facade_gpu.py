"""The Keras-shaped facade the north star calls "drop-in", under test on the GPU:
`fit_generator` with the reference trainer's callback set, checkpoints written by ModelCheckpoint and read back with
`load_weights(by_name=True)`, the `--restart` path of the entry script, and the Keras loss protocol
(localisation_part/training_dct_pascal_j2d_resnet.py:137-156,295-336)."""
import csv
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCHI, BATCH = "cb5_only", 2


def _generator(sizes, n_batches, seed0):
    """Keras generator contract: yields (inputs, targets) forever (here: a cycle of `n_batches` synthetic batches)."""
    from test_ssd_gpu import make_batch
    batches = [make_batch(ARCHI, sizes, BATCH, seed=seed0 + i) for i in range(n_batches)]
    i = 0
    while True:
        yield batches[i % n_batches]
        i += 1


def test_fit_generator_with_the_trainers_callbacks(cuda, tmp_path):
    from jpeg_detection_resnet_ssd_amd.keras.callbacks import (CSVLogger, EarlyStopping, ModelCheckpoint,
                                                               ReduceLROnPlateau, TerminateOnNaN)
    from test_ssd_gpu import build, make_batch
    model, sizes = build(ARCHI)
    pattern = str(tmp_path / "ssd300_epoch-{epoch:02d}_loss-{loss:.4f}_val_loss-{val_loss:.4f}.h5")
    log = str(tmp_path / "training_log.csv")
    seen = []

    class Spy(EarlyStopping):      # sees the same logs the other callbacks see
        def on_epoch_end(self, epoch, logs=None):
            seen.append(dict(logs))
            super(Spy, self).on_epoch_end(epoch, logs)

    cbs = [ModelCheckpoint(filepath=pattern, monitor="val_loss", verbose=0, save_best_only=True, save_weights_only=False,
                           mode="auto", period=1),
           CSVLogger(filename=log, separator=",", append=True),
           ReduceLROnPlateau(monitor="val_loss", factor=0.1, patience=1, min_delta=1e9),   # "never improves": fires
           TerminateOnNaN(), Spy(monitor="val_loss", min_delta=0, patience=10)]
    hist = model.fit_generator(generator=_generator(sizes, 3, 100), steps_per_epoch=3, epochs=3, callbacks=cbs,
                               validation_data=_generator(sizes, 2, 900), validation_steps=2, initial_epoch=0, verbose=0)
    assert hist.epoch == [0, 1, 2] and len(hist.history["loss"]) == 3 and len(hist.history["val_loss"]) == 3
    assert all(np.isfinite(v) for v in hist.history["loss"] + hist.history["val_loss"])
    assert model.optimizer.iterations == 9
    # CSVLogger: header + one row per epoch, the columns every epoch's logs carried
    rows = list(csv.reader(open(log)))
    assert rows[0][0] == "epoch" and {"loss", "val_loss", "lr"} <= set(rows[0]) and [r[0] for r in rows[1:]] == ["0", "1", "2"]
    col = rows[0].index("val_loss")
    assert [float(r[col]) for r in rows[1:]] == pytest.approx(hist.history["val_loss"], rel=1e-9)
    # ModelCheckpoint(save_best_only): one file per epoch that improved val_loss, named from the epoch's own logs
    files = sorted(glob.glob(str(tmp_path / "ssd300_epoch-*.h5")))
    best, expect = float("inf"), []
    for e, logs in enumerate(seen):
        if logs["val_loss"] < best:
            best = logs["val_loss"]
            expect.append(pattern.format(epoch=e + 1, **logs))
    assert files == sorted(expect) and len(files) >= 1
    # ReduceLROnPlateau: patience 1 with an unreachable min_delta -> the rate dropped by `factor` after epoch 1
    # (the first epoch sets `best`, the second and third do not beat it by 1e9)
    assert hist.history["lr"][0] == pytest.approx(0.001) and model.optimizer.lr <= 0.0001 * (1 + 1e-6)

    # checkpoint -> fresh model -> load_weights(by_name=True) -> identical predictions
    x, _ = make_batch(ARCHI, sizes, BATCH, seed=5)
    last = expect[-1]
    saved = dict(np.load(last))
    fresh, _ = build(ARCHI, seed=7)                      # other random init
    n = fresh.load_weights(last, by_name=True)
    assert n == len(fresh.weight_specs) == len(saved)
    probe, _ = build(ARCHI, seed=8)
    probe.set_weights_dict(saved, strict=True)
    # same weights, two models: bit-identical predictions.  (The forward pass has no arrival-order arithmetic: split-K
    # launches -- fc6 with K = 18432, the small maps behind it -- go through slabs and a fixed-order reduction,
    # dj_conv2d_nhwc_fwd_ws; until round 2 they used fp32 atomics and only the 38x38 source was reproducible.)
    a, b = fresh.predict(x, batch_size=BATCH), probe.predict(x, batch_size=BATCH)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, fresh.predict(x, batch_size=BATCH))

    # by_name with a SUBSET of the names plus names the model does not have: Keras loads what matches, silently skips
    # the rest; by_name=False insists on every weight
    subset = {k: v for k, v in saved.items() if k.startswith("fc")}
    subset["not_a_layer/kernel"] = np.zeros((3, 3), np.float32)
    part = str(tmp_path / "subset.h5")
    with open(part, "wb") as f:
        np.savez(f, **subset)
    other, _ = build(ARCHI, seed=9)
    before = other.get_weights_dict()
    assert other.load_weights(part, by_name=True) == len(subset) - 1
    after = other.get_weights_dict()
    for k in after:
        np.testing.assert_array_equal(after[k], saved[k] if k in subset else before[k])
    with pytest.raises(ValueError):
        other.load_weights(part, by_name=False)
    # a shape mismatch under a matching name is an error, as in Keras
    bad = dict(subset)
    bad["fc7/kernel"] = np.zeros((1, 1, 2, 2), np.float32)
    with open(part, "wb") as f:
        np.savez(f, **bad)
    with pytest.raises(ValueError):
        other.load_weights(part, by_name=True)


def test_validation_steps_default_and_early_stopping(cuda):
    from jpeg_detection_resnet_ssd_amd.keras.callbacks import EarlyStopping
    from test_ssd_gpu import build, make_batch
    model, sizes = build(ARCHI)
    with pytest.raises(ValueError, match="validation_steps"):
        model.fit_generator(_generator(sizes, 1, 1), steps_per_epoch=1, epochs=1, validation_data=_generator(sizes, 1, 2),
                            verbose=0)

    class Seq(object):                                   # keras.utils.Sequence surface: __len__ + __getitem__
        def __init__(self):
            self.items = [make_batch(ARCHI, sizes, BATCH, seed=40 + i) for i in range(2)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

        def __iter__(self):
            while True:
                for it in self.items:
                    yield it

    stop = EarlyStopping(monitor="val_loss", min_delta=1e9, patience=1)     # cannot improve: stops after epoch 2
    hist = model.fit_generator(_generator(sizes, 2, 10), steps_per_epoch=2, epochs=6, validation_data=Seq(),
                               callbacks=[stop], verbose=0)
    assert hist.epoch == [0, 1] and stop.stopped_epoch == 1 and model.stop_training


def test_loss_protocol(cuda):
    """compile(loss=...) takes the reference's SSDLoss.compute_loss (fused kernels) or ANY loss(y_true, y_pred) in torch
    ops (differentiated by torch autograd with respect to y_pred, the model's backward pass is the HIP launch list);
    what cannot work fails at compile time with a message.  Here the oracle's restatement of the SSD loss is passed as
    the user's loss: the loss value and every parameter gradient must equal the fused path's."""
    from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
    from jpeg_detection_resnet_ssd_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    from oracle import keras_ops as ko
    from test_ssd_gpu import build, make_batch, perturb_weights
    model, sizes = build(ARCHI)
    w0 = perturb_weights(model)
    x, y = make_batch(ARCHI, sizes, BATCH, seed=21)
    loss_fused = model.train_on_batch(x, y)
    g_fused = model.flat_gradients.clone()

    model.set_weights_dict(w0)
    model.compile(optimizer=SGD(lr=0.001, momentum=0.9), loss=lambda t, p: ko.ssd_loss(t, p))
    loss_user = model.train_on_batch(x, y)
    g_user = model.flat_gradients.clone()
    assert abs(loss_user - loss_fused) <= 1e-5 * abs(loss_fused)
    assert float((g_user - g_fused).norm()) <= 1e-4 * float(g_fused.norm())

    # the loss objects are callables in their own right (Keras loss protocol): (batch,) vector on device tensors
    plan = model._plan(BATCH, True, True)
    yt, yp = plan.y_true, plan.outputs[0].buf
    vec = SSDLoss(neg_pos_ratio=3, alpha=1.0).compute_loss(yt, yp)
    ref = ko.ssd_loss(yt.double().cpu(), yp.double().cpu())
    assert vec.shape == (BATCH,) and float((vec.double().cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    with pytest.raises(TypeError, match="CUDA"):
        SSDLoss().compute_loss(yt.cpu(), yp.cpu())

    # compile-time failures, each with a reason
    with pytest.raises(TypeError, match="cannot be compiled"):
        model.compile(optimizer=SGD(), loss=lambda t, p: np.asarray(p.detach().cpu()).sum())     # leaves torch
    with pytest.raises(TypeError, match="cannot be compiled"):
        model.compile(optimizer=SGD(), loss=lambda t: t)                                         # wrong signature
    with pytest.raises(NotImplementedError, match="no MI355X lowering"):
        model.compile(optimizer=SGD(), loss="mean_squared_error")


def test_entry_script_restart(cuda, tmp_path):
    """`training_dct_pascal_j2d_resnet.py` end to end on synthetic data: 2 epochs, then `--restart <checkpoint>` --
    weights come back through load_weights(by_name=True) and the epoch counter from the file name
    (TRAIN_SSD:137-149,301-336: `int(restart.split('-')[1].split('_')[0])`)."""
    env = dict(os.environ, LOCAL_WORK_DIR=str(tmp_path), EXPERIMENTS_OUTPUT_DIRECTORY=str(tmp_path / "out"),
               DJ_AUTOTUNE="table")
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(ROOT, "training_dct_pascal_j2d_resnet.py"), "-vd", "0", "--crop", "--p07p12",
            "--reg", "--resnet", "--archi", ARCHI, "--steps_per_epoch", "2", "--batch_size", "2", "--synthetic_images", "16"]
    r = subprocess.run(base + ["--epochs", "2"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = tmp_path / "out"
    ckpts = sorted(glob.glob(str(out / "ssd300_pascal_07+12_epoch-*.h5")))
    assert ckpts, os.listdir(out)
    rows = list(csv.reader(open(out / "ssd300_pascal_07+12_training_log.csv")))
    assert [r_[0] for r_ in rows[1:]] == ["0", "1"]
    last = ckpts[-1]
    epoch_in_name = int(os.path.basename(last).split("-")[1].split("_")[0])
    r = subprocess.run(base + ["--epochs", str(epoch_in_name + 1), "--restart", last], capture_output=True, text=True,
                       env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rows = list(csv.reader(open(out / "ssd300_pascal_07+12_training_log.csv")))
    # CSVLogger(append=True): the resumed run continues the same log at epoch index `epoch_in_name`, runs one epoch
    assert [r_[0] for r_ in rows[1:]] == ["0", "1", str(epoch_in_name)]
    assert ("Epoch %d/%d" % (epoch_in_name + 1, epoch_in_name + 1)) in r.stdout


@pytest.mark.parametrize("archi", ["deconv", "ssd_custom", "up_sampling"])
def test_training_forward_pass_is_bit_reproducible(archi, cuda):
    """No arrival-order arithmetic in the forward pass, training mode included: split-K launches reduce their slabs in a
    fixed order (and take the BatchNormalization statistics there), Conv2DTranspose runs unsplit, statistics are two-stage
    sums, the predictor heads on the side stream are ordered by events.  Three evaluations of the same batch: identical
    predictions, bit for bit, and identical batch statistics in every BatchNormalization."""
    from test_ssd_gpu import build, make_batch, perturb_weights
    model, sizes = build(archi)
    perturb_weights(model)
    x, y = make_batch(archi, sizes, 4, seed=17)
    plan = model._plan(4, True, True)
    model._upload(plan, x, y)
    runs = []
    for _ in range(3):
        plan.run_forward()
        torch.cuda.synchronize()
        runs.append([plan.outputs[0].buf.clone(), plan.loss_out.clone()])
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0])
        assert torch.equal(r[1], runs[0][1])

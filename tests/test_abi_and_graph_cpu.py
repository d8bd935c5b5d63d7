"""CPU: the C-ABI library loads and exports every symbol include/dj_hip.h declares (no compute calls), the
model builders reproduce the reference's graph structure, and argument validation raises like the reference."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from jpeg_detection_resnet_ssd_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "dj_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dj_[a-z0-9_]+)\s*\(", header))
    declared.discard("dj_conv2d_desc")
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "library does not export " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.dj_abi_version() == _lib.ABI_VERSION == 3
    assert lib.dj_reduce_rows(46208) == 722


def test_conv_desc_argument_errors_surface_through_the_abi():
    """Bad descriptors are refused on the host before any launch (no GPU needed)."""
    import ctypes
    from jpeg_detection_resnet_ssd_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(1, 8, 8, 16, 8, 8, 16, 3, 3, 1, 1, 1, 1, 1, 1, 8, 16)   # ld_x < in_c
    assert lib.dj_conv2d_fwd_stats_rows(ctypes.byref(d)) < 0
    assert b"ld_x" in lib.dj_last_error()
    with pytest.raises(_lib.DjError):
        _lib.check(lib.dj_conv2d_fwd_stats_rows(ctypes.byref(d)), "stats_rows")


def test_dgrad_with_bn_backward_statistics_refuses_what_it_cannot_do():
    """dj_conv2d_nhwc_dgrad_bnbwd checks its arguments on the host, before any launch: missing tensors, scale without
    shift, a z pitch below the channel count, and strided 1x1 convolutions (their dx is scattered, the accumulator tile is
    not the gradient tile)."""
    import ctypes
    from jpeg_detection_resnet_ssd_amd import _lib
    lib = _lib.load()
    fake = 0x1000      # never dereferenced: every call below must fail in the argument checks
    ok = _lib.ConvDesc(2, 8, 8, 32, 8, 8, 64, 3, 3, 1, 1, 1, 1, 1, 1, 32, 64)
    call = lib.dj_conv2d_nhwc_dgrad_bnbwd
    assert call(ctypes.byref(ok), fake, fake, fake, None, 32, fake, fake, fake, fake, fake, None) < 0
    assert b"null" in lib.dj_last_error()
    assert call(ctypes.byref(ok), fake, fake, fake, fake, 32, fake, fake, fake, None, fake, None) < 0
    assert b"scale/shift" in lib.dj_last_error()
    assert call(ctypes.byref(ok), fake, fake, fake, fake, 16, fake, fake, None, None, fake, None) < 0
    assert b"ld_z" in lib.dj_last_error()
    strided = _lib.ConvDesc(2, 8, 8, 32, 4, 4, 64, 1, 1, 2, 2, 1, 1, 0, 0, 32, 64)
    assert call(ctypes.byref(strided), fake, fake, fake, fake, 32, fake, fake, None, None, fake, None) < 0
    assert b"strided" in lib.dj_last_error()


def test_same_padding_rule():
    from jpeg_detection_resnet_ssd_amd.kernels import conv_geometry, same_padding
    assert same_padding(38, 2, 1) == (0, 1, 38)        # even kernel: pad after only
    assert same_padding(38, 3, 1) == (1, 1, 38)
    assert same_padding(5, 3, 1, 6) == (6, 6, 5)       # fc6, dilation 6
    assert same_padding(38, 1, 2) == (0, 0, 19)
    assert conv_geometry(5, 5, (3, 3), (2, 2), ((1, 1), (1, 1)), (1, 1)) == (1, 1, 3, 3)   # conv6_2
    assert conv_geometry(19, 19, (1, 1), (2, 2), "valid", (1, 1)) == (0, 0, 10, 10)


@pytest.mark.parametrize("archi,boxes,params,convs,bns", [
    ("ssd_custom", 8732, 52048238, 87, 71), ("deconv", 6716, 51708206, 76, 53), ("up_sampling", 6716, 51675310, 74, 53),
    ("y_cb4_cbcr_cb5", 6716, None, None, None), ("cb5_only", 6716, None, None, None)])
def test_ssd_graph_structure(archi, boxes, params, convs, bns):
    from jpeg_detection_resnet_ssd_amd import workloads
    model, sizes = workloads.build_ssd(archi, compile_model=False)
    assert model.outputs[0].shape == (None, boxes, 33)
    if params:
        assert model.count_params() == params
        kinds = [l.__class__.__name__ for l in model.layers]
        assert kinds.count("Conv2D") + kinds.count("Conv2DTranspose") == convs
        assert kinds.count("BatchNormalization") == bns
    names = [l.name for l in model.layers]
    for n in ["pool5_ssd", "fc6", "fc7", "conv6_1", "conv6_padding", "conv6_2", "conv9_2", "conv4_3_norm",
              "conv4_3_norm_mbox_conf_21", "fc7_mbox_loc", "conv9_2_mbox_priorbox", "mbox_conf", "mbox_conf_softmax",
              "predictions_ssd", "res5c_branch2c", "bn5c_branch2c"]:
        assert n in names, n
    # the reference trainer reads the predictor sizes back like this (training_dct_pascal_j2d_resnet.py:244-249)
    got = [model.get_layer("%s_mbox_conf_21" % s).output_shape[1:3]
           for s in ["conv4_3_norm", "fc7", "conv6_2", "conv7_2", "conv8_2", "conv9_2"]]
    assert [tuple(g) for g in got] == [tuple(s) for s in sizes]
    if archi == "ssd_custom":
        assert names[:6] == ["input_1", "input_2", "batch_normalization_1", "res1a2_branch2a", "bn1a2_branch2a", "activation_1"]
        assert [t.shape for t in model.inputs] == [(None, 38, 38, 64), (None, 19, 19, 128)]
        assert "batch_normalization_2" in names and "res2a5_branch1" in names
        l2s = {w.key: w.l2 for w in model.weight_specs}
        assert l2s["fc6/kernel"] == 0.0005 and l2s["fc6/bias"] == 0 and l2s["res5c_branch2c/kernel"] == 0
        assert l2s["conv4_3_norm_mbox_loc/kernel"] == 0.0005
        assert model.get_layer("conv4_3_norm").weight_specs[0].key == "conv4_3_norm/conv4_3_norm_gamma"
    if archi == "deconv":
        assert [t.shape for t in model.inputs] == [(None, 38, 38, 64), (None, 19, 19, 64), (None, 19, 19, 64)]
        assert "conv2d_transpose_1" in names and "conv2d_transpose_2" in names
        assert model.get_layer("conv2d_transpose_1").weight_specs[0].shape == (2, 2, 64, 64)
    if archi == "y_cb4_cbcr_cb5":
        assert "res2a4_branch2a" not in names   # dead branch of the reference: built but not part of the Model


def test_builder_argument_validation_matches_reference_messages():
    from jpeg_detection_resnet_ssd_amd import workloads
    from jpeg_detection_resnet_ssd_amd.models.keras_ssd300_dct_j2d_resnet import (ssd_resnet_EF_layers_custom,
                                                                                 ssd_resnet_EF_layers_identical)
    base = dict(workloads.SSD_ARGS)
    with pytest.raises(ValueError, match="Unknown network architecture"):
        ssd_resnet_EF_layers_identical(archi="nope", **base)
    bad = dict(base, scales=[0.1, 0.2])
    with pytest.raises(ValueError, match="len\\(scales\\) == 7"):
        ssd_resnet_EF_layers_custom(**bad)
    bad = dict(base, variances=[0.1, 0.1, 0.2])
    with pytest.raises(ValueError, match="4 variance values"):
        ssd_resnet_EF_layers_custom(**bad)
    bad = dict(base, variances=[0.1, 0.1, 0.2, -1])
    with pytest.raises(ValueError, match="All variances must be >0"):
        ssd_resnet_EF_layers_custom(**bad)
    bad = dict(base, steps=[8, 16])
    with pytest.raises(ValueError, match="step value per predictor layer"):
        ssd_resnet_EF_layers_custom(**bad)
    bad = dict(base, mode="bogus")
    with pytest.raises(ValueError, match="`mode` must be one of"):
        ssd_resnet_EF_layers_custom(**bad)
    bad = dict(base, aspect_ratios_per_layer=None, aspect_ratios_global=None)
    with pytest.raises(ValueError, match="cannot both be None"):
        ssd_resnet_EF_layers_custom(**bad)


def test_compute_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from jpeg_detection_resnet_ssd_amd import workloads
    model, sizes = workloads.build_ssd("ssd_custom")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.train_on_batch([np.zeros((1, 38, 38, 64)), np.zeros((1, 19, 19, 128))], np.zeros((1, 8732, 33)))


def test_synthetic_dct_follows_jpeg2dct_semantics():
    from jpeg_detection_resnet_ssd_amd.data import synthetic_dct as sd
    assert list(sd.LUMA_Q75[0]) == [8, 6, 5, 8, 12, 20, 26, 31]      # tests_generators.py:66-68: -616 = -77 * 8
    assert list(sd.CHROMA_Q75[0]) == [9, 9, 12, 24, 50, 50, 50, 50]
    y, cbcr = sd.dct_batch(2, seed=3)
    assert y.shape == (2, 38, 38, 64) and cbcr.shape == (2, 19, 19, 128) and y.dtype == np.float32
    assert np.all(np.round(y / sd.LUMA_Q75.reshape(64)) * sd.LUMA_Q75.reshape(64) == y)   # de-quantised = level * table
    assert (y == 0).mean() > 0.5
    ys, cb, cr = sd.dct_batch(1, seed=3, split_chroma=True)
    assert cb.shape == (1, 19, 19, 64) and cr.shape == (1, 19, 19, 64)
    flat = np.full((300, 300, 3), 200, np.uint8)
    yf, cbf, crf = sd.rgb_to_dct(flat)
    assert yf[0, 0, 0] == np.round((200 - 128) * 8 / 8) * 8 and np.all(yf[..., 1:] == 0) and np.all(cbf == 0)


@pytest.mark.parametrize("archi,params,inputs", [
    ("resnet_rgb", 25636712, [(None, 224, 224, 3)]),
    ("deconv", 28434280, [(None, 28, 28, 64), (None, 14, 14, 64), (None, 14, 14, 64)]),
    ("late_concat_rfa_thinner", 28726632, [(None, 28, 28, 64), (None, 14, 14, 128)]),
    ("up_sampling_rfa", 28401384, [(None, 28, 28, 64), (None, 14, 14, 128)]),
    ("up_sampling", None, None), ("cb5_only", None, None), ("late_concat_more_channels", None, None),
    ("y_cb4_cbcr_cb5", None, None)])
def test_classifier_graph_structure(archi, params, inputs):
    """ResNet50RGB has the stock Keras ResNet50 parameter count; the DCT archis match SURVEY 8(d)."""
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.vgg_jpeg_keras.networks.resnet_dct import ResNet50Custom, ResNet50RGB
    K.clear_session()
    m = ResNet50RGB(weights=None, archi="ignored") if archi == "resnet_rgb" else ResNet50Custom(weights=None, archi=archi)
    assert m.outputs[0].shape == (None, 1000)
    if params:
        assert m.count_params() == params
        assert [t.shape for t in m.inputs] == inputs
    names = [l.name for l in m.layers]
    assert "avg_pool" in names and "fc1000" in names and "res5c_branch2c" in names
    if archi == "resnet_rgb":
        assert names[:6] == ["input_1", "conv1_pad", "conv1", "bn_conv1", "activation_1", "pool1_pad"]
    with pytest.raises(RuntimeError, match="no network"):
        ResNet50Custom(weights="imagenet", archi="deconv")
    with pytest.raises(ValueError):
        ResNet50Custom(weights=None, archi="bogus")


def test_classification_config_and_horovod_scaling_rules():
    """TrainingConfiguration surface + the reference's Horovod scaling rules (config/resnet/config_file.py:121-150)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cfg_resnet", os.path.join(ROOT, "config", "resnet", "config_file.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    K.clear_session()
    cfg = mod.TrainingConfiguration(deconv=True, archi="deconv", load_pretrained_weights=False)
    assert (cfg.batch_size, cfg.steps_per_epoch, cfg.epochs, cfg.validation_steps) == (256, 5000, 120, 195)
    assert cfg.optimizer.get_config() == {"lr": 0.1, "momentum": 0.9, "decay": 0.0001, "nesterov": True}
    assert len(cfg.metrics) == 2 and cfg.network.count_params() == 28434280

    class FakeHvd:
        class callbacks:
            BroadcastGlobalVariablesCallback = staticmethod(lambda r: ("bcast", r))
            MetricAverageCallback = staticmethod(lambda: "avg")
            LearningRateWarmupCallback = staticmethod(lambda warmup_epochs, verbose: ("warmup", warmup_epochs))

        @staticmethod
        def size():
            return 16

        @staticmethod
        def rank():
            return 0

        @staticmethod
        def DistributedOptimizer(o):
            return o
    cfg.prepare_horovod(FakeHvd)
    assert abs(cfg.optimizer.lr - 0.1 * 16 / 4) < 1e-12            # lr * size / batch_size_divider
    assert cfg.batch_size == 64 and cfg.steps_per_epoch == 5000 // 4 and cfg.validation_steps == 3 * 195 // 16
    assert cfg.callbacks[0] == ("bcast", 0) and cfg.callbacks[2] == ("warmup", 5)
    cfg.prepare_training_generators()
    x, y = cfg.train_generator[0]
    assert [a.shape for a in x] == [(64, 28, 28, 64), (64, 14, 14, 64), (64, 14, 14, 64)] and y.shape == (64, 1000)


def test_tuning_tables_follow_the_arithmetic_mode():
    """`K.set_floatx` switches the in-tree tile table (fp32: tuned/gfx950_conv.json, fp16/bf16 MFMA:
    tuned/gfx950_conv_f16.json); both hold the benchmark geometries, entries are [cfg, splits, ms] with an optional
    trailing 1 for choices made inside the training step, and every cfg index exists in the library."""
    import json
    from jpeg_detection_resnet_ssd_amd import _lib, engine
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    lib = _lib.load()
    ncfg = lib.dj_conv2d_tune_configs()                      # 'float32': fp32 MFMA variants + their split-bf16 twins
    assert ncfg % 2 == 0
    key = "0,32,10,10,2048,10,10,1024,3,3,1,1,6,6,6,6"      # fc6 forward, batch 32
    try:
        assert K.floatx() == "float32"
        fp32 = engine._tune_db()
        K.set_floatx("float16")
        assert lib.dj_conv2d_tune_configs() == ncfg // 2
        half = engine._tune_db()
        assert half is not fp32 and key in fp32 and key in half
        K.set_floatx("bfloat16")
        assert engine._tune_db() is not fp32
        K.set_floatx("float32x3")                            # fp32 products as three bf16 MFMAs: a table of its own
        x3 = engine._tune_db()
        assert K.floatx() == "float32x3" and x3 is not fp32 and x3 is not half and key in x3
        K.set_floatx("float32x6")
        assert K.floatx() == "float32x6" and engine._tune_db() is not fp32
        K.set_floatx("float32_mfma")                         # fp32 MFMA instructions only: the table measured over those
        mfma = engine._tune_db()
        assert mfma is not fp32 and key in mfma and lib.dj_conv2d_tune_configs() == ncfg // 2
        assert all(v[0] < ncfg // 2 for v in mfma.values())
        with pytest.raises(ValueError):
            K.set_floatx("float64")
    finally:
        K.set_floatx("float32")
    assert engine._tune_db() is fp32
    for path in (engine._TUNE_DB, engine._TUNE_DB_MFMA, engine._TUNE_DB_LOWP, engine._TUNE_DB_X3):
        with open(path) as f:
            table = json.load(f)
        assert table["arch"] == "gfx950" and table["n_configs"] == (ncfg if path == engine._TUNE_DB else ncfg // 2)
        for k, v in table["entries"].items():
            assert len(k.split(",")) == 16 and len(v) in (3, 4), k
            assert 0 <= v[0] < ncfg and v[1] >= 1, k
            if int(k.split(",")[0]) == 4:
                assert v[1] == 1, k           # a forward launch that takes BatchNormalization statistics cannot split K


def test_entry_scripts_and_tools_compile():
    """The trainer / evaluation entry scripts and the measurement tools are only run on the GPU box: keep them at least
    syntactically alive on every CPU run."""
    import glob
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "*.py")))
    assert len(files) >= 15
    for f in files:
        py_compile.compile(f, doraise=True)


def test_library_settings_are_thread_safe_and_per_thread():
    """include/dj_hip.h, 'state': the arithmetic mode is a process default with a per-thread override, the tuner's
    override table is keyed by geometry AND mode and may be written from several threads at once (no GPU needed: none of
    these entry points launches anything)."""
    import ctypes
    import threading
    from jpeg_detection_resnet_ssd_amd import _lib
    lib = _lib.load()
    assert lib.dj_set_thread_compute_mode(-1) in (-1, 0, 1, 2)
    assert lib.dj_set_compute_mode(0) in (0, 1, 2) and lib.dj_get_compute_mode() == 0
    seen, errors = {}, []
    barrier = threading.Barrier(3)

    def worker(tid, mode):
        try:
            d = _lib.ConvDesc(4, 10, 10, 64, 10, 10, 64, 3, 3, 1, 1, 1, 1, 1, 1, 64, 64)
            c, sp = ctypes.c_int(0), ctypes.c_int(0)
            assert lib.dj_get_compute_mode() == 0                 # a new thread follows the process default
            assert lib.dj_set_thread_compute_mode(mode) == -1
            barrier.wait()
            for i in range(3000):
                assert lib.dj_get_compute_mode() == mode           # ... and nobody else's override leaks in
                _lib.check(lib.dj_conv2d_tune_set(1, d, (tid + i) % 3, 1 + i % 4), "tune_set")
                d2 = _lib.ConvDesc(4, 10, 10, 64 + 32 * (i % 5), 10, 10, 64, 3, 3, 1, 1, 1, 1, 1, 1, 64 + 32 * (i % 5), 64)
                _lib.check(lib.dj_conv2d_tune_set(2, d2, i % 3, 1), "tune_set")
                _lib.check(lib.dj_conv2d_default_config(0, d2, ctypes.byref(c), ctypes.byref(sp)), "default_config")
                if i % 7 == 0:
                    _lib.check(lib.dj_conv2d_tune_set(2, d2, -1, 1), "tune_set")
            # a bad call's message belongs to the calling thread
            assert lib.dj_conv2d_tune_set(0, None, 0, 1) < 0
            seen[tid] = (lib.dj_get_compute_mode(), lib.dj_last_error().decode())
            lib.dj_set_thread_compute_mode(-1)
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=worker, args=(t, m)) for t, m in ((0, 0), (1, 1), (2, 2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert {k: v[0] for k, v in seen.items()} == {0: 0, 1: 1, 2: 2}
    assert all("tune_set" in v[1] for v in seen.values())
    assert lib.dj_get_compute_mode() == 0                            # the main thread never left the default
    # one override table per mode: an entry registered under fp16 is not the fp32 entry of the same geometry
    d = _lib.ConvDesc(2, 7, 7, 32, 7, 7, 32, 1, 1, 1, 1, 1, 1, 0, 0, 32, 32)
    c, sp = ctypes.c_int(0), ctypes.c_int(0)
    lib.dj_set_thread_compute_mode(1)
    _lib.check(lib.dj_conv2d_tune_set(0, d, 2, 1), "tune_set")
    lib.dj_set_thread_compute_mode(-1)
    for direction in (0, 1, 2):
        for dd in (d, _lib.ConvDesc(4, 10, 10, 64, 10, 10, 64, 3, 3, 1, 1, 1, 1, 1, 1, 64, 64)):
            for mode in (0, 1, 2):
                lib.dj_set_thread_compute_mode(mode)
                _lib.check(lib.dj_conv2d_tune_set(direction, dd, -1, 1), "tune_set")
    lib.dj_set_thread_compute_mode(-1)

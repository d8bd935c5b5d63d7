"""DecodeDetections on the GPU (dj_decode_detections through the C ABI) against the reference-generated fixture and
against the host numpy decoder; plus mode='inference' of the SSD builder end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "decode.npz")


def canon(rows):
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 6)
    rows = rows[rows[:, 1] > 0]
    return rows[np.lexsort((rows[:, 2], rows[:, 0], -rows[:, 1]))]


def run_decode(y_pred, thresh, iou_t, top_k, nms_max=400):
    from jpeg_detection_resnet_ssd_amd.engine import call, query
    y = torch.from_numpy(y_pred).cuda().contiguous()
    b, n, w = y.shape
    ws = torch.empty(query("dj_decode_detections_workspace_floats", b, n, w - 12, nms_max), device="cuda")
    out = torch.full((b, top_k, 6), -7.0, device="cuda")
    call("dj_decode_detections", y, b, n, w - 12, thresh, iou_t, top_k, nms_max, 1, 300, 300, ws, out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("thresh,top_k,keys", [(0.3, 200, ("d0", "d1")), (0.05, 50, ("e0", "e1"))])
def test_decode_against_reference_fixture(thresh, top_k, keys):
    g = np.load(GOLD)
    out = run_decode(g["y_pred"], thresh, 0.45, top_k)
    for b, key in enumerate(keys):
        got, ref = canon(out[b]), canon(g[key])
        assert got.shape == ref.shape
        np.testing.assert_array_equal(got[:, 0], ref[:, 0])
        np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=1e-6)
        np.testing.assert_allclose(got[:, 2:], ref[:, 2:], atol=2e-2)      # pixels on a 300x300 canvas, f32 exp
        conf = out[b][:, 1]
        assert np.all(conf[:-1] >= conf[1:])                                # sorted like tf.nn.top_k


def test_decode_padding_and_ragged():
    from jpeg_detection_resnet_ssd_amd.ssd_encoder_decoder.ssd_output_decoder import decode_detections
    g = np.load(GOLD)
    y = g["y_pred"].copy()
    y[1, :, 1:21] = 0.0                                                     # image 1: nothing above threshold
    y[1, :, 0] = 1.0
    out = run_decode(y, 0.6, 0.45, 200)
    ref = decode_detections(y.astype(np.float64), confidence_thresh=0.6, iou_threshold=0.45, top_k=200,
                            img_height=300, img_width=300)
    assert np.all(out[1] == 0.0)
    got = canon(out[0])
    assert got.shape == canon(ref[0]).shape and got.shape[0] < 200
    np.testing.assert_allclose(got[:, :2], canon(ref[0])[:, :2], rtol=1e-6)
    assert np.all(out[0][got.shape[0]:] == 0.0)


def test_decode_nms_cap_and_errors():
    from jpeg_detection_resnet_ssd_amd._lib import DjError
    g = np.load(GOLD)
    out = run_decode(g["y_pred"], 0.01, 0.45, 30, nms_max=3)               # at most 3 per class
    for b in range(2):
        cls = out[b][out[b][:, 1] > 0][:, 0]
        assert np.bincount(cls.astype(int)).max() <= 3
    with pytest.raises(DjError):
        run_decode(g["y_pred"], 0.01, 0.45, 30, nms_max=1000)              # 20 * 1000 > 8192 merge capacity


def test_inference_mode_model():
    """mode='inference' graph: training-mode weights shared, output (B, top_k, 6) == host decode of mode='training'."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.models.keras_ssd300_dct_j2d_resnet import ssd_resnet_EF_layers_custom
    from jpeg_detection_resnet_ssd_amd.ssd_encoder_decoder.ssd_output_decoder import decode_detections
    train, sizes = workloads.build_ssd("ssd_custom", compile_model=False)
    weights = train.get_weights_dict()
    for k in weights:                       # random init -> tame box offsets so that exp() stays finite
        if "mbox_loc" in k:
            weights[k] = weights[k] * 1e-5
        if "mbox_conf" in k:                # ... and keep the softmax away from saturated (tied) confidences
            weights[k] = weights[k] * 1e-5
    train.set_weights_dict(weights)
    x = workloads.synthetic_batch("ssd_custom", sizes, 2, seed=3)[0]
    raw = train.predict(x, batch_size=2)
    assert raw[..., 1:21].max() < 0.5
    K.clear_session()                       # auto layer names (batch_normalization_N) restart, as in Keras
    K.set_random_seed(7)
    args = dict(workloads.SSD_ARGS, mode="inference", confidence_thresh=0.05, top_k=20)
    infer = ssd_resnet_EF_layers_custom(archi="ssd_custom", **args)
    assert infer.set_weights_dict(weights) == len(weights)
    det = infer.predict(x, batch_size=2)
    assert det.shape == (2, 20, 6)
    ref = decode_detections(raw.astype(np.float64), confidence_thresh=0.05, iou_threshold=0.45, top_k=20,
                            img_height=300, img_width=300)
    for b in range(2):
        got, want = canon(det[b]), canon(ref[b])
        assert got.shape == want.shape
        np.testing.assert_array_equal(got[:, 0], want[:, 0])
        np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=1e-5)
        np.testing.assert_allclose(got[:, 2:], want[:, 2:], atol=5e-2)


def test_evaluator_runs_an_inference_model_end_to_end():
    """Evaluator + mode='inference' model (on-device DecodeDetections) over a small synthetic set; with mode='training'
    + host decode_detections the per-class prediction lists are the same."""
    from jpeg_detection_resnet_ssd_amd import workloads
    from jpeg_detection_resnet_ssd_amd.data.generators import SyntheticDataGeneratorDCT
    from jpeg_detection_resnet_ssd_amd.eval_utils.average_precision_evaluator import Evaluator
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.models.keras_ssd300_dct_j2d_resnet import ssd_resnet_EF_layers_custom
    train, sizes = workloads.build_ssd("ssd_custom", compile_model=False)
    weights = train.get_weights_dict()
    for k in weights:
        if "mbox_loc" in k or "mbox_conf" in k:
            weights[k] = weights[k] * 1e-5
    train.set_weights_dict(weights)
    K.clear_session()
    infer = ssd_resnet_EF_layers_custom(archi="ssd_custom", **dict(workloads.SSD_ARGS, mode="inference",
                                                                  confidence_thresh=0.05, top_k=20))
    infer.set_weights_dict(weights)
    data = SyntheticDataGeneratorDCT(n_images=6, seed=11)
    ev_i = Evaluator(infer, 20, data, model_mode="inference")
    # border_pixels also steers the host decoder's NMS ('include' adds 1 px to the areas, TF's NMS in the layer does not)
    m_i = ev_i(300, 300, batch_size=4, verbose=False, border_pixels="half")
    ev_t = Evaluator(train, 20, data, model_mode="training")
    m_t = ev_t(300, 300, batch_size=4, verbose=False, decoding_confidence_thresh=0.05, decoding_top_k=20,
               border_pixels="half")
    assert 0.0 <= m_i <= 1.0 and abs(m_i - m_t) < 1e-9
    for c in range(1, 21):
        a = sorted((p[0], round(float(p[1]), 5)) for p in ev_i.prediction_results[c])
        b = sorted((p[0], round(float(p[1]), 5)) for p in ev_t.prediction_results[c])
        assert a == b
    assert sum(len(r) for r in ev_i.prediction_results) == 6 * 20


def test_decode_fast_against_reference_fixture():
    """DecodeDetectionsFast kernels (arg-max class, one class-agnostic NMS) vs the reference's decode_detections_fast."""
    from jpeg_detection_resnet_ssd_amd.engine import call, query
    g = np.load(GOLD)
    y = torch.from_numpy(g["y_pred"]).cuda().contiguous()
    b, n, w = y.shape
    ws = torch.empty(query("dj_decode_detections_fast_workspace_floats", b, n, 400), device="cuda")
    out = torch.full((b, 200, 6), -7.0, device="cuda")
    call("dj_decode_detections_fast", y, b, n, w - 12, 0.3, 0.45, 200, 400, 1, 300, 300, ws, out)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i, key in enumerate(("f0", "f1")):
        got, ref = canon(out[i]), canon(g[key])
        assert got.shape == ref.shape
        np.testing.assert_array_equal(got[:, 0], ref[:, 0])
        np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=1e-6)
        np.testing.assert_allclose(got[:, 2:], ref[:, 2:], atol=2e-2)
        assert np.all(out[i][:-1, 1] >= out[i][1:, 1])


def test_inference_fast_mode_model():
    from jpeg_detection_resnet_ssd_amd import workloads
    from jpeg_detection_resnet_ssd_amd.keras import backend as K
    from jpeg_detection_resnet_ssd_amd.models.keras_ssd300_dct_j2d_resnet import ssd_resnet_EF_layers_identical
    K.clear_session()
    m = ssd_resnet_EF_layers_identical(archi="deconv", **dict(workloads.SSD_ARGS, mode="inference_fast", top_k=30))
    assert m.layers[-1].__class__.__name__ in ("DecodeDetections", "DecodeDetectionsFast") and m.layers[-1].fast
    w = m.get_weights_dict()
    for k in w:                                   # random init: keep exp() of the box offsets finite
        if "mbox_loc" in k or "mbox_conf" in k:
            w[k] = w[k] * 1e-5
    m.set_weights_dict(w)
    x = workloads.synthetic_batch("deconv", [(38, 38), (10, 10), (5, 5), (5, 5), (3, 3), (1, 1)], 2, seed=3)[0]
    det = m.predict(x, batch_size=2)
    assert det.shape == (2, 30, 6) and np.isfinite(det).all()

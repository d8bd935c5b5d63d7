"""JPEG -> DCT coefficient reader (csrc/dj_jpeg.cpp behind include/dj_jpeg.h, `jpeg2dct.numpy` surface) against
coefficients decoded by libjpeg 9's jpeg_read_coefficients (tests/golden/make_jpeg_fixtures.py), bit-exact."""
import io
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "jpeg_coefficients.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def j2d():
    from jpeg_detection_resnet_ssd_amd import _build
    _build.build_jpeg_library()
    from jpeg_detection_resnet_ssd_amd.jpeg2dct import numpy as j2d
    return j2d


def names(gold):
    return sorted({k.split("/")[0] for k in gold.files if k.endswith("/coef0")})


def test_all_fixture_cases_bit_exact(gold, j2d):
    assert len(names(gold)) >= 8
    for name in names(gold):
        data = gold[name + "/jpeg"].tobytes()
        n = sum(1 for k in gold.files if k.startswith(name + "/coef"))
        raw = j2d.loads(data, normalized=False)
        deq = j2d.loads(data, normalized=True)
        inf = j2d.info(data)
        assert inf.n_components == n
        for c in range(n):
            want, q = gold["%s/coef%d" % (name, c)], gold["%s/quant%d" % (name, c)]
            assert raw[c].dtype == np.int16 and raw[c].shape == want.shape, name
            np.testing.assert_array_equal(raw[c], want, err_msg=name)
            np.testing.assert_array_equal(deq[c], (want.astype(np.int32) * q.astype(np.int32)).astype(np.int16), err_msg=name)
            np.testing.assert_array_equal(np.array(inf.quant[c][:]), q)
        if n == 1:
            assert deq[1].shape == (0, 0, 64)


def test_emission_shapes_of_the_trainers(gold, j2d):
    """300x300 -> Y (38,38,64), Cb/Cr (19,19,64); 224x224 -> (28,28,64) / (14,14,64)
    (object_detection_2d_data_generator_dct_j2d.py:1167-1195, generators.py:120-130); quality-75 tables as in the
    reference's known answer (luma row 8,6,5,8,12)."""
    y, cb, cr = j2d.loads(gold["ssd300_default/jpeg"].tobytes())
    assert y.shape == (38, 38, 64) and cb.shape == (19, 19, 64) and cr.shape == (19, 19, 64)
    y, cb, cr = j2d.loads(gold["cls224_default/jpeg"].tobytes())
    assert y.shape == (28, 28, 64) and cb.shape == (14, 14, 64)
    inf = j2d.info(gold["cls224_default/jpeg"].tobytes())
    assert list(inf.quant[0][:5]) == [8, 6, 5, 8, 12]
    assert (inf.h_samp[0], inf.v_samp[0], inf.h_samp[1], inf.v_samp[1]) == (2, 2, 1, 1)


def test_batch_decode_matches_single(gold, j2d):
    data = gold["ssd300_default/jpeg"].tobytes()
    y, cb, cr = j2d.decode_batch([data] * 5, (38, 38), (19, 19), n_threads=3)
    ry, rcb, rcr = j2d.loads(data)
    assert y.dtype == np.float32 and y.shape == (5, 38, 38, 64)
    for i in range(5):
        np.testing.assert_array_equal(y[i], ry.astype(np.float32))
        np.testing.assert_array_equal(cb[i], rcb.astype(np.float32))
        np.testing.assert_array_equal(cr[i], rcr.astype(np.float32))
    with pytest.raises(ValueError):
        j2d.decode_batch([data, gold["cls224_default/jpeg"].tobytes()], (38, 38), (19, 19))


def test_errors(gold, j2d):
    with pytest.raises(ValueError):
        j2d.loads(gold["progressive/jpeg"].tobytes())
    with pytest.raises(ValueError):
        j2d.loads(b"not a jpeg at all")
    data = gold["ssd300_default/jpeg"].tobytes()
    with pytest.raises(ValueError):
        j2d.loads(data[:200])                      # truncated inside the headers
    j2d.loads(data[:len(data) // 2])               # truncated scan: zero-filled tail like libjpeg, must not crash


def _hostile_inputs(gold):
    """Crafted files (name -> bytes) that must be REJECTED cleanly (a second SOF after a scan had sized the
    coefficient planes made the next scan write past them)."""
    odd, big = gold["odd_53x37_q90/jpeg"].tobytes(), gold["ssd300_default/jpeg"].tobytes()
    assert odd[-2:] == b"\xff\xd9" and big[:2] == b"\xff\xd8"
    sos = big.index(b"\xff\xda")
    sof = big.index(b"\xff\xc0")
    out = {
        "two_frames": odd[:-2] + big[2:],                                   # small frame, scan, LARGER frame, scan
        "two_frames_small_second": big[:-2] + odd[2:],
        "sos_len2_at_eof": big[:sos] + b"\xff\xda\x00\x02",                # SOS with an empty body as the last bytes
        "sos_len3": big[:sos] + b"\xff\xda\x00\x03\x03",
        "huge_4x4": big[:sof + 5] + b"\xff\xff\xff\xff" + big[sof + 9:sof + 11] + b"\x44" + big[sof + 12:],
        "sof_zero_components": big[:sof + 9] + b"\x00" + big[sof + 10:],
        "garbage_after_soi": b"\xff\xd8" + bytes(range(256)) * 8,
        "scan_without_tables": big[:2] + big[sof:],
    }
    rng = np.random.default_rng(5)
    for i in range(6):                                                       # random byte flips inside the headers
        b = bytearray(big)
        for pos in rng.integers(2, sos + 12, size=4):
            b[int(pos)] = int(rng.integers(0, 256))
        out["flip_%d" % i] = bytes(b)
    return out


def test_hostile_files_are_rejected(gold, j2d):
    bad = _hostile_inputs(gold)
    for name in ("two_frames", "two_frames_small_second", "sos_len2_at_eof", "sos_len3", "huge_4x4",
                 "sof_zero_components", "garbage_after_soi", "scan_without_tables"):
        with pytest.raises(ValueError):
            j2d.loads(bad[name])
    for name in [k for k in bad if k.startswith("flip_")]:                   # may decode or fail, must not crash
        try:
            j2d.loads(bad[name])
        except ValueError:
            pass
    with pytest.raises(ValueError):
        j2d.decode_batch([bad["two_frames"]] * 2, (7, 5), (4, 3))


def test_hostile_files_under_address_sanitizer(gold, tmp_path):
    """The same files through an AddressSanitizer + UBSan build of dj_jpeg.cpp (CPU build only; harness:
    tests/jpeg_asan_harness.cpp): no report, every crafted file answered with rc = -1, good files with rc = 0."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "jpeg_asan")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(root, "tests", "jpeg_asan_harness.cpp"),
           os.path.join(root, "jpeg_detection_resnet_ssd_amd", "csrc", "dj_jpeg.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitizer" in (r.stderr or "").lower():
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    files, must_fail = [], set()
    for name, data in _hostile_inputs(gold).items():
        path = tmp_path / (name + ".jpg")
        path.write_bytes(data)
        files.append(str(path))
        if not name.startswith("flip_"):
            must_fail.add(str(path))
    good = []
    for name in names(gold):
        if name == "progressive":
            continue
        path = tmp_path / ("ok_" + name + ".jpg")
        path.write_bytes(gold[name + "/jpeg"].tobytes())
        good.append(str(path))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + files + good, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        (r.stdout[-1500:], r.stderr[-3000:])
    lines = {ln.split(": ", 1)[0]: ln for ln in r.stdout.splitlines() if ": info=" in ln}
    for f in must_fail:
        assert " rc=-1" in lines[f], lines[f]
    for f in good:
        assert " rc=0" in lines[f], lines[f]


def test_pil_round_trip_if_available(j2d):
    """Coefficients -> dequantise -> IDCT reproduces PIL's own decode of the same file (sanity of the whole chain)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    img = np.kron(rng.uniform(40, 200, (8, 8)), np.ones((8, 8))).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="jpeg", quality=95)
    (y, _, _) = j2d.loads(buf.getvalue(), normalized=True)
    k = np.arange(8)
    basis = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k[:, None] == 0, np.sqrt(1 / 8), np.sqrt(2 / 8))
    rec = np.einsum("ux,hwuv,vy->hxwy", basis, y.reshape(8, 8, 8, 8).astype(float), basis).reshape(64, 64) + 128
    dec = np.asarray(Image.open(io.BytesIO(buf.getvalue())), dtype=float)
    assert np.abs(rec - dec).max() <= 2.0


def test_emission_step_of_the_generators(j2d):
    """emit_dct_inputs == the generator's per-image loop: PIL save -> loads -> X_y / X_cbcr (or X_cb, X_cr)."""
    pytest.importorskip("PIL.Image")
    from jpeg_detection_resnet_ssd_amd.data.jpeg_dct import blocks_for, emit_dct_inputs, rgb_to_jpeg_bytes
    rng = np.random.default_rng(11)
    batch = np.clip(np.kron(rng.uniform(0, 255, (3, 50, 50, 3)), np.ones((1, 6, 6, 1))) + rng.normal(0, 9, (3, 300, 300, 3)),
                    0, 255).astype(np.uint8)
    assert blocks_for(300, 300) == ((38, 38), (19, 19)) and blocks_for(224, 224) == ((28, 28), (14, 14))
    X_y, X_cbcr = emit_dct_inputs(batch, deconv=False, n_threads=2)
    X_y2, X_cb, X_cr = emit_dct_inputs(batch, deconv=True, n_threads=2)
    assert X_y.shape == (3, 38, 38, 64) and X_cbcr.shape == (3, 19, 19, 128) and X_cb.shape == (3, 19, 19, 64)
    for i in range(3):
        dct_y, dct_cb, dct_cr = j2d.loads(rgb_to_jpeg_bytes(batch[i]))
        np.testing.assert_array_equal(X_y[i], dct_y)
        np.testing.assert_array_equal(X_cbcr[i], np.concatenate([dct_cb, dct_cr], axis=-1))
        np.testing.assert_array_equal(X_cr[i], dct_cr)
    np.testing.assert_array_equal(X_y, X_y2)


def test_jpeg_library_exports_every_declared_symbol(j2d):
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "dj_jpeg.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(dj_jpeg_[a-z0-9_]+)\s*\(", header))
    assert declared == {"dj_jpeg_last_error", "dj_jpeg_read_info", "dj_jpeg_read_coefficients", "dj_jpeg_decode_batch_f32"}
    lib = ctypes.CDLL(os.path.join(root, "jpeg_detection_resnet_ssd_amd", "csrc", "libdj_jpeg.so"))
    for name in declared:
        assert hasattr(lib, name)

"""Known answers for the JPEG coefficient reader (jpeg2dct stand-in).  Runs in the BUILD CONTAINER only:
PIL writes synthetic JPEGs (the reference's emission path: Image.fromarray(x).save(f, format="jpeg"),
object_detection_2d_data_generator_dct_j2d.py:1176-1178), the image's libjpeg 9 (jpeg_read_coefficients, what jpeg2dct
wraps) decodes them through tests/golden/jpeg_coef_dump.c, and JPEG bytes + coefficients are stored in
tests/golden/jpeg_coefficients.npz.

    gcc -I/opt/conda/include tests/golden/jpeg_coef_dump.c -L/opt/conda/lib -ljpeg -Wl,-rpath,/opt/conda/lib -o /tmp/jpeg_coef_dump
    python tests/golden/make_jpeg_fixtures.py /tmp/jpeg_coef_dump
"""
import io
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))


def smooth_image(h, w, seed, channels=3):
    rng = np.random.default_rng(seed)
    small = rng.uniform(0, 255, (h // 6 + 2, w // 6 + 2, channels))
    img = np.kron(small, np.ones((6, 6, 1)))[:h, :w]
    img += rng.normal(0, 12, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 0] if channels == 1 else img


def dump(dumper, jpeg_bytes):
    with tempfile.NamedTemporaryFile(suffix=".jpg") as f:
        f.write(jpeg_bytes)
        f.flush()
        raw = subprocess.run([dumper, f.name], check=True, capture_output=True).stdout
    pos = 0
    n = int(np.frombuffer(raw, np.int32, 1, pos)[0]); pos += 4
    planes, quants = [], []
    for _ in range(n):
        bh, bw = np.frombuffer(raw, np.int32, 2, pos); pos += 8
        q = np.frombuffer(raw, np.int32, 64, pos).copy(); pos += 256
        c = np.frombuffer(raw, np.int16, int(bh) * int(bw) * 64, pos).reshape(int(bh), int(bw), 64).copy()
        pos += c.size * 2
        planes.append(c); quants.append(q)
    assert pos == len(raw)
    return planes, quants


def main(dumper):
    cases = {
        # name: (image, PIL save kwargs)
        "ssd300_default": (smooth_image(300, 300, 1), {}),                       # the SSD trainer's emission
        "cls224_default": (smooth_image(224, 224, 2), {}),                       # the classifier's
        "odd_53x37_q90": (smooth_image(37, 53, 3), {"quality": 90}),
        "s444_q50": (smooth_image(64, 80, 4), {"quality": 50, "subsampling": 0}),
        "s422_optimize": (smooth_image(48, 72, 5), {"subsampling": 1, "optimize": True}),
        "gray_100x60": (smooth_image(60, 100, 6, channels=1), {}),
        "restart_rows": (smooth_image(96, 128, 7), {"restart_marker_rows": 1}),
        "restart_blocks": (smooth_image(80, 80, 8), {"restart_marker_blocks": 3, "quality": 30}),
        "noise_q100": (np.random.default_rng(9).integers(0, 256, (40, 40, 3), dtype=np.uint8), {"quality": 100}),
    }
    store = {}
    for name, (img, kw) in cases.items():
        buf = io.BytesIO()
        try:
            Image.fromarray(img).save(buf, format="jpeg", **kw)
        except TypeError as e:
            print("skip", name, e)
            continue
        data = buf.getvalue()
        planes, quants = dump(dumper, data)
        store[name + "/jpeg"] = np.frombuffer(data, np.uint8)
        for c, (p, q) in enumerate(zip(planes, quants)):
            store["%s/coef%d" % (name, c)] = p
            store["%s/quant%d" % (name, c)] = q.astype(np.int16)
        print(name, len(data), "bytes", [p.shape for p in planes])
    buf = io.BytesIO()
    Image.fromarray(smooth_image(64, 64, 10)).save(buf, format="jpeg", progressive=True)
    store["progressive/jpeg"] = np.frombuffer(buf.getvalue(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "jpeg_coefficients.npz"), **store)
    print("wrote", os.path.join(OUT, "jpeg_coefficients.npz"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/jpeg_coef_dump")

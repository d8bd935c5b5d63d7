"""Synthetic stand-ins for the reference's data generators (host side, numpy): JPEG-DCT coefficient
batches shaped and distributed like `jpeg2dct.numpy.loads` output for PIL-saved JPEGs (quality 75,
4:2:0), and random ground-truth boxes encoded by SSDInputEncoder.

Emission contract being mimicked: localisation_part/data_generator/object_detection_2d_data_generator_dct_j2d.py
:1167-1195 (Y (38,38,64), CbCr = concat(cb, cr) (19,19,128), or Cb/Cr separately for the deconv archi);
coefficients are DE-QUANTISED (value = quantised level x table entry) in natural raster order, as pinned
by classification_part/vgg_jpeg_keras/tests/generators/tests_generators.py:66-68
(-616 = -77 x 8 with the libjpeg quality-75 luma table)."""
import numpy as np
from scipy.fft import dctn
from scipy.ndimage import uniform_filter

# Annex-K base tables; libjpeg quality 75 -> scale 50 -> (base*50 + 50) // 100
_LUMA_BASE = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]).reshape(8, 8)
_CHROMA_BASE = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32).reshape(8, 8)


def quant_table(base, quality=75):
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    return np.clip((base * scale + 50) // 100, 1, 255).astype(np.float64)


LUMA_Q75 = quant_table(_LUMA_BASE)
CHROMA_Q75 = quant_table(_CHROMA_BASE)


def _blocks_dct(plane, table):
    """plane (H, W) float (level-shifted), H and W multiples of 8 -> (H/8, W/8, 64) de-quantised coefficients."""
    h, w = plane.shape
    blk = plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)
    coef = dctn(blk, axes=(2, 3), norm="ortho")
    q = np.round(coef / table) * table
    return q.reshape(h // 8, w // 8, 64)


def _pad_edge(plane, mult):
    h, w = plane.shape
    ph, pw = (-h) % mult, (-w) % mult
    return np.pad(plane, ((0, ph), (0, pw)), mode="edge")


def rgb_to_dct(img):
    """uint8 RGB (H, W, 3) -> (Y (H16/8, W16/8, 64), Cb, Cr (H16/16, W16/16, 64)) as jpeg2dct would return
    them for a quality-75 4:2:0 JPEG of `img` (JFIF BT.601 full-range YCbCr, 2x2 box chroma subsampling)."""
    rgb = img.astype(np.float64)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = -0.168735892 * r - 0.331264108 * g + 0.5 * b + 128.0
    cr = 0.5 * r - 0.418687589 * g - 0.081312411 * b + 128.0
    y, cb, cr = (np.clip(np.round(p), 0, 255) for p in (y, cb, cr))
    y, cb, cr = (_pad_edge(p, 16) for p in (y, cb, cr))

    def sub(p):
        return np.round(p.reshape(p.shape[0] // 2, 2, p.shape[1] // 2, 2).mean(axis=(1, 3)))
    return (_blocks_dct(y - 128.0, LUMA_Q75), _blocks_dct(sub(cb) - 128.0, CHROMA_Q75),
            _blocks_dct(sub(cr) - 128.0, CHROMA_Q75))


def smooth_random_image(rng, size=300):
    """Low-pass-filtered noise with photo-like spectral decay plus a few flat rectangles."""
    img = rng.uniform(0, 255, size=(size, size, 3))
    coarse = uniform_filter(rng.uniform(0, 255, size=(size, size, 3)), size=(31, 31, 1), mode="reflect")
    mid = uniform_filter(img, size=(7, 7, 1), mode="reflect")
    out = 0.65 * (coarse - 127.5) * 4.0 + 0.3 * (mid - 127.5) * 2.0 + 0.05 * (img - 127.5) + 127.5
    for _ in range(int(rng.integers(1, 4))):
        x0, y0 = rng.integers(0, size - 40, size=2)
        w, h = rng.integers(20, 120, size=2)
        out[y0:y0 + h, x0:x0 + w] = 0.5 * out[y0:y0 + h, x0:x0 + w] + 0.5 * rng.uniform(0, 255, size=3)
    return np.clip(out, 0, 255).astype(np.uint8)


def dct_batch(batch, seed=1234, size=300, split_chroma=False, dtype=np.float32):
    """-> [Y (B,g,g,64), CbCr (B,g/2,g/2,128)] or [Y, Cb, Cr] (split_chroma, the `deconv` archi)."""
    rng = np.random.default_rng(seed)
    ys, cbs, crs = [], [], []
    for _ in range(batch):
        y, cb, cr = rgb_to_dct(smooth_random_image(rng, size))
        ys.append(y)
        cbs.append(cb)
        crs.append(cr)
    y, cb, cr = (np.stack(a).astype(dtype) for a in (ys, cbs, crs))
    if split_chroma:
        return [y, cb, cr]
    return [y, np.concatenate([cb, cr], axis=-1)]


def fast_dct_batch(batch, seed=1234, grid=38, split_chroma=False, dtype=np.float32):
    """Throughput-only stand-in: Laplace-distributed quantised levels x table, same shapes / sparsity
    pattern class as dct_batch but ~100x cheaper to generate."""
    rng = np.random.default_rng(seed)
    decay = np.exp(-0.45 * (np.arange(8)[:, None] + np.arange(8)[None, :])).reshape(64)

    def plane(g, table, dc_scale):
        lv = rng.laplace(0.0, 6.0, size=(batch, g, g, 64)) * decay
        lv[..., 0] = rng.normal(0.0, dc_scale, size=(batch, g, g))
        return (np.round(lv) * table.reshape(64)).astype(dtype)
    y = plane(grid, LUMA_Q75, 40.0)
    cb = plane(grid // 2, CHROMA_Q75, 10.0)
    cr = plane(grid // 2, CHROMA_Q75, 10.0)
    if split_chroma:
        return [y, cb, cr]
    return [y, np.concatenate([cb, cr], axis=-1)]


def random_ground_truth(batch, seed=1234, size=300, n_classes=20, max_boxes=6, min_side=20):
    """List of (k, 5) arrays (class_id, xmin, ymin, xmax, ymax), 1..max_boxes boxes per image."""
    rng = np.random.default_rng(seed + 7919)
    out = []
    for _ in range(batch):
        k = int(rng.integers(1, max_boxes + 1))
        rows = []
        for _ in range(k):
            w = int(rng.integers(min_side, size // 2))
            h = int(rng.integers(min_side, size // 2))
            x0 = int(rng.integers(0, size - w))
            y0 = int(rng.integers(0, size - h))
            rows.append([int(rng.integers(1, n_classes + 1)), x0, y0, x0 + w, y0 + h])
        out.append(np.array(rows, dtype=np.float64))
    return out

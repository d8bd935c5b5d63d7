"""RGB batch -> JPEG -> de-quantised DCT coefficient tensors: the emission step at the end of the reference's
generators (localisation_part/data_generator/object_detection_2d_data_generator_dct_j2d.py:1167-1195;
classification_part/vgg_jpeg_keras/generators/generators.py:120-130,179-187,337-346), with the in-tree coefficient
reader in place of jpeg2dct and the whole batch decoded by host threads into float32 tensors ready for upload."""
import io

import numpy as np

from ..jpeg2dct import numpy as j2d


def rgb_to_jpeg_bytes(image, **save_kwargs):
    """`Image.fromarray(image).save(fake_file, format="jpeg")` (PIL defaults: quality 75, 4:2:0)."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.asarray(image, dtype=np.uint8)).save(buf, format="jpeg", **save_kwargs)
    return buf.getvalue()


def blocks_for(height, width):
    """Block grids of a 4:2:0 JPEG: Y ceil(h/8) x ceil(w/8), chroma ceil(ceil(h/2)/8) x ceil(ceil(w/2)/8)."""
    return ((-(-height // 8), -(-width // 8)), (-(-(-(-height // 2)) // 8), -(-(-(-width // 2)) // 8)))


def emit_dct_inputs(batch_X, deconv=False, n_threads=None, jpeg_bytes=None, **save_kwargs):
    """batch_X: (B, H, W, 3) uint8 (or a list of equally sized images).  Returns `[X_y, X_cbcr]`, or
    `[X_y, X_cb, X_cr]` when deconv=True -- float32, 300x300 -> (B,38,38,64) / (B,19,19,128 | 64).
    `jpeg_bytes` may pass already-encoded JPEGs instead of pixels."""
    if jpeg_bytes is None:
        jpeg_bytes = [rgb_to_jpeg_bytes(img, **save_kwargs) for img in batch_X]
    inf = j2d.info(jpeg_bytes[0])
    y_blocks, c_blocks = (inf.blocks_h[0], inf.blocks_w[0]), (inf.blocks_h[1], inf.blocks_w[1])
    y, cb, cr = j2d.decode_batch(jpeg_bytes, y_blocks, c_blocks, normalized=True, n_threads=n_threads)
    if deconv:
        return [y, cb, cr]
    return [y, np.concatenate([cb, cr], axis=-1)]

/* dj_jpeg.h -- C ABI of the JPEG -> DCT-coefficient reader (host code, libdj_jpeg.so).
 *
 * Stands in for the third-party `jpeg2dct` extension the reference's generators call
 * (`from jpeg2dct.numpy import load, loads`):
 *   localisation_part/data_generator/object_detection_2d_data_generator_dct_j2d.py:1167-1195
 *   classification_part/vgg_jpeg_keras/generators/generators.py:120-130,179-187,337-346
 * jpeg2dct (un-vendored submodule, unpinned in C/Pipfile:101) wraps libjpeg's jpeg_read_coefficients(): entropy
 * decoding only, no inverse DCT.  Output per component: [blocks_h][blocks_w][64] int16 in NATURAL (row-major 8x8)
 * coefficient order, blocks_w = ceil(ceil(width * h_samp / h_max) / 8) (libjpeg's width_in_blocks, not the MCU-padded
 * count); normalized != 0 multiplies by the quantisation table (jpeg2dct's default), which is what the reference's
 * known answer pins (classification_part/vgg_jpeg_keras/tests/generators/tests_generators.py:66-68).
 *
 * Supported: baseline / extended sequential Huffman JPEG (SOF0, SOF1), 8-bit, 1-4 components, any sampling factors,
 * interleaved or per-component scans, restart intervals.  Progressive and arithmetic-coded files return an error.
 * All functions return 0 or a negative code; text via dj_jpeg_last_error() (thread-local).  Thread-safe. */
#ifndef DJ_JPEG_H
#define DJ_JPEG_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dj_jpeg_info {
  int width, height, n_components;
  int h_samp[4], v_samp[4];
  int blocks_w[4], blocks_h[4]; /* logical block grid of each component (what is returned) */
  int quant[4][64];             /* quantisation table of each component, natural order */
  int sof;                      /* 0 baseline, 1 extended sequential, 2 progressive, ... */
} dj_jpeg_info;

const char* dj_jpeg_last_error(void);

/* Parse the headers up to the first scan. */
int dj_jpeg_read_info(const unsigned char* data, long size, dj_jpeg_info* info);

/* Decode all coefficients.  planes[c] receives blocks_h[c] * blocks_w[c] * 64 int16 values (caller-allocated from
 * dj_jpeg_read_info's geometry); plane_capacity[c] = number of int16 the caller allocated (checked). */
int dj_jpeg_read_coefficients(const unsigned char* data, long size, int normalized, short* const* planes,
                              const long* plane_capacity, dj_jpeg_info* info);

/* A batch of n three-component 4:2:0 (or any layout with matching block grids) JPEGs decoded by n_threads host threads
 * straight into float32 batch tensors ready for upload: y [n][yh][yw][64], cb / cr [n][ch][cw][64] -- the arrays the
 * reference's generators fill one image at a time.  Images whose block grid differs from (yh, yw, ch, cw) fail. */
int dj_jpeg_decode_batch_f32(const unsigned char* const* data, const long* sizes, int n, int normalized, float* y,
                             float* cb, float* cr, int yh, int yw, int ch, int cw, int n_threads);

#ifdef __cplusplus
}
#endif
#endif

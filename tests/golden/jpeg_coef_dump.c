/* Known-answer generator for the JPEG coefficient reader (runs in the build container only, against the image's
 * libjpeg 9): jpeg_read_coefficients() -> for every component blocks_h, blocks_w and the quantised coefficients in
 * natural order, as text-free binary on stdout:  int32 n_components, then per component int32 blocks_h, int32
 * blocks_w, int32 quant[64], int16 coef[blocks_h*blocks_w*64].  jpeg2dct does exactly this and multiplies by quant. */
#include <stdio.h>
#include <stdlib.h>
#include <jpeglib.h>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  struct jpeg_decompress_struct cinfo;
  struct jpeg_error_mgr jerr;
  cinfo.err = jpeg_std_error(&jerr);
  jpeg_create_decompress(&cinfo);
  jpeg_stdio_src(&cinfo, f);
  jpeg_read_header(&cinfo, TRUE);
  jvirt_barray_ptr* coefs = jpeg_read_coefficients(&cinfo);
  int n = cinfo.num_components;
  fwrite(&n, 4, 1, stdout);
  for (int c = 0; c < n; ++c) {
    jpeg_component_info* ci = &cinfo.comp_info[c];
    int bh = ci->height_in_blocks, bw = ci->width_in_blocks;
    fwrite(&bh, 4, 1, stdout);
    fwrite(&bw, 4, 1, stdout);
    for (int k = 0; k < 64; ++k) {
      int q = ci->quant_table->quantval[k];
      fwrite(&q, 4, 1, stdout);
    }
    for (int r = 0; r < bh; ++r) {
      JBLOCKARRAY rows = (cinfo.mem->access_virt_barray)((j_common_ptr)&cinfo, coefs[c], r, 1, FALSE);
      for (int b = 0; b < bw; ++b) fwrite(rows[0][b], 2, 64, stdout);
    }
  }
  jpeg_finish_decompress(&cinfo);
  jpeg_destroy_decompress(&cinfo);
  fclose(f);
  return 0;
}

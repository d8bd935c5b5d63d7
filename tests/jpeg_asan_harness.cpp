// Test harness (CPU only): feeds every file named on the command line to the three entry points of csrc/dj_jpeg.cpp.
// tests/test_jpeg_reader_cpu.py compiles it together with dj_jpeg.cpp under -fsanitize=address,undefined: a heap
// overflow on a crafted file aborts the process, a clean rejection prints "rc=-1".
#include "../include/dj_jpeg.h"
#include <cstdio>
#include <vector>

static std::vector<unsigned char> slurp(const char* path) {
  std::vector<unsigned char> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  unsigned char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    std::vector<unsigned char> d = slurp(argv[i]);
    if (d.empty()) {
      printf("%s: empty\n", argv[i]);
      continue;
    }
    dj_jpeg_info info;
    int rc_info = dj_jpeg_read_info(d.data(), (long)d.size(), &info);
    int rc = -1;
    if (rc_info == 0 && info.n_components >= 1 && info.n_components <= 4) {
      // planes sized from the FIRST frame header, as jpeg2dct/numpy.py does
      std::vector<std::vector<short>> planes(4);
      short* ptrs[4] = {nullptr, nullptr, nullptr, nullptr};
      long caps[4] = {0, 0, 0, 0};
      for (int c = 0; c < info.n_components; ++c) {
        long n = (long)info.blocks_h[c] * info.blocks_w[c] * 64;
        if (n > (64L << 20)) n = 0;   // the reader must refuse rather than be handed gigabytes
        planes[c].resize((size_t)n);
        ptrs[c] = planes[c].data();
        caps[c] = n;
      }
      rc = dj_jpeg_read_coefficients(d.data(), (long)d.size(), 1, ptrs, caps, nullptr);
      if (info.n_components == 3) {
        const int yh = info.blocks_h[0], yw = info.blocks_w[0], ch = info.blocks_h[1], cw = info.blocks_w[1];
        if ((long)yh * yw <= (1L << 16)) {
          std::vector<float> y((size_t)2 * yh * yw * 64), cb((size_t)2 * ch * cw * 64), cr((size_t)2 * ch * cw * 64);
          const unsigned char* datas[2] = {d.data(), d.data()};
          long sizes[2] = {(long)d.size(), (long)d.size()};
          int rb = dj_jpeg_decode_batch_f32(datas, sizes, 2, 1, y.data(), cb.data(), cr.data(), yh, yw, ch, cw, 2);
          if (rb != rc) printf("%s: batch rc %d differs from single rc %d\n", argv[i], rb, rc);
        }
      }
    }
    printf("%s: info=%d rc=%d%s%s\n", argv[i], rc_info, rc, rc ? " " : "", rc ? dj_jpeg_last_error() : "");
  }
  return 0;
}

// JPEG -> quantised / de-quantised DCT coefficients, entropy decoding only (ITU-T T.81 sequential Huffman mode).
// Host code behind include/dj_jpeg.h; takes the place of jpeg2dct (libjpeg's jpeg_read_coefficients) in the
// reference's data generators (object_detection_2d_data_generator_dct_j2d.py:1167-1195).
#include "../../include/dj_jpeg.h"

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[256] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

// zig-zag position -> natural (row-major) index, T.81 figure A.6
const unsigned char kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  // canonical decoding (T.81 F.2.2.3): for code length l, codes in [mincode[l], maxcode[l]] map to vals[valptr[l] + ...]
  int32_t maxcode[18];
  int32_t mincode[17];
  int valptr[17];
  unsigned char vals[256];
  // one-step table on the next 9 bits: (length << 8) | symbol, 0 = longer code
  uint16_t fast[512];

  int build(const unsigned char counts[16], const unsigned char* symbols, int n) {
    memcpy(vals, symbols, (size_t)n);
    memset(fast, 0, sizeof(fast));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
        if (l <= 9) {
          int lo = code << (9 - l), hi = lo + (1 << (9 - l));
          for (int f = lo; f < hi; ++f) fast[f] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      maxcode[l] = counts[l - 1] ? code - 1 : -1;
      if (code > (1 << l)) return fail("corrupt Huffman table");
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
    return 0;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int td = 0, ta = 0;
  int blocks_w = 0, blocks_h = 0;        // logical grid (ceil(component size / 8))
  int padded_w = 0, padded_h = 0;        // grid padded to whole MCUs of an interleaved scan
  int16_t* coef = nullptr;               // padded_h * padded_w * 64, natural order, quantised (lives in a Scratch)
  int pred = 0;
};

// Coefficient planes reused from image to image by one thread: a fresh 300 KB allocation per image is an mmap +
// page faults + munmap, which costs more than the entropy decoding and serialises threads on the address-space lock.
struct Scratch {
  std::vector<int16_t> plane[4];
};

struct BitReader {
  const unsigned char* p;
  const unsigned char* end;
  uint64_t acc = 0;
  int nbits = 0;
  int marker = 0;  // pending marker byte seen in the entropy-coded segment (0 = none)
  const unsigned char* marker_pos = nullptr;  // its 0xFF

  void fill() {
    while (nbits <= 56) {
      unsigned b = 0;
      if (!marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;                                // stuffed zero
          } else {
            // a marker: stop consuming, feed zeros (T.81 F.2.2.5 pads with 1-bits; decoders see the marker first)
            const unsigned char* q = p + 1;
            while (q < end && *q == 0xFF) ++q;     // fill bytes
            marker_pos = q - 1;
            marker = (q < end) ? *q : 0xD9;
            p = (q < end) ? q + 1 : end;
            b = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= (uint64_t)b << (56 - nbits);
      nbits += 8;
    }
  }
  inline unsigned peek(int n) { return (unsigned)(acc >> (64 - n)); }
  inline void skip(int n) {
    acc <<= n;
    nbits -= n;
  }
  inline int receive(int n) {
    if (n == 0) return 0;
    if (nbits < n) fill();
    unsigned v = peek(n);
    skip(n);
    return (int)v;
  }
  void reset() {
    acc = 0;
    nbits = 0;
  }
};

inline int extend(int v, int s) { return (v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

inline int decode_symbol(BitReader& br, const Huff& h) {
  if (br.nbits < 16) br.fill();
  unsigned look = br.peek(9);
  uint16_t f = h.fast[look];
  if (f) {
    br.skip(f >> 8);
    return f & 0xFF;
  }
  int code = (int)br.peek(10);
  for (int l = 10; l <= 16; ++l) {
    if (code <= h.maxcode[l] && h.maxcode[l] >= 0) {
      br.skip(l);
      return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    code = (int)br.peek(l + 1);
  }
  return -1;
}

// 16384 x 16384 pixels at 1x1 sampling = 4 Mi blocks = 512 MB of int16 coefficients per component
static const long kMaxBlocksPerPlane = 4L << 20;

struct Decoder {
  const unsigned char* data;
  long size;
  int width = 0, height = 0, ncomp = 0, sof = -1;
  int hmax = 1, vmax = 1;
  int restart_interval = 0;
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Component comp[4];
  bool saw_sof = false;
  Scratch* scratch = nullptr;

  static int be16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

  int parse_dqt(const unsigned char* p, int len) {
    while (len > 0) {
      int pq = p[0] >> 4, tq = p[0] & 15;
      if (tq > 3 || pq > 1) return fail("bad DQT");
      int need = 1 + 64 * (pq + 1);
      if (len < need) return fail("truncated DQT");
      for (int k = 0; k < 64; ++k) qt[tq][kNatural[k]] = pq ? (uint16_t)be16(p + 1 + 2 * k) : p[1 + k];
      qt_present[tq] = true;
      p += need;
      len -= need;
    }
    return 0;
  }

  int parse_dht(const unsigned char* p, int len) {
    while (len > 0) {
      if (len < 17) return fail("truncated DHT");
      int tc = p[0] >> 4, th = p[0] & 15;
      if (tc > 1 || th > 3) return fail("bad DHT");
      int n = 0;
      for (int i = 0; i < 16; ++i) n += p[1 + i];
      if (n > 256 || len < 17 + n) return fail("truncated DHT");
      Huff& h = tc ? ac[th] : dc[th];
      if (h.build(p + 1, p + 17, n)) return -1;
      p += 17 + n;
      len -= 17 + n;
    }
    return 0;
  }

  int parse_sof(const unsigned char* p, int len, int kind) {
    // one frame per file: a second SOF after a scan has sized the coefficient planes would make later scans write with
    // the new geometry into the old planes
    if (saw_sof) return fail("second frame header (SOF) in one file");
    if (len < 6) return fail("truncated SOF");
    if (p[0] != 8) return fail("only 8-bit JPEG is supported (precision %d)", p[0]);
    height = be16(p + 1);
    width = be16(p + 3);
    ncomp = p[5];
    if (ncomp < 1 || ncomp > 4 || len < 6 + 3 * ncomp) return fail("bad SOF component count");
    if (width <= 0 || height <= 0) return fail("bad image size");
    sof = kind;
    hmax = vmax = 1;
    for (int c = 0; c < ncomp; ++c) {
      comp[c].id = p[6 + 3 * c];
      comp[c].h = p[7 + 3 * c] >> 4;
      comp[c].v = p[7 + 3 * c] & 15;
      comp[c].tq = p[8 + 3 * c];
      if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) return fail("bad SOF");
      if (comp[c].h > hmax) hmax = comp[c].h;
      if (comp[c].v > vmax) vmax = comp[c].v;
    }
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    // bound the coefficient planes the scans will allocate (65535 x 65535 with 4x4 sampling would ask for 8.6 GB each)
    if ((long)mcux * hmax * (long)mcuy * vmax > kMaxBlocksPerPlane)
      return fail("image of %d x %d pixels exceeds the reader's limit of %ld blocks per component", width, height,
                  kMaxBlocksPerPlane);
    for (int c = 0; c < ncomp; ++c) {
      const int cw = (width * comp[c].h + hmax - 1) / hmax, chh = (height * comp[c].v + vmax - 1) / vmax;
      comp[c].blocks_w = (cw + 7) / 8;
      comp[c].blocks_h = (chh + 7) / 8;
      comp[c].padded_w = mcux * comp[c].h;
      comp[c].padded_h = mcuy * comp[c].v;
    }
    saw_sof = true;
    return 0;
  }

  int decode_block(BitReader& br, Component& c, int16_t* blk) {
    const Huff& hd = dc[c.td];
    const Huff& ha = ac[c.ta];
    int s = decode_symbol(br, hd);
    if (s < 0 || s > 11) return fail("corrupt DC code");
    int diff = s ? extend(br.receive(s), s) : 0;
    c.pred += diff;
    blk[0] = (int16_t)c.pred;
    for (int k = 1; k < 64;) {
      int rs = decode_symbol(br, ha);
      if (rs < 0) return fail("corrupt AC code");
      int r = rs >> 4;
      s = rs & 15;
      if (s == 0) {
        if (r != 15) break;   // end of block
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return fail("AC run past the block end");
      blk[kNatural[k]] = (int16_t)extend(br.receive(s), s);
      ++k;
    }
    return 0;
  }

  int decode_scan(const unsigned char* hdr, int len, const unsigned char* ecs, const unsigned char* end,
                  const unsigned char** next) {
    if (!saw_sof) return fail("SOS before SOF");
    if (len < 1) return fail("bad SOS");
    int ns = hdr[0];
    if (ns < 1 || ns > 4 || len < 1 + 2 * ns + 3) return fail("bad SOS");
    Component* sc[4];
    for (int i = 0; i < ns; ++i) {
      int id = hdr[1 + 2 * i], idx = -1;
      for (int c = 0; c < ncomp; ++c)
        if (comp[c].id == id) idx = c;
      if (idx < 0) return fail("SOS names an unknown component");
      sc[i] = &comp[idx];
      sc[i]->td = hdr[2 + 2 * i] >> 4;
      sc[i]->ta = hdr[2 + 2 * i] & 15;
      if (sc[i]->td > 3 || sc[i]->ta > 3 || !dc[sc[i]->td].present || !ac[sc[i]->ta].present)
        return fail("scan uses an undefined Huffman table");
      sc[i]->pred = 0;
    }
    const unsigned char* tail = hdr + 1 + 2 * ns;
    if (tail[0] != 0 || tail[1] != 63 || tail[2] != 0) return fail("not a sequential scan (Ss/Se/Ah/Al)");
    for (int c = 0; c < ncomp; ++c)
      if (!comp[c].coef) {
        const size_t need = (size_t)comp[c].padded_w * comp[c].padded_h * 64;
        if (scratch->plane[c].size() < need) scratch->plane[c].resize(need);
        comp[c].coef = scratch->plane[c].data();
        memset(comp[c].coef, 0, need * sizeof(int16_t));
      }

    BitReader br;
    br.p = ecs;
    br.end = end;
    int mcus_x, mcus_y;
    if (ns == 1) {  // non-interleaved: one block per MCU over the component's logical grid
      mcus_x = sc[0]->blocks_w;
      mcus_y = sc[0]->blocks_h;
    } else {
      mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
      mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
    }
    long count = 0;
    int expect_rst = 0;
    for (int my = 0; my < mcus_y; ++my)
      for (int mx = 0; mx < mcus_x; ++mx) {
        if (restart_interval && count && count % restart_interval == 0) {
          // byte-align, consume RSTn
          br.reset();
          if (!br.marker) br.fill(), br.reset();   // the marker follows the (already consumed) padding bits
          if (br.marker != 0xD0 + expect_rst) return fail("missing restart marker RST%d", expect_rst);
          br.marker = 0;
          expect_rst = (expect_rst + 1) & 7;
          for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
        }
        ++count;
        if (ns == 1) {
          Component& c = *sc[0];
          if (decode_block(br, c, c.coef + ((size_t)my * c.padded_w + mx) * 64)) return -1;
        } else {
          for (int i = 0; i < ns; ++i) {
            Component& c = *sc[i];
            for (int by = 0; by < c.v; ++by)
              for (int bx = 0; bx < c.h; ++bx)
                if (decode_block(br, c, c.coef + ((size_t)(my * c.v + by) * c.padded_w + (mx * c.h + bx)) * 64))
                  return -1;
          }
        }
      }
    // position after the scan: the pending marker (if the reader met one) or search for the next
    if (br.marker) {
      *next = br.marker_pos;
    } else {
      const unsigned char* q = br.p;
      while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
      *next = q;
    }
    return 0;
  }

  // headers_only: stop at the first SOS
  int run(bool headers_only) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG (no SOI)");
    const unsigned char* p = data + 2;
    const unsigned char* end = data + size;
    bool any_scan = false;
    while (p + 1 < end) {
      if (p[0] != 0xFF) {
        ++p;
        continue;
      }
      int m = p[1];
      if (m == 0xFF) {
        ++p;
        continue;
      }
      p += 2;
      if (m == 0xD9) break;                                  // EOI
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;   // standalone
      if (p + 2 > end) return fail("truncated marker segment");
      int len = be16(p);
      if (len < 2 || p + len > end) return fail("truncated marker segment");
      const unsigned char* body = p + 2;
      int blen = len - 2;
      if (m == 0xDB) {
        if (parse_dqt(body, blen)) return -1;
      } else if (m == 0xC4) {
        if (parse_dht(body, blen)) return -1;
      } else if (m == 0xC0 || m == 0xC1) {
        if (parse_sof(body, blen, m - 0xC0)) return -1;
      } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC) || m == 0xC3) {
        sof = m - 0xC0;
        if (headers_only && blen >= 6) {                      // still report the geometry
          int keep = sof;
          if (parse_sof(body, blen, keep)) return -1;
          return 0;
        }
        return fail("unsupported JPEG process SOF%d (only baseline / extended sequential Huffman)", m - 0xC0);
      } else if (m == 0xDD) {
        if (blen < 2) return fail("bad DRI");
        restart_interval = be16(body);
      } else if (m == 0xDA) {
        if (headers_only) return 0;
        const unsigned char* next = nullptr;
        if (decode_scan(body, blen, p + len, end, &next)) return -1;
        any_scan = true;
        p = next;
        continue;
      }
      p += len;
    }
    if (!saw_sof) return fail("no frame header");
    if (!headers_only && !any_scan) return fail("no scan data");
    return 0;
  }

  void fill_info(dj_jpeg_info* info) const {
    memset(info, 0, sizeof(*info));
    info->width = width;
    info->height = height;
    info->n_components = ncomp;
    info->sof = sof;
    for (int c = 0; c < ncomp; ++c) {
      info->h_samp[c] = comp[c].h;
      info->v_samp[c] = comp[c].v;
      info->blocks_w[c] = comp[c].blocks_w;
      info->blocks_h[c] = comp[c].blocks_h;
      for (int k = 0; k < 64; ++k) info->quant[c][k] = qt_present[comp[c].tq] ? qt[comp[c].tq][k] : 0;
    }
  }

  template <typename T>
  int emit(int c, int normalized, T* out) const {
    const Component& cc = comp[c];
    if (!qt_present[cc.tq] && normalized) return fail("component %d has no quantisation table", c);
    if (!cc.coef) return fail("component %d has no scan data", c);
    for (int by = 0; by < cc.blocks_h; ++by)
      for (int bx = 0; bx < cc.blocks_w; ++bx) {
        const int16_t* src = cc.coef + ((size_t)by * cc.padded_w + bx) * 64;
        T* dst = out + ((size_t)by * cc.blocks_w + bx) * 64;
        if (normalized)
          for (int k = 0; k < 64; ++k) dst[k] = (T)(int16_t)(src[k] * (int)qt[cc.tq][k]);   // jpeg2dct stores shorts
        else
          for (int k = 0; k < 64; ++k) dst[k] = (T)src[k];
      }
    return 0;
  }
};

thread_local Scratch g_scratch;

int decode_one_f32(const unsigned char* data, long size, int normalized, float* y, float* cb, float* cr, int yh, int yw,
                   int ch, int cw, Scratch* scratch) {
  Decoder d;
  d.data = data;
  d.size = size;
  d.scratch = scratch;
  if (d.run(false)) return -1;
  if (d.ncomp != 3) return fail("batch decoding expects 3 components, file has %d", d.ncomp);
  if (d.comp[0].blocks_h != yh || d.comp[0].blocks_w != yw || d.comp[1].blocks_h != ch || d.comp[1].blocks_w != cw ||
      d.comp[2].blocks_h != ch || d.comp[2].blocks_w != cw)
    return fail("block grid %dx%d / %dx%d does not match the batch tensors %dx%d / %dx%d", d.comp[0].blocks_h,
                d.comp[0].blocks_w, d.comp[1].blocks_h, d.comp[1].blocks_w, yh, yw, ch, cw);
  if (d.emit<float>(0, normalized, y) || d.emit<float>(1, normalized, cb) || d.emit<float>(2, normalized, cr)) return -1;
  return 0;
}

}  // namespace

extern "C" const char* dj_jpeg_last_error(void) { return g_err; }

extern "C" int dj_jpeg_read_info(const unsigned char* data, long size, dj_jpeg_info* info) {
  if (!data || !info || size <= 0) return fail("read_info: null argument");
  Decoder d;
  d.data = data;
  d.size = size;
  d.scratch = &g_scratch;
  if (d.run(true)) return -1;
  if (!d.saw_sof) return fail("no frame header");
  d.fill_info(info);
  return 0;
}

extern "C" int dj_jpeg_read_coefficients(const unsigned char* data, long size, int normalized, short* const* planes,
                                         const long* plane_capacity, dj_jpeg_info* info) {
  if (!data || !planes || !plane_capacity || size <= 0) return fail("read_coefficients: null argument");
  Decoder d;
  d.data = data;
  d.size = size;
  d.scratch = &g_scratch;
  if (d.run(false)) return -1;
  for (int c = 0; c < d.ncomp; ++c) {
    long need = (long)d.comp[c].blocks_h * d.comp[c].blocks_w * 64;
    if (!planes[c] || plane_capacity[c] < need) return fail("plane %d too small (%ld < %ld)", c, plane_capacity[c], need);
    if (d.emit<short>(c, normalized, planes[c])) return -1;
  }
  if (info) d.fill_info(info);
  return 0;
}

extern "C" int dj_jpeg_decode_batch_f32(const unsigned char* const* data, const long* sizes, int n, int normalized,
                                        float* y, float* cb, float* cr, int yh, int yw, int ch, int cw, int n_threads) {
  if (!data || !sizes || !y || !cb || !cr || n <= 0) return fail("decode_batch: null argument");
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n) n_threads = n;
  std::atomic<int> next(0), failed(-1);
  std::vector<std::string> msgs((size_t)n_threads);
  const size_t ys = (size_t)yh * yw * 64, cs = (size_t)ch * cw * 64;
  auto work = [&](int t) {
    Scratch scratch;   // one set of planes per worker for the whole batch
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n) break;
      if (decode_one_f32(data[i], sizes[i], normalized, y + i * ys, cb + i * cs, cr + i * cs, yh, yw, ch, cw, &scratch)) {
        int exp = -1;
        if (failed.compare_exchange_strong(exp, i)) msgs[(size_t)t] = g_err;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  if (failed.load() >= 0) {
    for (auto& m : msgs)
      if (!m.empty()) return fail("image %d: %s", failed.load(), m.c_str());
    return fail("image %d failed", failed.load());
  }
  return 0;
}

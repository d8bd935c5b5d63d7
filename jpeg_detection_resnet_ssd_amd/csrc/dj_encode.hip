// SSDInputEncoder on device: ground-truth boxes -> y_true [batch][n_boxes][n_classes + 12] (fp32), one workgroup per image.
// Restates localisation_part/ssd_encoder_decoder/ssd_input_encoder.py:277-418 (`__call__`, coords='centroids'),
// matching_utils.py:22-116 (match_bipartite_greedy, match_multi) and bounding_box_utils.py:283-383 (iou) in the
// reference's own float64 arithmetic and operation order (IEEE double operators, FMA contraction off), so
// every matching decision and tie-break (np.argmax: first maximum) is the reference's.  Index/compare work, no MFMA;
// it replaces a 230 ms host numpy pass per 32 images and a 37 MB host->device copy per step by a few KB of labels.
#include "../../include/dj_hip.h"
#include "dj_common.h"

// every product below must be rounded before it is added to anything (numpy evaluates op by op)
#pragma clang fp contract(off)

// plain operators compiled under the pragma above: the HIP header's __dadd_rn/__dmul_rn are `x + y` / `x * y` built
// with contraction allowed, so inlined they can still fuse into an FMA
__device__ __forceinline__ double enc_add(double a, double b) { return a + b; }
__device__ __forceinline__ double enc_sub(double a, double b) { return a - b; }
__device__ __forceinline__ double enc_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double enc_div(double a, double b) { return a / b; }

#define DJ_ENC_GT_CAP 128
#define DJ_ENC_BOX_CAP 12288
#define DJ_ENC_THREADS 1024

struct DjEncodeParams {
  const double* labels;   // [batch][max_gt][5] = class id, xmin, ymin, xmax, ymax (pixels)
  const int* n_gt;        // [batch]
  const double* anchors;  // [n_boxes][8] = cx, cy, w, h (template coordinates), 4 variances
  float* y_true;
  int max_gt, n_boxes, n_classes, normalize, border, multi, background_id;
  double img_h, img_w, pos_thr, neg_lim;
};

struct GtBox {
  double cx, cy, w, h;     // what the reference stores in labels_one_hot
  double x0, y0, x1, y1;   // corners used by iou()
  double area;
};

__device__ __forceinline__ double enc_iou(const GtBox& g, double ax0, double ay0, double ax1, double ay1, double aarea) {
  double iw = fmax(0.0, enc_sub(fmin(g.x1, ax1), fmax(g.x0, ax0)));
  double ih = fmax(0.0, enc_sub(fmin(g.y1, ay1), fmax(g.y0, ay0)));
  double inter = enc_mul(iw, ih);
  return enc_div(inter, enc_sub(enc_add(g.area, aarea), inter));
}

__device__ __forceinline__ void enc_anchor_corners(const double* an, double d, double& x0, double& y0, double& x1,
                                                   double& y1, double& area) {
  double hw = enc_div(an[2], 2.0), hh = enc_div(an[3], 2.0);
  x0 = enc_sub(an[0], hw);
  y0 = enc_sub(an[1], hh);
  x1 = enc_add(an[0], hw);
  y1 = enc_add(an[1], hh);
  area = enc_mul(enc_add(enc_sub(x1, x0), d), enc_add(enc_sub(y1, y0), d));
}

// wave-wide arg-max with np.argmax tie-breaking (larger value, then lower index)
__device__ __forceinline__ void enc_wave_argmax(double& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(v, o);
    int oi = __shfl_xor(i, o);
    if (ov > v || (ov == v && oi < i)) {
      v = ov;
      i = oi;
    }
  }
}

__global__ __launch_bounds__(DJ_ENC_THREADS) void dj_ssd_encode_kernel(DjEncodeParams p) {
  __shared__ GtBox gt[DJ_ENC_GT_CAP];
  __shared__ int gt_class[DJ_ENC_GT_CAP];
  __shared__ double best_val[DJ_ENC_GT_CAP];
  __shared__ int best_idx[DJ_ENC_GT_CAP];
  __shared__ int redo[DJ_ENC_GT_CAP];
  __shared__ int n_redo;
  __shared__ int match[DJ_ENC_GT_CAP];
  __shared__ unsigned char gt_done[DJ_ENC_GT_CAP];
  __shared__ short assign[DJ_ENC_BOX_CAP];  // -1 unmatched, -2 neutral, -3 taken by the bipartite pass (temporary), >= 0 gt
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = DJ_ENC_THREADS / 64;
  const int n = min(p.n_gt[img], p.max_gt);
  const double d = (double)p.border;
  const int N = p.n_boxes;

  // ---- ground truth: normalise, corners -> centroids (what is stored), centroids -> corners (what iou() uses) ----
  if (tid < n) {
    const double* l = p.labels + ((size_t)img * p.max_gt + tid) * 5;
    double x0 = l[1], y0 = l[2], x1 = l[3], y1 = l[4];
    if (p.normalize) {
      y0 = enc_div(y0, p.img_h);
      y1 = enc_div(y1, p.img_h);
      x0 = enc_div(x0, p.img_w);
      x1 = enc_div(x1, p.img_w);
    }
    GtBox g;
    g.cx = enc_div(enc_add(x0, x1), 2.0);
    g.cy = enc_div(enc_add(y0, y1), 2.0);
    g.w = enc_add(enc_sub(x1, x0), d);
    g.h = enc_add(enc_sub(y1, y0), d);
    double hw = enc_div(g.w, 2.0), hh = enc_div(g.h, 2.0);
    g.x0 = enc_sub(g.cx, hw);
    g.y0 = enc_sub(g.cy, hh);
    g.x1 = enc_add(g.cx, hw);
    g.y1 = enc_add(g.cy, hh);
    g.area = enc_mul(enc_add(enc_sub(g.x1, g.x0), d), enc_add(enc_sub(g.y1, g.y0), d));
    gt[tid] = g;
    gt_class[tid] = (int)l[0];
    gt_done[tid] = 0;
    match[tid] = 0;
  }
  for (int a = tid; a < N; a += DJ_ENC_THREADS) assign[a] = -1;
  __syncthreads();

  if (n > 0) {
    // ---- greedy bipartite matching.  best_val/best_idx[g] = row arg-max of the (virtually zeroed) weight matrix ----
    for (int g = wave; g < n; g += n_waves) {
      double bv = -1.0;
      int bi = 0x7fffffff;
      const GtBox gg = gt[g];
      for (int a = lane; a < N; a += 64) {
        double ax0, ay0, ax1, ay1, aa;
        enc_anchor_corners(p.anchors + (size_t)a * 8, d, ax0, ay0, ax1, ay1, aa);
        double v = enc_iou(gg, ax0, ay0, ax1, ay1, aa);
        if (v > bv) {
          bv = v;
          bi = a;
        }
      }
      enc_wave_argmax(bv, bi);
      if (lane == 0) {
        best_val[g] = bv;
        best_idx[g] = bi;
      }
    }
    __syncthreads();
    for (int it = 0; it < n; ++it) {
      if (tid == 0) {
        double bv = best_val[0];
        int bg = 0;
        for (int g = 1; g < n; ++g)
          if (best_val[g] > bv) {
            bv = best_val[g];
            bg = g;
          }
        const int a = best_idx[bg];
        match[bg] = a;
        gt_done[bg] = 1;     // its row is all zeros from now on: arg-max = (0, index 0)
        best_val[bg] = 0.0;
        best_idx[bg] = 0;
        assign[a] = -3;      // the column is zeroed
        int c = 0;
        for (int g = 0; g < n; ++g)
          if (!gt_done[g] && best_idx[g] == a && best_val[g] > 0.0) redo[c++] = g;
        n_redo = c;
      }
      __syncthreads();
      const int nr = n_redo;
      for (int r = wave; r < nr; r += n_waves) {
        const int g = redo[r];
        double bv = 0.0;        // zeroed entries and natural zeros tie at 0 -> index 0
        int bi = 0;
        const GtBox gg = gt[g];
        double lv = -1.0;
        int li = 0x7fffffff;
        for (int a = lane; a < N; a += 64) {
          if (assign[a] == -3) continue;
          double ax0, ay0, ax1, ay1, aa;
          enc_anchor_corners(p.anchors + (size_t)a * 8, d, ax0, ay0, ax1, ay1, aa);
          double v = enc_iou(gg, ax0, ay0, ax1, ay1, aa);
          if (v > lv) {
            lv = v;
            li = a;
          }
        }
        enc_wave_argmax(lv, li);
        if (lv > 0.0) {
          bv = lv;
          bi = li;
        }
        if (lane == 0) {
          best_val[g] = bv;
          best_idx[g] = bi;
        }
      }
      __syncthreads();
    }
    // y_encoded[i, bipartite_matches, :-8] = labels_one_hot  (duplicate indices: the last ground truth wins)
    if (tid == 0)
      for (int g = 0; g < n; ++g) assign[match[g]] = (short)g;
    __syncthreads();

    // ---- multi matching and neutral boxes: per anchor, over the columns not taken by the bipartite pass ----
    for (int a = tid; a < N; a += DJ_ENC_THREADS) {
      if (assign[a] != -1) continue;
      double ax0, ay0, ax1, ay1, aa;
      enc_anchor_corners(p.anchors + (size_t)a * 8, d, ax0, ay0, ax1, ay1, aa);
      double bv = enc_iou(gt[0], ax0, ay0, ax1, ay1, aa);
      int bg = 0;
      for (int g = 1; g < n; ++g) {
        double v = enc_iou(gt[g], ax0, ay0, ax1, ay1, aa);
        if (v > bv) {
          bv = v;
          bg = g;
        }
      }
      if (p.multi && bv >= p.pos_thr)
        assign[a] = (short)bg;
      else if (bv >= p.neg_lim)
        assign[a] = -2;
    }
    __syncthreads();
  }

  // ---- write the rows: [one-hot | offsets | anchor | variances], offsets in the reference's order of operations ----
  const int width = p.n_classes + 12;
  float* out = p.y_true + (size_t)img * N * width;
  const long total = (long)N * width;
  for (long i = tid; i < total; i += DJ_ENC_THREADS) {
    const int a = (int)(i / width), col = (int)(i - (long)a * width);
    const int s = assign[a];
    const double* an = p.anchors + (size_t)a * 8;
    float v;
    if (col < p.n_classes) {
      if (s >= 0)
        v = (col == gt_class[s]) ? 1.f : 0.f;
      else
        v = (col == p.background_id && s != -2) ? 1.f : 0.f;   // -2: neutral, its background flag is cleared
    } else if (col < p.n_classes + 4) {
      const int k = col - p.n_classes;
      double r = 0.0;
      if (s >= 0) {
        const GtBox g = gt[s];
        if (k == 0)
          r = enc_div(enc_sub(g.cx, an[0]), enc_mul(an[2], an[4]));
        else if (k == 1)
          r = enc_div(enc_sub(g.cy, an[1]), enc_mul(an[3], an[5]));
        else if (k == 2)
          r = enc_div(log(enc_div(g.w, an[2])), an[6]);
        else
          r = enc_div(log(enc_div(g.h, an[3])), an[7]);
      }
      v = (float)r;
    } else {
      v = (float)an[col - p.n_classes - 4];
    }
    out[i] = v;
  }
}

extern "C" int dj_ssd_encode_targets(const double* labels, const int* n_gt, int batch, int max_gt, const double* anchors,
                                     int n_boxes, int n_classes, int img_height, int img_width, int normalize_coords,
                                     int border_pixels, int multi, double pos_iou_threshold, double neg_iou_limit,
                                     int background_id, float* y_true, void* stream) {
  DJ_CHECK_ARG(labels && n_gt && anchors && y_true, "ssd_encode: null tensor");
  DJ_CHECK_ARG(batch > 0 && n_boxes > 0 && n_classes > 1 && img_height > 0 && img_width > 0, "ssd_encode: bad sizes");
  DJ_CHECK_ARG(max_gt >= 1 && max_gt <= DJ_ENC_GT_CAP, "ssd_encode: at most %d ground-truth boxes per image", DJ_ENC_GT_CAP);
  DJ_CHECK_ARG(n_boxes <= DJ_ENC_BOX_CAP, "ssd_encode: more than %d anchor boxes", DJ_ENC_BOX_CAP);
  DJ_CHECK_ARG(border_pixels >= -1 && border_pixels <= 1, "ssd_encode: border_pixels must be -1 (exclude), 0 (half), 1 (include)");
  DJ_CHECK_ARG(pos_iou_threshold > 0.0 && neg_iou_limit > 0.0, "ssd_encode: thresholds must be positive");
  DJ_CHECK_ARG(background_id >= 0 && background_id < n_classes, "ssd_encode: background_id out of range");
  DjEncodeParams p;
  p.labels = labels;
  p.n_gt = n_gt;
  p.anchors = anchors;
  p.y_true = y_true;
  p.max_gt = max_gt;
  p.n_boxes = n_boxes;
  p.n_classes = n_classes;
  p.normalize = normalize_coords;
  p.border = border_pixels;
  p.multi = multi;
  p.background_id = background_id;
  p.img_h = (double)img_height;
  p.img_w = (double)img_width;
  p.pos_thr = pos_iou_threshold;
  p.neg_lim = neg_iou_limit;
  hipLaunchKernelGGL(dj_ssd_encode_kernel, dim3(batch), dim3(DJ_ENC_THREADS), 0, (hipStream_t)stream, p);
  DJ_CHECK_LAUNCH("dj_ssd_encode_targets");
  return DJ_OK;
}

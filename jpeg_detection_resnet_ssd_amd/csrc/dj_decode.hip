// DecodeDetections on device: box decoding, per-class confidence threshold + greedy NMS, top-k merge.
// Replaces the TF subgraph of localisation_part/keras_layers/keras_layer_DecodeDetections.py:109-265
// (tf.image.non_max_suppression per class inside two tf.map_fn loops) == the numpy
// localisation_part/ssd_encoder_decoder/ssd_output_decoder.py:111-226 (`decode_detections`, `_greedy_nms`).
// Integer/index work: HBM/LDS bound, no MFMA.
#include "../../include/dj_hip.h"
#include "dj_common.h"

// boxes[b*N + i] = (xmin, ymin, xmax, ymax) from centroid offsets + anchors + variances (last 12 columns)
__global__ __launch_bounds__(256) void dj_decode_boxes_kernel(const float* y_pred, long nbox, int width, float sx,
                                                               float sy, f32x4* boxes) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nbox; i += (long)gridDim.x * 256) {
    const float* p = y_pred + i * width + (width - 12);
    float cx = p[0] * p[8] * p[6] + p[4];
    float cy = p[1] * p[9] * p[7] + p[5];
    float w = expf(p[2] * p[10]) * p[6];
    float h = expf(p[3] * p[11]) * p[7];
    boxes[i] = f32x4{(cx - 0.5f * w) * sx, (cy - 0.5f * h) * sy, (cx + 0.5f * w) * sx, (cy + 0.5f * h) * sy};
  }
}

#define DJ_NMS_CAP 12288

// one workgroup per (class, image): greedy NMS in descending score order (ties: lowest box index), a box is
// dropped when IoU > iou_thresh with a kept one; at most max_out kept.  kept rows: [class, conf, xmin, ymin, xmax, ymax]
// fast_score / fast_class (DecodeDetectionsFast): per-box score and class id instead of one class column of y_pred
__global__ __launch_bounds__(256) void dj_nms_class_kernel(const float* y_pred, const f32x4* boxes, int N, int width,
                                                            float conf_thresh, float iou_thresh, int max_out,
                                                            float* kept, int* counts, const float* fast_score,
                                                            const float* fast_class) {
  __shared__ float score[DJ_NMS_CAP];
  __shared__ float wmax[4];
  __shared__ int widx[4];
  __shared__ int s_best;
  const int cls = blockIdx.x + 1, img = blockIdx.y, n_fg = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* yp = y_pred + (size_t)img * N * width;
  const f32x4* bx = boxes + (size_t)img * N;
  for (int i = tid; i < N; i += 256) {
    float s = fast_score ? fast_score[(size_t)img * N + i] : yp[(size_t)i * width + cls];
    score[i] = (s > conf_thresh) ? s : -1.f;
  }
  __syncthreads();
  float* out = kept + ((size_t)img * n_fg + blockIdx.x) * max_out * 6;
  int nk = 0;
  while (nk < max_out) {
    float m = -1.f;
    int mi = 0x7fffffff;
    for (int i = tid; i < N; i += 256) {
      float s = score[i];
      if (s > m) {  // strict: the lowest index wins among equal scores within a thread (indices ascend)
        m = s;
        mi = i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float om = __shfl_xor(m, o);
      int oi = __shfl_xor(mi, o);
      if (om > m || (om == m && oi < mi)) {
        m = om;
        mi = oi;
      }
    }
    if (lane == 0) {
      wmax[wave] = m;
      widx[wave] = mi;
    }
    __syncthreads();
    if (tid == 0) {
      float bm = wmax[0];
      int bi = widx[0];
      for (int w = 1; w < 4; ++w)
        if (wmax[w] > bm || (wmax[w] == bm && widx[w] < bi)) {
          bm = wmax[w];
          bi = widx[w];
        }
      s_best = (bm > 0.f) ? bi : -1;
    }
    __syncthreads();
    const int best = s_best;
    if (best < 0) break;
    const f32x4 bb = bx[best];
    if (tid == 0) {
      float* o = out + (size_t)nk * 6;
      o[0] = fast_class ? fast_class[(size_t)img * N + best] : (float)cls;
      o[1] = score[best];
      o[2] = bb.x;
      o[3] = bb.y;
      o[4] = bb.z;
      o[5] = bb.w;
    }
    ++nk;
    const float area_b = (bb.z - bb.x) * (bb.w - bb.y);
    __syncthreads();  // score[best] read above before it is cleared below
    for (int i = tid; i < N; i += 256) {
      if (score[i] <= 0.f) continue;
      if (i == best) {
        score[i] = -1.f;
        continue;
      }
      f32x4 c = bx[i];
      float iw = fmaxf(0.f, fminf(bb.z, c.z) - fmaxf(bb.x, c.x));
      float ih = fmaxf(0.f, fminf(bb.w, c.w) - fmaxf(bb.y, c.y));
      float inter = iw * ih;
      float uni = area_b + (c.z - c.x) * (c.w - c.y) - inter;
      if (inter / uni > iou_thresh) score[i] = -1.f;
    }
    __syncthreads();
  }
  if (tid == 0) counts[img * n_fg + blockIdx.x] = nk;
}

// one workgroup per image: sort the (<= 8192) kept rows by confidence (descending; ties: class, then NMS order)
// and write the first top_k, zero-padded.
#define DJ_TOPK_CAP 8192
__global__ __launch_bounds__(1024) void dj_topk_merge_kernel(const float* kept, const int* counts, int n_fg, int max_out,
                                                              int top_k, float* out) {
  __shared__ unsigned long long key[DJ_TOPK_CAP];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int total = n_fg * max_out;
  const float* kp = kept + (size_t)img * total * 6;
  for (int i = tid; i < DJ_TOPK_CAP; i += 1024) {
    unsigned long long k = 0ull;
    if (i < total) {
      int c = i / max_out, j = i - c * max_out;
      if (j < counts[img * n_fg + c]) {
        unsigned s = __float_as_uint(kp[(size_t)i * 6 + 1]);  // confidences are > 0: bit patterns order like uints
        k = ((unsigned long long)s << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      }
    }
    key[i] = k;
  }
  __syncthreads();
  // bitonic sort, descending
  for (int size = 2; size <= DJ_TOPK_CAP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < DJ_TOPK_CAP / 2; t += 1024) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        unsigned long long a = key[lo], b = key[hi];
        if ((a < b) == desc) {
          key[lo] = b;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  float* o = out + (size_t)img * top_k * 6;
  for (int r = tid; r < top_k; r += 1024) {
    unsigned long long k = (r < DJ_TOPK_CAP) ? key[r] : 0ull;
    if (k != 0ull) {
      unsigned i = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
      for (int q = 0; q < 6; ++q) o[(size_t)r * 6 + q] = kp[(size_t)i * 6 + q];
    } else {
      for (int q = 0; q < 6; ++q) o[(size_t)r * 6 + q] = 0.f;
    }
  }
}

extern "C" long dj_decode_detections_workspace_floats(int batch, int n_boxes, int n_classes, int nms_max_output_size) {
  long n_fg = n_classes - 1;
  return (long)batch * n_boxes * 4 + (long)batch * n_fg * nms_max_output_size * 6 + (long)batch * n_fg + 16;
}

extern "C" int dj_decode_detections(const float* y_pred, int batch, int n_boxes, int n_classes, float confidence_thresh,
                                    float iou_threshold, int top_k, int nms_max_output_size, int normalize_coords,
                                    int img_height, int img_width, float* workspace, float* out, void* stream) {
  DJ_CHECK_ARG(y_pred && workspace && out, "decode_detections: null tensor");
  DJ_CHECK_ARG(batch > 0 && n_boxes > 0 && n_classes > 1 && top_k > 0 && nms_max_output_size > 0,
               "decode_detections: bad sizes");
  DJ_CHECK_ARG(n_boxes <= DJ_NMS_CAP, "decode_detections: more than %d boxes per image", DJ_NMS_CAP);
  const int n_fg = n_classes - 1;
  DJ_CHECK_ARG((long)n_fg * nms_max_output_size <= DJ_TOPK_CAP,
               "decode_detections: (n_classes-1) * nms_max_output_size exceeds %d", DJ_TOPK_CAP);
  hipStream_t s = (hipStream_t)stream;
  const int width = n_classes + 12;
  f32x4* boxes = reinterpret_cast<f32x4*>(workspace);
  float* kept = workspace + (size_t)batch * n_boxes * 4;
  int* counts = reinterpret_cast<int*>(kept + (size_t)batch * n_fg * nms_max_output_size * 6);
  const long nbox = (long)batch * n_boxes;
  const float sx = normalize_coords ? (float)img_width : 1.f, sy = normalize_coords ? (float)img_height : 1.f;
  long blocks = (nbox + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dj_decode_boxes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y_pred, nbox, width, sx, sy, boxes);
  DJ_CHECK_LAUNCH("dj_decode_boxes");
  hipLaunchKernelGGL(dj_nms_class_kernel, dim3(n_fg, batch), dim3(256), 0, s, y_pred, boxes, n_boxes, width,
                     confidence_thresh, iou_threshold, nms_max_output_size, kept, counts, (const float*)nullptr,
                     (const float*)nullptr);
  DJ_CHECK_LAUNCH("dj_nms_class");
  hipLaunchKernelGGL(dj_topk_merge_kernel, dim3(batch), dim3(1024), 0, s, kept, counts, n_fg, nms_max_output_size, top_k,
                     out);
  DJ_CHECK_LAUNCH("dj_topk_merge");
  return DJ_OK;
}

// ---- DecodeDetectionsFast (localisation_part/keras_layers/keras_layer_DecodeDetectionsFast.py:108-215): every box keeps
// only its arg-max class (tf.argmax: first maximum) and that confidence; background boxes are dropped, the rest is
// thresholded, ONE class-agnostic NMS, top-k.
__global__ __launch_bounds__(256) void dj_decode_fast_prep_kernel(const float* y_pred, long nbox, int width, int n_classes,
                                                                   float* score, float* cls) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nbox; i += (long)gridDim.x * 256) {
    const float* p = y_pred + i * width;
    float m = p[0];
    int mi = 0;
    for (int c = 1; c < n_classes; ++c)
      if (p[c] > m) {
        m = p[c];
        mi = c;
      }
    score[i] = (mi != 0) ? m : -1.f;   // class 0 = background: never a detection
    cls[i] = (float)mi;
  }
}

extern "C" long dj_decode_detections_fast_workspace_floats(int batch, int n_boxes, int nms_max_output_size) {
  return (long)batch * n_boxes * 6 + (long)batch * nms_max_output_size * 6 + batch + 16;
}

extern "C" int dj_decode_detections_fast(const float* y_pred, int batch, int n_boxes, int n_classes,
                                         float confidence_thresh, float iou_threshold, int top_k, int nms_max_output_size,
                                         int normalize_coords, int img_height, int img_width, float* workspace, float* out,
                                         void* stream) {
  DJ_CHECK_ARG(y_pred && workspace && out, "decode_detections_fast: null tensor");
  DJ_CHECK_ARG(batch > 0 && n_boxes > 0 && n_classes > 1 && top_k > 0 && nms_max_output_size > 0,
               "decode_detections_fast: bad sizes");
  DJ_CHECK_ARG(n_boxes <= DJ_NMS_CAP, "decode_detections_fast: more than %d boxes per image", DJ_NMS_CAP);
  DJ_CHECK_ARG(nms_max_output_size <= DJ_TOPK_CAP, "decode_detections_fast: nms_max_output_size exceeds %d", DJ_TOPK_CAP);
  hipStream_t s = (hipStream_t)stream;
  const int width = n_classes + 12;
  const long nbox = (long)batch * n_boxes;
  f32x4* boxes = reinterpret_cast<f32x4*>(workspace);
  float* score = workspace + nbox * 4;
  float* cls = score + nbox;
  float* kept = cls + nbox;
  int* counts = reinterpret_cast<int*>(kept + (size_t)batch * nms_max_output_size * 6);
  const float sx = normalize_coords ? (float)img_width : 1.f, sy = normalize_coords ? (float)img_height : 1.f;
  long blocks = (nbox + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dj_decode_boxes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y_pred, nbox, width, sx, sy, boxes);
  DJ_CHECK_LAUNCH("dj_decode_boxes");
  hipLaunchKernelGGL(dj_decode_fast_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y_pred, nbox, width, n_classes,
                     score, cls);
  DJ_CHECK_LAUNCH("dj_decode_fast_prep");
  hipLaunchKernelGGL(dj_nms_class_kernel, dim3(1, batch), dim3(256), 0, s, y_pred, boxes, n_boxes, width, confidence_thresh,
                     iou_threshold, nms_max_output_size, kept, counts, (const float*)score, (const float*)cls);
  DJ_CHECK_LAUNCH("dj_nms_fast");
  hipLaunchKernelGGL(dj_topk_merge_kernel, dim3(batch), dim3(1024), 0, s, kept, counts, 1, nms_max_output_size, top_k, out);
  DJ_CHECK_LAUNCH("dj_topk_merge");
  return DJ_OK;
}

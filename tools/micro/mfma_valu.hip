// Microbenchmark: how much of the VALU / SALU / LDS / VMEM issue hides behind v_mfma_f32_32x32x2_f32 on gfx950?
// One wave per SIMD (256-thread blocks, one per CU) or two (512), a loop of 16 independent MFMAs with NV extra VALU
// (v_fma_f32 on private registers), NS SALU adds, NL ds_read_b128, NG buffer loads per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NS, int NL, int NG, int NC>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
  __shared__ float lds[4096];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  int s = iters;
  f32x4 l = {0, 0, 0, 0};
  float g = 0.f;
  lds[threadIdx.x] = a;
  __syncthreads();
  const float* gp = in + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(m + q) & 7]) : "v"(a));
#pragma unroll
      for (int q = 0; q < NC; ++q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[(m + q) & 7]) : "v"(a));
#pragma unroll
      for (int q = 0; q < NS; ++q) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        f32x4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 16));
        l += t;
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        float t;
        asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(gp));
        g += t;
      }
    }
  }
  float r = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int q = 0; q < 16; ++q) r += acc[i][q];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + s + l.x + g;
}

template <int NV, int NS, int NL, int NG, int NC>
void run(const char* name, int threads, float* out, float* in) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NS, NL, NG, NC>), dim3(256), dim3(threads), 0, 0, out, in, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NS, NL, NG, NC>), dim3(256), dim3(threads), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double mfma = (double)iters * 16 * (threads / 64) * 256;
  double tf = mfma * 4096 * 2 / 2 / (ms * 1e-3) / 1e12;   // 32*32*2*2 flop per mfma
  double cyc_per_mfma_at_2p4 = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (threads / 256));
  printf("%-44s threads %3d  %8.3f ms  %6.1f TF  (%.1f cyc/MFMA/SIMD at 2.4 GHz)\n", name, threads, ms, tf, cyc_per_mfma_at_2p4);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 512 * 256 * 4);
  hipMalloc(&in, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  for (int threads = 256; threads <= 512; threads += 256) {
    run<0, 0, 0, 0, 0>("bare MFMA", threads, out, in);
    run<1, 0, 0, 0, 0>("+1 v_fma per MFMA", threads, out, in);
    run<2, 0, 0, 0, 0>("+2 v_fma per MFMA", threads, out, in);
    run<4, 0, 0, 0, 0>("+4 v_fma per MFMA", threads, out, in);
    run<8, 0, 0, 0, 0>("+8 v_fma per MFMA", threads, out, in);
    run<0, 0, 0, 0, 4>("+4 v_cndmask per MFMA", threads, out, in);
    run<0, 4, 0, 0, 0>("+4 s_add per MFMA", threads, out, in);
    run<0, 8, 0, 0, 0>("+8 s_add per MFMA", threads, out, in);
    run<0, 0, 1, 0, 0>("+1 ds_read_b128 per MFMA", threads, out, in);
    run<0, 0, 2, 0, 0>("+2 ds_read_b128 per MFMA", threads, out, in);
    run<0, 0, 0, 1, 0>("+1 global_load_dword per MFMA", threads, out, in);
    run<0, 0, 0, 2, 0>("+2 global_load_dword per MFMA", threads, out, in);
    run<2, 2, 1, 1, 0>("+2 fma 2 s_add 1 ds_read 1 gload per MFMA", threads, out, in);
  }
  return 0;
}

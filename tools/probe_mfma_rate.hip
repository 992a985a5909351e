// Sustained rate of the two bf16 MFMA shapes in the access pattern of the halo kernel's tap (one wave per SIMD, one workgroup per
// CU, 64 accumulator registers per lane), whole chip busy, by ORDER of the 32 (16) MFMAs of a tap:
//   order 0: A fragment shared by consecutive MFMAs (the kernels' order: for tm, for tn)   1: B shared (for tn, for tm)
//   order 2: diagonal (neither operand repeats back to back)
// Prints ns per tap from the difference of a long and a short launch.
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/probe_mfma_rate tools/probe_mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int SHAPE, int ORDER>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, const bf16x8* src) {
  __shared__ char ballast[100 * 1024];
  if (iters < 0) ballast[threadIdx.x] = 1;
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + i * 64) & 511]; b[i] = src[(threadIdx.x + i * 64 + 33) & 511]; }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            const int i = ORDER == 0 ? x : (ORDER == 1 ? y : y), j = ORDER == 0 ? y : (ORDER == 1 ? x : (x + y) & 3);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk * 4 + i], b[kk * 4 + j], acc[i][j], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            const int i = ORDER == 0 ? x : y, j = ORDER == 0 ? y : (ORDER == 1 ? x : (x + y) & 1);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks * 2 + i], b[ks * 2 + j], acc[i][j], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE, int ORDER>
static double run(float* out, const bf16x8* src, int iters, hipEvent_t e0, hipEvent_t e1) {
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int l = 0; l < 20; ++l) k<SHAPE, ORDER><<<256, 256>>>(out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e-3 / 20;
}
template <int SHAPE, int ORDER>
static void report(float* out, const bf16x8* src, hipEvent_t e0, hipEvent_t e1) {
  const double t0 = run<SHAPE, ORDER>(out, src, 100, e0, e1), t1 = run<SHAPE, ORDER>(out, src, 4100, e0, e1);
  const double per = (t1 - t0) / 4000, fl = 2.0 * 64 * 64 * 64 * 4 * 256;
  printf("mfma %dx order %d: %.1f ns per tap (64x64x64 per wave), %.0f TFLOP/s, %.2f ns per MFMA; short launch %.1f us\n", SHAPE, ORDER, per * 1e9, fl / per / 1e12,
         per * 1e9 / (SHAPE == 16 ? 32 : 16), t0 * 1e6);
}
int main() {
  float* out; bf16x8* src;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&src, 8192);
  short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (short)(0x3f80 + (i * 37 % 64));
  hipMemcpy(src, h, 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  report<16, 0>(out, src, e0, e1); report<16, 1>(out, src, e0, e1); report<16, 2>(out, src, e0, e1);
  report<32, 0>(out, src, e0, e1); report<32, 1>(out, src, e0, e1); report<32, 2>(out, src, e0, e1);
  return 0;
}

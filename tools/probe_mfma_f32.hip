// What does v_mfma_f32_32x32x2_f32 deliver when NOTHING else happens?  (the ceiling of conv_f32 / labelprop_f32)
// Every wave issues chains of MFMAs on register operands: CHAINS independent accumulators (1, 2 or 4), waves per SIMD 1..4.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[0] = s;
}
// the same chains on RANDOM operands (16 different register pairs per lane, cycled): the multiplier arrays toggle as they do on real
// data - does the clock hold under that load, or is the documented peak a low-toggle figure?
template <int CHAINS>
__global__ __launch_bounds__(256) void krand(float* out, int iters, unsigned seed) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a[16], b[16];
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int u = 0; u < 16; ++u) {
    s = s * 1664525u + 1013904223u; a[u] = ((int)(s >> 9) % 2001 - 1000) * 1.0e-3f;
    s = s * 1664525u + 1013904223u; b[u] = ((int)(s >> 9) % 2001 - 1000) * 1.0e-3f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[c], 0, 0, 0);
  }
  float t = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) t += acc[c][i];
  if (t == 12345.678f) out[0] = t;
}
template <int CHAINS>
static void run_rand(int wgs_per_cu, int cus, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(krand<CHAINS>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 10, 7u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(krand<CHAINS>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 7u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)cus * wgs_per_cu * 4 * iters * 16.0 * CHAINS * 4096.0;
  printf("RANDOM operands, chains %d, %d waves/SIMD, %d iterations: %.1f TFLOP/s (%.2f ms)\n", CHAINS, wgs_per_cu, iters, flop / ms / 1e9, ms);
  hipFree(out);
}
template <int CHAINS>
static void run(int wgs_per_cu, int cus) {
  float* out; hipMalloc(&out, 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)cus * wgs_per_cu * 4 * iters * 16.0 * CHAINS * 4096.0;
  printf("chains %d, %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", CHAINS, wgs_per_cu, flop / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  const int cus = p.multiProcessorCount;
  run<1>(1, cus); run<1>(2, cus); run<1>(4, cus); run<2>(1, cus); run<4>(1, cus); run<4>(2, cus);
  run_rand<1>(2, cus, 4000); run_rand<1>(3, cus, 4000); run_rand<2>(2, cus, 4000); run_rand<1>(3, cus, 40000); run<1>(3, cus);
  return 0;
}

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
  return 0;
}

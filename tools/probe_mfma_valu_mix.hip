// Do vector instructions run in the shadow of v_mfma_f32_32x32x2_f32 (64 clk per SIMD), or does the fp32-input MFMA occupy the vector
// lanes?  Every wave issues MFMAs on register operands with NV independent VALU instructions (v_add_u32 / v_fma_f32, 4 clk each) after
// each one; 1..4 waves per SIMD.  If the two pipes overlap, the MFMA rate holds until NV * 4 clk approaches 64 clk.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  unsigned v[8]; float f[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; f[i] = a + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
        else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j & 7]) : "v"(f[(j + 1) & 7]));
        else asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  for (int i = 0; i < 8; ++i) s += (float)v[i] + f[i];
  if (s == 12345.678f) out[0] = s;
}
// the same with v_mfma_f32_32x32x16_bf16 (8 passes = 32 clk per SIMD): is the bf16 matrix pipe independent of the vector lanes?
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
template <int NV>
__global__ __launch_bounds__(256) void kb(float* out, int iters, float a0) {
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + threadIdx.x * 1e-3f); b[i] = (__bf16)1.0f; }
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
    }
  }
  float s = 0.f;
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  for (int i = 0; i < 8; ++i) s += (float)v[i];
  if (s == 12345.678f) out[0] = s;
}
template <int NV>
static void runb(int wgs_per_cu, int cus) {
  float* out; hipMalloc(&out, 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kb<NV>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kb<NV>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)cus * wgs_per_cu * 4 * iters * 16.0 * 32768.0;
  printf("%2d x v_add_u32    per bf16 MFMA (32x32x16), %d waves/SIMD: %6.1f TFLOP/s of MFMA (%.2f ms)\n", NV, wgs_per_cu, flop / ms / 1e9, ms);
  hipFree(out);
}
template <int NV, int KIND>
static void run(int wgs_per_cu, int cus) {
  float* out; hipMalloc(&out, 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)cus * wgs_per_cu * 4 * iters * 16.0 * 4096.0;
  const char* kn = KIND == 0 ? "v_add_u32" : KIND == 1 ? "v_fma_f32" : "v_mul_lo_u32";
  printf("%2d x %-12s per MFMA, %d waves/SIMD: %6.1f TFLOP/s of MFMA (%.2f ms)\n", NV, kn, wgs_per_cu, flop / ms / 1e9, ms);
  hipFree(out);
}
template <int KIND> static void sweep(int cus) {
  for (int w = 1; w <= 4; w += 3) {
    run<0, KIND>(w, cus); run<1, KIND>(w, cus); run<2, KIND>(w, cus); run<4, KIND>(w, cus); run<6, KIND>(w, cus);
    run<8, KIND>(w, cus); run<12, KIND>(w, cus); run<16, KIND>(w, cus);
  }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  sweep<0>(p.multiProcessorCount); sweep<1>(p.multiProcessorCount); sweep<2>(p.multiProcessorCount);
  for (int w = 1; w <= 4; w += 3) {
    const int cus = p.multiProcessorCount;
    runb<0>(w, cus); runb<1>(w, cus); runb<2>(w, cus); runb<4>(w, cus); runb<6>(w, cus); runb<8>(w, cus); runb<12>(w, cus); runb<16>(w, cus);
  }
  return 0;
}

// Probe: does v_cvt_pk_bf16_f32 (gfx950) agree bit-for-bit with the integer round-to-nearest-even used
// by the CPU emulator build, for every non-NaN fp32 pattern class (normals, denormals, inf, ties)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ uint32_t hw(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ uint16_t sw(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__global__ void k(unsigned long long* bad, uint32_t* first) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t i = tid; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t u = (uint32_t)i;
    const float f = __builtin_bit_cast(float, u);
    if (f != f) continue;
    const uint32_t a = hw(f, -f), b = (uint32_t)sw(f) | ((uint32_t)sw(-f) << 16);
    if (a != b) {
      if (atomicAdd(bad, 1ull) == 0) { first[0] = u; first[1] = a; first[2] = b; }
    }
  }
}
int main() {
  unsigned long long* bad; uint32_t* first;
  hipMalloc(&bad, 8); hipMalloc(&first, 12);
  hipMemset(bad, 0, 8); hipMemset(first, 0, 12);
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, first);
  unsigned long long hb; uint32_t hf[3];
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 12, hipMemcpyDeviceToHost);
  printf("cvt_pk_bf16_f32 vs integer RNE over all 2^32 patterns: mismatches %llu (first %08x hw %08x sw %08x)\n", hb, hf[0], hf[1], hf[2]);
  return 0;
}

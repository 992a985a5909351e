// How fast does a CU pull L2-/L1-resident data through the vector memory path, by instruction flavour?  (what bounds the operand
// staging of labelprop_f32: tools/probe_lp_stage.hip found the LDS-DMA ring's transfers ALONE take 3.0 ms per 22 GB, wherever the
// data comes from)  Every wave issues ROUNDS x 8 loads of one flavour on a 1 MB region (L2-resident), 3 workgroups per CU.
//   0  buffer_load_dwordx4 ... lds, lane = (16-byte group, row): 8 rows x 128 B per instruction, lanes 0-7 on 8 different lines
//   1  buffer_load_dwordx4 ... lds, lane = (row, group): lanes 0-7 = one 128-byte line
//   2  buffer_load_dwordx4 ... lds, 1 KB contiguous
//   3  buffer_load_dword ... lds, 256 B contiguous
//   4  global_load_dwordx4 -> VGPRs, 1 KB contiguous
//   5  global_load_dwordx4 -> VGPRs, 4 KB stride between lanes (the register-staged kernel's pattern)
//   6  global_load_dwordx4 -> VGPRs, lane = (row, group) as 1
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return (u32x4){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
template <int F>
__global__ __launch_bounds__(256) void k(const float* src, float* out, int rounds, unsigned region) {
  __shared__ __attribute__((aligned(16))) float lds[4][8][256];     // 8 KB per wave: 8 transfers in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 rs = make_rsrc(src, region);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned lbase = (unsigned)(size_t)(lds_ptr)&lds[wave][0][0];
  unsigned vo;
  const unsigned row_stride = 4096;
  if (F == 0) vo = (lane & 7) * row_stride + (lane >> 3) * 16;
  else if (F == 1 || F == 6) vo = (lane >> 3) * row_stride + (lane & 7) * 16;
  else if (F == 2 || F == 4) vo = lane * 16;
  else if (F == 3) vo = lane * 4;
  else vo = lane * row_stride;
  const unsigned wbase = ((blockIdx.x * 4 + wave) * 8 * row_stride * 8) % (region / 2);
  f32x4 accv = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned so = __builtin_amdgcn_readfirstlane((wbase + (unsigned)((r * 8 + i) & 63) * 128u) % (region / 2));
      if (F <= 3) {
        const unsigned m = __builtin_amdgcn_readfirstlane(lbase + i * 1024);
        if (F == 3)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(m), "v"(vo), "s"(rs), "s"(so) : "memory");
        else
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(vo), "s"(rs), "s"(so) : "memory");
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src) + so + vo);
        accv += v;
      }
    }
    if (F <= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (accv[0] + accv[1] + accv[2] + accv[3] == 12345.f) out[0] = accv[0];
  if (F <= 3 && lds[wave][0][lane] == 12345.f) out[1] = 1.f;
}
template <int F>
static void run(const float* src, float* out, unsigned region, int cus) {
  const int rounds = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<F>, dim3(cus * 3), dim3(256), 0, 0, src, out, 10, region);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<F>, dim3(cus * 3), dim3(256), 0, 0, src, out, rounds, region);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)cus * 3 * 4 * rounds * 8;
  const double bytes = instr * (F == 3 ? 256.0 : 1024.0);
  printf("flavour %d: %.2f ms, %.1f clk per wave instruction per CU (2.4 GHz), %.1f B/clk/CU, %.2f TB/s aggregate\n", F, ms,
         ms * 1e-3 * 2.4e9 / (instr / cus), bytes / cus / (ms * 1e-3 * 2.4e9), bytes / ms / 1e9);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const unsigned region = 64u << 20;
  float* src; hipMalloc(&src, region); hipMemset(src, 0, region);
  float* out; hipMalloc(&out, 16);
  run<0>(src, out, region, cus); run<1>(src, out, region, cus); run<2>(src, out, region, cus); run<3>(src, out, region, cus);
  run<4>(src, out, region, cus); run<5>(src, out, region, cus); run<6>(src, out, region, cus);
  return 0;
}

// How fast do key rows enter LDS, and do the two ways of getting them there ADD UP?  One workgroup per CU (one wave per SIMD, as pass 1 of
// the two-pass label propagation runs), every wave fills its own ring of 8 KB stages from a stream of 128-byte lines (8 consecutive
// lanes = one line, 64 lanes x 16 B = 8 lines per instruction, as lp2_score_kernel's pieces):
//   D pieces of a stage by LDS-DMA (buffer_load_dwordx4 ... lds), R = 8 - D pieces by buffer_load_dwordx4 -> VGPR -> ds_write_b128,
//   RING - 1 stages in flight, the landed stage is "consumed" by one ds_read_b128 per lane.
// Streams: hot (every wave re-reads the same 1 MB: L2 hits) or cold (each wave walks its own 16 MB region: HBM / MALL).
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/probe_lds_fill tools/probe_lds_fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define SB 8192
template <int D, int RING>
__global__ __launch_bounds__(256, 1) void fill(const unsigned char* src, float* out, int stages, unsigned span) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[4][RING][SB];
  __shared__ unsigned char ballast[160 * 1024 - 4 * RING * SB - 1024];      // one workgroup per CU
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (stages < 0) ballast[threadIdx.x] = 1;
  // hot: every wave walks the SAME 1 MB (L2 hits); cold: each wave its own 16 MB region of the 1 GB buffer
  const unsigned long long base = (unsigned long long)(src + (span <= (1u << 20) ? 0 : ((size_t)(blockIdx.x * 4 + wave) * span) % (size_t)(1u << 30)));
  const u32x4 rs = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, span, 0x00020000u};
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, span, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  u32x4 rv[RING][8 - D > 0 ? 8 - D : 1];
  auto issue = [&](int st) {      // stage st -> slot st % RING
    const unsigned off = ((unsigned)st * SB) % span + lane * 16;
    unsigned char* dst = &ring[wave][st % RING][0];
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)dst);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (p < D) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off + p * 1024), "s"(lds_addr + p * 1024), "s"(rs) : "memory");
      } else {
        rv[st % RING][p - D] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off + p * 1024, 0, 0);
      }
    }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < RING - 1; ++d) issue(d);
  for (int s0 = 0; s0 < stages; s0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {      // (slots are compile-time constants: the staged registers are indexed statically)
      const int st = s0 + u;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (RING - 2)) : "memory");
      unsigned char* slot = &ring[wave][u][0];
#pragma unroll
      for (int p = D; p < 8; ++p) *reinterpret_cast<u32x4*>(slot + p * 1024 + lane * 16) = rv[u][p - D];
      const f32x4 v = *reinterpret_cast<const f32x4*>(slot + ((lane * 16 + st * 64) & (SB - 16)));
      acc += v;
      // refill the slot the PREVIOUS stage used
      const int nxt = st + RING - 1;
      const int ps = (u + RING - 1) % RING;
      {
        const unsigned off = ((unsigned)nxt * SB) % span + lane * 16;
        unsigned char* dst = &ring[wave][ps][0];
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)dst);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          if (p < D) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off + p * 1024), "s"(lds_addr + p * 1024), "s"(rs) : "memory");
          } else {
            rv[ps][p - D] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off + p * 1024, 0, 0);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <int D, int RING>
static void run(const unsigned char* src, float* out, int cus, unsigned span, const char* what) {
  const int stages = 3000 / RING * RING;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill<D, RING>), dim3(cus), dim3(256), 0, 0, src, out, 30, span);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((fill<D, RING>), dim3(cus), dim3(256), 0, 0, src, out, stages, span);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)cus * 4 * stages * SB;
  printf("%-4s ring %d, %d DMA + %d register pieces per stage: %6.1f GB/s per CU (%.2f TB/s, %.2f ms)\n", what, RING, D, 8 - D,
         bytes / cus / ms / 1e6, bytes / ms / 1e9, ms);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned char* src; hipMalloc(&src, (size_t)1 << 30); hipMemset(src, 1, (size_t)1 << 30);
  float* out; hipMalloc(&out, 4);
  const unsigned hot = 1u << 20, cold = 1u << 24;
  for (int pass = 0; pass < 2; ++pass) {
    const unsigned span = pass ? cold : hot; const char* w = pass ? "cold" : "hot";
    run<8, 3>(src, out, cus, span, w); run<6, 3>(src, out, cus, span, w); run<4, 3>(src, out, cus, span, w); run<2, 3>(src, out, cus, span, w);
    run<0, 3>(src, out, cus, span, w); run<8, 4>(src, out, cus, span, w); run<4, 4>(src, out, cus, span, w); run<0, 4>(src, out, cus, span, w);
  }
  return 0;
}

// What does an LDS-DMA piece (buffer_load_dwordx4 ... offen lds) write for a lane whose offset lies PAST the buffer's
// num_records?  (A register load returns zeros; the halo kernels' DMA patch relies on the LDS form writing zeros too.)
// LDS is pre-filled with 0xAB; even lanes fetch in range, odd lanes out of range (0x80000000 + soffset).
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/probe_dma_oob tools/probe_dma_oob.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void probe(const unsigned* src, unsigned bytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0xABABABABu;
  __syncthreads();
  const unsigned long long base = (unsigned long long)src;
  const u32x4 rs = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, bytes, 0x00020000u};
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds);
  const unsigned off = (lane & 1) ? 0x80000000u : lane * 16u;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(off), "s"(lds_addr), "s"(rs), "s"(128u) : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *src, *out, h[256], hs[1024];
  for (int i = 0; i < 1024; ++i) hs[i] = 0x10000u + i;
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(src, 4096, out);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  int zeros = 0, kept = 0, other = 0, good = 0;
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < 4; ++k) {
      const unsigned v = h[l * 4 + k];
      if (l & 1) { if (v == 0) ++zeros; else if (v == 0xABABABABu) ++kept; else ++other; }
      else if (v == 0x10000u + 32 + l * 4 + k) ++good;
    }
  printf("in-range dwords correct: %d / 128; out-of-range lanes: %d dwords ZERO, %d dwords UNTOUCHED, %d other\n", good, zeros, kept, other);
  return 0;
}

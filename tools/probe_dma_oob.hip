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
// Round 6 (advisor r05): is the SCALAR offset part of the range check?  The deep halo schedule and the wgrad ring send pieces out of
// range through soffset (an in-range voffset + soffset >= num_records).  mode 1: voffset = 16 lane, soffset = 0x80000000, num_records
// 4096 - the buffer behind the descriptor really is > 2 GiB long and holds a marker at base + 2 GiB, so a check that ignored soffset
// would fetch the marker (no fault either way).  mode 2: voffset = 16 lane, soffset = 4096 - 512: lanes >= 32 end up just past
// num_records.
__global__ void probe_soff(const unsigned* src, unsigned bytes, unsigned soff, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0xABABABABu;
  __syncthreads();
  const unsigned long long base = (unsigned long long)src;
  const u32x4 rs = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, bytes, 0x00020000u};
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds);
  const unsigned off = lane * 16u;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(off), "s"(lds_addr), "s"(rs), "s"(soff) : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
static void run_soff(unsigned* big, unsigned* out, unsigned soff, const char* what) {
  unsigned h[256];
  probe_soff<<<1, 64>>>(big, 4096, soff, out);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  int zeros = 0, kept = 0, marker = 0, inrange = 0, other = 0;
  for (int i = 0; i < 256; ++i) {
    if (h[i] == 0) ++zeros; else if (h[i] == 0xABABABABu) ++kept; else if (h[i] == 0x5EEDBEEFu) ++marker;
    else if ((h[i] & 0xFFFF0000u) == 0x10000u) ++inrange; else ++other;
  }
  printf("%s: %d dwords ZERO, %d UNTOUCHED, %d fetched from base + soffset past num_records (MARKER), %d in-range data, %d other\n", what, zeros, kept, marker, inrange, other);
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
  unsigned* big;
  const size_t two_gib = 1ull << 31;
  if (hipMalloc(&big, two_gib + 65536) != hipSuccess) { printf("cannot allocate 2 GiB\n"); return 1; }
  hipMemcpy(big, hs, 4096, hipMemcpyHostToDevice);
  unsigned mk[1024];
  for (int i = 0; i < 1024; ++i) mk[i] = 0x5EEDBEEFu;
  hipMemcpy((char*)big + two_gib, mk, 4096, hipMemcpyHostToDevice);
  hipMemcpy((char*)big + 4096, mk, 4096, hipMemcpyHostToDevice);
  run_soff(big, out, 0x80000000u, "soffset 0x80000000, voffset in range (expect 256 ZERO if soffset is range-checked)");
  run_soff(big, out, 4096u - 512u, "soffset num_records - 512, voffset 16 lane (expect 128 in-range + 128 ZERO if soffset is range-checked)");
  return 0;
}

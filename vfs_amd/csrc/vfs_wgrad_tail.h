// In-launch split-K reduction of the weight-gradient kernels (round 6).
//
// Rounds 1-5: every weight-gradient launch wrote fp32 partials [nsplit][Cout][Ktot] and a second launch (wgrad_reduce) summed them
// into the gradient: 58 extra launches per ResNet-50 step on the side stream.  Here the LAST workgroup of a (k-column, cout) tile to
// arrive - a device-scope ticket per tile, the pattern of the BatchNorm reductions (bn.hip) - sums the tile's partials in ASCENDING
// SPLIT ORDER (deterministic whoever arrives last; its own partial is re-read like the others so that the order never depends on
// the arrival) and adds the sum to the gradient in the reference's OIHW layout.
//
// Hand-over between workgroups that sit on different XCDs (each XCD has its own L2): the partials are written with `sc1`
// (device-scope) buffer stores and read back with `sc1` buffer loads - the instructions a relaxed agent-scope atomic compiles to,
// 16 bytes per lane, whole lines per wave (layout below) - and the ticket is drawn after `s_waitcnt vmcnt(0)` + a workgroup barrier (vfs_release_workgroup: the stores
// of every wave have been acknowledged by the memory side).  No __threadfence(): its L2 write-back of megabytes of unrelated
// dirty lines tripled the BatchNorm reductions (vfs_common.h).
// tickets: unsigned[VFS_WGRAD_TICKETS], zero before the first launch; every launch leaves them at zero.
#pragma once
#include "vfs_conv.h"

#define VFS_WGRAD_TICKETS 4096
#define VFS_SC1 16      // cache-policy operand of the raw buffer builtins: sc1 = device scope

// Layout of the partials in this mode: NOT [split][Cout][Ktot] (the layout wgrad_reduce reads) but the REGISTER IMAGE of the
// workgroups, float4 [split][tile][piece i][thread t]: piece i of all 256 threads is 4 KB contiguous, so every wave-wide `sc1` store
// writes - and every `sc1` load of the tail fetches - eight whole 128-byte lines.  (First version, round 6: the [Cout][Ktot] layout
// with sc1 accesses - a wave-wide store scattered 64 separate 16-byte pieces that write THROUGH the L2 one by one and the 3x3 kernel
// went from 59 to 830 us.)  A tile is NI pieces x 256 threads x 16 bytes; the workspace holds nsplit x ntiles of them
// (= nsplit x Cout x Ktot floats, Ktot rounded up to 128 for the generic kernel's half-empty tiles).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wgt_partial_rsrc(const WgradArgs& a, int ntiles, int NI) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)a.partial, 0, (unsigned)((size_t)a.nsplit * ntiles * NI * 4096), 0x00020000);
}
// byte offset of piece i of this thread in (split, tile)
__device__ __forceinline__ unsigned wgt_piece_off(int split, int tile, int ntiles, int NI, int i) {
  return (unsigned)(((((size_t)split * ntiles + tile) * NI + i) * 256 + threadIdx.x) * 16);
}
__device__ __forceinline__ void wgt_store_piece(const __amdgpu_buffer_rsrc_t& prs, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs, off, 0, VFS_SC1);
}
// all threads of the workgroup, after their wgt_store calls: true in the workgroup that completes tile `tile`
__device__ __forceinline__ bool wgt_last_arriver(const WgradArgs& a, int tile) {
  __shared__ unsigned s_wgt_last;
  vfs_release_workgroup();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tk = vfs_ticket_agent(a.tickets + tile);
    const bool last = tk == (unsigned)a.nsplit - 1u;
    if (last) vfs_store_agent(a.tickets + tile, 0u);      // ready for the next launch (nobody else touches this ticket any more)
    s_wgt_last = last ? 1u : 0u;
  }
  __syncthreads();
  return s_wgt_last != 0u;
}
// sum of NP pieces of this thread over all splits, ascending split order; off[i] = byte offset of piece i in split 0 (WGT_SKIP: no such
// piece - the lane's voffset lies past num_records and the load returns zeros; the split offset travels in the voffset as well, never
// in the scalar offset: tools/probe_dma_oob.hip only vouches for the range check of the former).  Four splits per round: 4 NP
// independent loads in flight per lane.
#define WGT_SKIP 0xFFFFFFFFu
template <int NP>
__device__ __forceinline__ void wgt_sum_splits(const WgradArgs& a, const __amdgpu_buffer_rsrc_t& prs, unsigned sstride, const unsigned (&off)[NP], f32x4 (&sum)[NP]) {
#pragma unroll
  for (int i = 0; i < NP; ++i) sum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= a.nsplit; s += 4) {
    u32x4 v[4][NP];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < NP; ++i)
        v[q][i] = __builtin_amdgcn_raw_buffer_load_b128(prs, off[i] == WGT_SKIP ? WGT_SKIP : off[i] + (unsigned)(s + q) * sstride, 0, VFS_SC1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < NP; ++i) sum[i] += __builtin_bit_cast(f32x4, v[q][i]);
  }
  for (; s < a.nsplit; ++s) {
    u32x4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, off[i] == WGT_SKIP ? WGT_SKIP : off[i] + (unsigned)s * sstride, 0, VFS_SC1);
#pragma unroll
    for (int i = 0; i < NP; ++i) sum[i] += __builtin_bit_cast(f32x4, v[i]);
  }
}
// grad (+=) of four consecutive k-columns kc..kc+3 of output channel cout, reference layout OIHW:
// k = (r KW + s) Cin + cin -> grad[cout][cin][r][s]   (1x1: the four floats are contiguous)
__device__ __forceinline__ void wgt_add_grad(const WgradArgs& a, int cout, int kc, f32x4 v) {
  const int Cin = a.g.C, taps = a.g.KH * a.g.KW;
  if (taps == 1) {
    f32x4* p = reinterpret_cast<f32x4*>(a.grad + (size_t)cout * Cin + kc);
    *p = *p + v;
    return;
  }
  const int tap = kc / Cin, cin = kc - tap * Cin;      // Cin % 4 == 0: the four columns share the tap
  float* p = a.grad + ((size_t)cout * Cin + cin) * taps + tap;
#pragma unroll
  for (int e = 0; e < 4; ++e) p[e * taps] += v[e];
}

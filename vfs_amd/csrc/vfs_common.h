// Shared device helpers for the vfs_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all activation / packed-weight storage

typedef __attribute__((ext_vector_type(8))) short bf16x8;      // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;       // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16-byte memory vector
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;  // 8-byte memory vector

#define VFS_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __builtin_bit_cast(float, (uint32_t)v << 16);
}
__device__ __forceinline__ float bflo(uint32_t packed) {  // low bf16 of a dword
  return __builtin_bit_cast(float, packed << 16);
}
__device__ __forceinline__ float bfhi(uint32_t packed) {  // high bf16 of a dword
  return __builtin_bit_cast(float, packed & 0xffff0000u);
}
// round-to-nearest-even fp32 -> bf16: the compiler lowers the __bf16 vector conversion to ONE
// v_cvt_pk_bf16_f32 per pair on gfx950 (tools/probe_cvt.hip: identical to the integer
// "+0x7fff+lsb" rounding for all 2^32 non-NaN patterns; the emulator build converts in software)
typedef __attribute__((ext_vector_type(2))) float vfs_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 vfs_bf16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const vfs_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vfs_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane: four v_add_f32 with a
// DPP source operand (xor-1 / xor-2 quad permutes, row_half_mirror, row_mirror).  VALU only - unlike
// __shfl_xor (ds_bpermute_b32) there is no LDS round trip and no lgkmcnt wait per step.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);   // row_half_mirror
  x += dpp_mov<0x140>(x);   // row_mirror
  return x;
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

// LDS-DMA (buffer_load_dwordx4 ... lds): 16 bytes per lane from a raw buffer straight into LDS at
// lds_wave_base + 16*lane (wave-uniform base), no VGPR staging and no ds_write pass.  Issued from
// inline asm on purpose: through the builtin hipcc parks a `s_waitcnt vmcnt(0)` in front of the next
// LDS read (it cannot prove the read does not alias the transfer), which serialises the very latency
// the transfer is supposed to hide.  Untracked by the compiler, so the issuing wave must call
// vfs_dma_wait_all() and pass a barrier before ANY wave reads the destination (MI355X guide,
// "LDS-DMA data is ordered for a ds_read only by the issuing waves' vmcnt followed by a barrier").
// The compiler's own vmcnt bookkeeping stays safe: extra loads in the in-order queue only make its
// counted waits stricter.  (tests/emu supplies host versions of these three.)
#ifndef VFS_EMU
typedef u32x4 vfs_rsrc_words;
__device__ __forceinline__ vfs_rsrc_words vfs_make_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return (vfs_rsrc_words){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void vfs_dma16_async(vfs_rsrc_words rsrc, void* lds_wave_base, unsigned voffset, unsigned soffset) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds_wave_base);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc), "s"(soffset)
      : "memory");
}
// the same with the LDS byte address given as a wave-uniform 32-bit value (one s_add per piece instead of a generic-pointer
// conversion with its null check) and m0 declared clobbered instead of saved and restored: 4 instructions per piece instead of 10
typedef unsigned vfs_lds_t;
__device__ __forceinline__ vfs_lds_t vfs_lds_addr(const void* lds) {
  typedef __attribute__((address_space(3))) const void* lds_cptr;
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)lds);
}
// (m0 is a reserved register: clang warns that it cannot promise to keep a clobbered one - nothing in these kernels lives in it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void vfs_dma16_async_at(vfs_rsrc_words rsrc, vfs_lds_t lds_addr, unsigned voffset, unsigned soffset) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, %3 offen lds"
      :
      : "v"(voffset), "s"(lds_addr), "s"(rsrc), "s"(soffset)
      : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void vfs_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's vector-memory operations (DMA pieces) are still in flight
template <int N>
__device__ __forceinline__ void vfs_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#else
template <int N>
inline void vfs_dma_wait() {}
#endif

// Device-coherent (agent scope) relaxed accesses for small inter-workgroup hand-offs: sc1 stores / loads
// that bypass the non-coherent per-XCD L2 state, WITHOUT the L2 write-back + invalidate a full
// __threadfence() costs on gfx950 (measured: a 512-workgroup reduction went from 17 to 50 us with
// fences while megabytes of dirty conv output sat in the L2).  vfs_release_workgroup() makes the wave
// wait until its own stores have been performed (acknowledged).  (Plain HIP intrinsics: tests/emu/hip/hip_runtime.h defines __hip_atomic_* / the
// fences for the host build, so nothing here is conditional.)
__device__ __forceinline__ void vfs_store_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double vfs_load_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vfs_store_agent(float* p, f32x4 v) {
  __hip_atomic_store(p, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 2, v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 3, v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 vfs_load_agent4(const float* p) {
  f32x4 v;
  v[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v[1] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v[2] = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v[3] = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
__device__ __forceinline__ void vfs_store_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float vfs_load_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vfs_store_agent(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned vfs_ticket_agent(unsigned* p) {
  return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Round 3 FIX: the workgroup-scope release fence alone compiles to NO wait for vector-memory stores (in the non-tgsplit memory
// model the waves of a workgroup share their CU's L1, so workgroup scope needs none): the ISA was `global_store ... sc1`, `s_barrier`,
// `global_atomic_add` - the ticket could be performed at the L2 while a chunk sum was still in flight, and the workgroup that drew
// the last ticket could read a stale sum (seen as a wrong loss in ~1 of 30 eager ResNet-18 full-size runs,
// test_train_step_properties_r18_full_size).  `s_waitcnt vmcnt(0)` first: the sc1 stores of this wave have been acknowledged by
// the memory side before its threads reach the barrier in front of the ticket.
__device__ __forceinline__ void vfs_release_workgroup() {
  vfs_dma_wait_all();      // s_waitcnt vmcnt(0)  (host emulation: nothing to wait for)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
// System scope (other GPUs over xGMI, other processes through hipIpc mappings; csrc/p2p.hip): accesses that are performed
// at the memory, not in this GPU's caches.
__device__ __forceinline__ void vfs_store_system(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double vfs_load_system(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void vfs_store_system_release(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long vfs_load_system_acquire(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void vfs_fence_system() { __threadfence_system(); }
__device__ __forceinline__ void vfs_spin_pause() { __builtin_amdgcn_s_sleep(8); }

__device__ __forceinline__ u32x4 zero16() {
  u32x4 z = {0u, 0u, 0u, 0u};
  return z;
}

// unpack 8 bf16 (one 16-byte vector) to fp32
__device__ __forceinline__ void unpack8(u32x4 v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// Bit-packed ReLU mask of 8 consecutive bf16 values (bit i <-> value i > 0, exactly the test `float(v) > 0.f`: positive
// zero, negative values and NaNs give 0).  A residual unit's output y is only needed as `y > 0` by its BatchNorm backward:
// two passes read it (statistics, apply); the mask is 1/16 of the bytes.
__device__ __forceinline__ unsigned mask8_of(u32x4 v) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned bits = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bits |= (((w[i] & 0xffffu) - 1u) < 0x7f80u ? 1u : 0u) << (2 * i);
    bits |= (((w[i] >> 16) - 1u) < 0x7f80u ? 1u : 0u) << (2 * i + 1);
  }
  return bits;
}
// Layout: SLAB-major, uint8 [C/64][M][8] (one slab [M][C/8] when C < 64): the byte of pixel m, channels c..c+7 (c % 8 == 0).
// Every kernel that touches the mask owns <= 64 channels of a range of pixels, so its bytes are contiguous; with a plain
// [M][C/8] layout a workgroup used 8 bytes of every 64-byte line it fetched and the mask cost as much as the activation.
__device__ __forceinline__ size_t mask8_index(long long m, int c, long long M, int C) {
  if (C < 64) return (size_t)m * (C >> 3) + (c >> 3);
  return ((size_t)(c >> 6) * M + m) * 8 + ((c & 63) >> 3);
}
__device__ __forceinline__ unsigned mask8_load(const void* bits, long long m, int c, long long M, int C) {
  return reinterpret_cast<const unsigned char*>(bits)[mask8_index(m, c, M, C)];
}
#define VFS_MASK_BITS 2     // value of the `relu` argument of the BatchNorm-backward entry points: `y` is a bit-packed mask

// error codes of the C ABI (include/vfs_hip.h)
#define VFS_OK 0
#define VFS_ERR_SHAPE (-1)
#define VFS_ERR_LAUNCH (-2)
#define VFS_ERR_ARG (-3)

int vfs_set_error(int code, const char* msg);  // capi.cpp
int vfs_check_launch(const char* what);        // capi.cpp

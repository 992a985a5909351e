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

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

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

// error codes of the C ABI (include/vfs_hip.h)
#define VFS_OK 0
#define VFS_ERR_SHAPE (-1)
#define VFS_ERR_LAUNCH (-2)
#define VFS_ERR_ARG (-3)

int vfs_set_error(int code, const char* msg);  // capi.cpp
int vfs_check_launch(const char* what);        // capi.cpp

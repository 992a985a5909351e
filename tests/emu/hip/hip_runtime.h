// CPU emulation of the HIP subset the vfs_amd kernels use -- TEST INFRASTRUCTURE.
//
// The kernels under vfs_amd/csrc are compiled UNMODIFIED by host clang with
// `-I tests/emu` so that `#include <hip/hip_runtime.h>` resolves here.  One OS
// thread runs one workgroup at a time; the workgroup's threads are ucontext
// fibers scheduled round-robin, yielding at __syncthreads() and at wave-level
// collectives (shuffles, ballots, MFMA).  MFMA builtins are emulated from the
// documented gfx950 fragment layouts (cdna_hip_programming.md section 3):
//   16x16x32 bf16: A[i][k] in lane i+16*(k/8) elem k%8, B[k][j] in lane
//   j+16*(k/8) elem k%8, D[4*(lane>>4)+r][lane&15] in reg r;
//   32x32x16 bf16: A[i][k] in lane i+32*(k/8), D[(r&3)+8*(r>>2)+4*(lane>>5)][lane&31].
// This checks index math, LDS layouts, epilogues and reductions on the CPU; it
// does not model timing, bank conflicts or the memory model.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define VFS_EMU 1

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
  memcpy(d, s, n);
  return hipSuccess;
}
static const int hipMemcpyDeviceToDevice = 3;
// the host calls of csrc/p2p.hip: a "window" is plain host memory, an IPC handle carries the pointer itself
struct hipIpcMemHandle_t { char reserved[64]; };
static const unsigned hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1;
inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 2; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof(*h)); memcpy(h, &p, sizeof(p)); return hipSuccess; }
inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, &h, sizeof(*p)); return hipSuccess; }
inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
static const int hipDeviceAttributeMultiprocessorCount = 63;
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 8; return hipSuccess; }   // "8 CUs"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

namespace emu {

// Fiber switch.  x86-64: a dozen instructions (callee-saved registers + the stack pointer) - glibc's swapcontext also saves and
// restores the signal mask with a system call per switch, which was more than half of the emulated kernels' run time.  Other
// hosts keep ucontext.
#if defined(__x86_64__)
#define EMU_FAST_SWITCH 1
__attribute__((naked, noinline)) static void ctx_switch(void** /*save sp here: rdi*/, void* /*continue on this sp: rsi*/) {
  asm volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\t"
      "movq %rsi, %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
      "ret");
}
#else
#define EMU_FAST_SWITCH 0
#endif

struct Fiber {
#if EMU_FAST_SWITCH
  void* sp = nullptr;
#else
  ucontext_t ctx;
#endif
  char* stack = nullptr;
  dim3 tid;
  int lin = 0;
  bool done = false;
};

struct WaveX {  // per-wave exchange area
  uint32_t u32[64];
  uint32_t x32[2][64];      // shfl_u32: two generations
  uint64_t u64[64];
  short a[64][8];
  short b[64][8];
  int count = 0;
  int gen = 0;
};

struct BlockCtx {
  std::vector<Fiber> fibers;
#if EMU_FAST_SWITCH
  void* sched = nullptr;
#else
  ucontext_t sched;
#endif
  int cur = 0;
  int nthreads = 0;
  int bar_count = 0, bar_gen = 0;
  std::vector<WaveX> waves;
  dim3 bid, bdim, gdim;
  std::function<void()> body;
};

inline thread_local BlockCtx* B = nullptr;

#if EMU_FAST_SWITCH
inline void yield() { ctx_switch(&B->fibers[B->cur].sp, B->sched); }
#else
inline void yield() { swapcontext(&B->fibers[B->cur].ctx, &B->sched); }
#endif

inline void block_barrier() {
  int gen = B->bar_gen;
  if (++B->bar_count == B->nthreads) {
    B->bar_count = 0;
    B->bar_gen++;
  } else {
    while (B->bar_gen == gen) yield();
  }
}

inline int lane_id() { return B->fibers[B->cur].lin & 63; }
inline int wave_id() { return B->fibers[B->cur].lin >> 6; }
inline int wave_size(int w) { return std::min(64, B->nthreads - w * 64); }

inline void wave_barrier() {
  WaveX& W = B->waves[wave_id()];
  int gen = W.gen;
  if (++W.count == wave_size(wave_id())) {
    W.count = 0;
    W.gen++;
  } else {
    while (W.gen == gen) yield();
  }
}

static void fiber_entry() {
  B->body();
  B->fibers[B->cur].done = true;
  yield();      // never resumed
#if EMU_FAST_SWITCH
  __builtin_trap();
#endif
}

inline void run_block(BlockCtx& ctx) {
  B = &ctx;
  const size_t STK = 256 * 1024;
  int n = ctx.nthreads;
  for (int i = 0; i < n; ++i) {
    Fiber& f = ctx.fibers[i];
    f.done = false;
    if (!f.stack) f.stack = (char*)malloc(STK);
#if EMU_FAST_SWITCH
    // first switch: six zeroed callee-saved registers are popped, then `ret` enters fiber_entry with the stack as after a call
    uintptr_t top = ((uintptr_t)f.stack + STK) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    sp[6] = (void*)fiber_entry;
    sp[7] = nullptr;
    f.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STK;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
  }
  ctx.bar_count = 0;
  for (auto& w : ctx.waves) w.count = 0;
  int remaining = n;
  while (remaining > 0) {
    int progressed = 0;
    for (int i = 0; i < n; ++i) {
      if (ctx.fibers[i].done) continue;
      ctx.cur = i;
#if EMU_FAST_SWITCH
      ctx_switch(&ctx.sched, ctx.fibers[i].sp);
#else
      swapcontext(&ctx.sched, &ctx.fibers[i].ctx);
#endif
      if (ctx.fibers[i].done) --remaining;
      ++progressed;
    }
    if (!progressed) break;
  }
}

template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, Args... args) {
  size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  int nthreads = block.x * block.y * block.z;
  unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  unsigned nworkers = (unsigned)std::min<size_t>(hw, nblocks);
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    BlockCtx ctx;
    ctx.nthreads = nthreads;
    ctx.fibers.resize(nthreads);
    ctx.waves.resize((nthreads + 63) / 64);
    ctx.bdim = block;
    ctx.gdim = grid;
    for (int i = 0; i < nthreads; ++i) {
      ctx.fibers[i].lin = i;
      ctx.fibers[i].tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    }
    ctx.body = [&]() { kernel(args...); };
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      ctx.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / ((size_t)grid.x * grid.y));
      run_block(ctx);
    }
    for (auto& f : ctx.fibers) free(f.stack);
    B = nullptr;
  };
  if (nworkers <= 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nworkers; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
}

// ---- wave collectives -----------------------------------------------------
inline uint32_t shfl_u32(uint32_t v, int src) {
  // ONE barrier per exchange: the slot array alternates with the barrier generation (the same for every lane until the last one
  // arrives), so the next exchange writes the other array; the one after it can only start once every lane has entered the next
  // exchange, i.e. has finished reading this one.  (Shuffles and DPP moves dominate the emulated run time of the wave-level kernels.)
  WaveX& W = B->waves[wave_id()];
  const int par = W.gen & 1;
  W.x32[par][lane_id()] = v;
  wave_barrier();
  return W.x32[par][src & 63];
}
inline uint64_t ballot(int pred) {
  WaveX& W = B->waves[wave_id()];
  W.u32[lane_id()] = pred ? 1u : 0u;
  wave_barrier();
  uint64_t m = 0;
  for (int i = 0; i < wave_size(wave_id()); ++i) m |= (uint64_t)(W.u32[i] & 1) << i;
  wave_barrier();
  return m;
}
template <typename T>
inline T shfl_t(T v, int src) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "shfl size");
  if (sizeof(T) == 4) {
    uint32_t u;
    memcpy(&u, &v, 4);
    u = shfl_u32(u, src);
    T r;
    memcpy(&r, &u, 4);
    return r;
  } else {
    uint64_t u;
    memcpy(&u, &v, 8);
    uint32_t lo = shfl_u32((uint32_t)u, src), hi = shfl_u32((uint32_t)(u >> 32), src);
    u = ((uint64_t)hi << 32) | lo;
    T r;
    memcpy(&r, &u, 8);
    return r;
  }
}

inline float bf16_to_f32(short s) {
  uint32_t u = ((uint32_t)(uint16_t)s) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

typedef __attribute__((ext_vector_type(8))) short v8s;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(16))) float v16f;

inline v4f mfma_16x16x32_bf16(v8s a, v8s b, v4f c) {
  WaveX& W = B->waves[wave_id()];
  int l = lane_id();
  for (int i = 0; i < 8; ++i) {
    W.a[l][i] = a[i];
    W.b[l][i] = b[i];
  }
  wave_barrier();
  v4f d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k)
      acc = fmaf(bf16_to_f32(W.a[row + 16 * (k >> 3)][k & 7]), bf16_to_f32(W.b[col + 16 * (k >> 3)][k & 7]), acc);
    d[r] = acc;
  }
  wave_barrier();
  return d;
}

inline v16f mfma_32x32x16_bf16(v8s a, v8s b, v16f c) {
  WaveX& W = B->waves[wave_id()];
  int l = lane_id();
  for (int i = 0; i < 8; ++i) {
    W.a[l][i] = a[i];
    W.b[l][i] = b[i];
  }
  wave_barrier();
  v16f d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k)
      acc = fmaf(bf16_to_f32(W.a[row + 32 * (k >> 3)][k & 7]), bf16_to_f32(W.b[col + 32 * (k >> 3)][k & 7]), acc);
    d[r] = acc;
  }
  wave_barrier();
  return d;
}

// v_mfma_f32_32x32x2_f32 (f32 in): A[i][k] in lane i + 32*k, B[k][j] in lane j + 32*k, one VGPR each;
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) - bitwise a k-ordered fmaf chain (cdna_hip_programming.md section 3)
inline v16f mfma_32x32x2_f32(float a, float b, v16f c) {
  WaveX& W = B->waves[wave_id()];
  int l = lane_id();
  memcpy(&W.u32[l], &a, 4);
  uint32_t bu;
  memcpy(&bu, &b, 4);
  W.u64[l] = bu;
  wave_barrier();
  v16f d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, &W.u32[row + 32 * k], 4);
      uint32_t t32 = (uint32_t)W.u64[col + 32 * k];
      memcpy(&bv, &t32, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  wave_barrier();
  return d;
}

}  // namespace emu

#define threadIdx (emu::B->fibers[emu::B->cur].tid)
#define blockIdx (emu::B->bid)
#define blockDim (emu::B->bdim)
#define gridDim (emu::B->gdim)
#define warpSize 64

inline void __syncthreads() { emu::block_barrier(); }
#define __builtin_amdgcn_s_barrier() emu::block_barrier()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
// lanes are fibers here, not lockstep: the (code-free on hardware) wave barrier is a real rendezvous
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)

template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  int l = emu::lane_id();
  return emu::shfl_t(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  return emu::shfl_t(v, emu::lane_id() ^ mask);
}
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = emu::lane_id();
  int src = l + (int)d;
  if ((src & ~(width - 1)) != (l & ~(width - 1))) src = l;
  return emu::shfl_t(v, src);
}
template <typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = emu::lane_id();
  int src = l - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (l & ~(width - 1))) src = l;
  return emu::shfl_t(v, src);
}
inline unsigned long long __ballot(int p) { return emu::ballot(p); }
inline int __any(int p) { return emu::ballot(p) != 0; }
inline int __all(int p) {
  int n = emu::wave_size(emu::wave_id());
  unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
  return emu::ballot(p) == full;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll(x); }

#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma_16x16x32_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_readfirstlane(x) __shfl((x), 0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, sync) ((void)0)   // an instruction-scheduling hint: nothing to emulate
// v_mov_b32 with a DPP control (gfx9 encodings): quad_perm 0x00-0xFF, row_shl 0x101-0x10F,
// row_shr 0x111-0x11F, row_ror 0x121-0x12F, row_mirror 0x140, row_half_mirror 0x141.
// All lanes active, full row/bank masks (the only form the kernels use).
namespace emu {
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)row_mask; (void)bank_mask;
  const int l = lane_id(), row = l & ~15, i = l & 15;
  int from = -1;
  if (ctrl >= 0 && ctrl <= 0xFF) from = row | (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) from = (i + (ctrl & 15) < 16) ? row | (i + (ctrl & 15)) : -1;
  else if (ctrl >= 0x111 && ctrl <= 0x11F) from = (i - (ctrl & 15) >= 0) ? row | (i - (ctrl & 15)) : -1;
  else if (ctrl >= 0x121 && ctrl <= 0x12F) from = row | ((i - (ctrl & 15)) & 15);
  else if (ctrl == 0x140) from = row | (15 - i);
  else if (ctrl == 0x141) from = row | (i & 8) | (7 - (i & 7));
  else { fprintf(stderr, "emu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
  const uint32_t got = shfl_u32((uint32_t)src, from < 0 ? l : from);
  if (from < 0) return bound_ctrl ? 0 : old;
  return (int)got;
}
}  // namespace emu
#define __builtin_amdgcn_update_dpp emu::update_dpp

// raw buffer descriptor + bounds-checked 16-byte load (out of range -> zeros, as the hardware)
namespace emu {
struct rsrc_t {
  const char* base;
  unsigned num_records;
};
inline rsrc_t make_buffer_rsrc(void* p, int stride, unsigned num, unsigned flags) {
  (void)stride; (void)flags;
  return rsrc_t{(const char*)p, num};
}
typedef __attribute__((ext_vector_type(4))) unsigned int v4u;
inline v4u raw_buffer_load_b128(rsrc_t r, unsigned voff, unsigned soff, int aux) {
  (void)aux;
  v4u v = {0u, 0u, 0u, 0u};
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 16 <= r.num_records) memcpy(&v, r.base + o, 16);
  return v;
}
}  // namespace emu
// ds_read_b64_tr_b16 (probed on gfx950): inside each 16-lane group the lanes' 8-byte chunks form a
// 64-element array F (lane 0's 4 elements first); lane i receives F[i], F[16+i], F[32+i], F[48+i]
namespace emu {
typedef __attribute__((ext_vector_type(4))) short v4s;
inline v4s ds_read_tr16_b64(uintptr_t p) {
  WaveX& W = B->waves[wave_id()];
  const int l = lane_id();
  W.u64[l] = (uint64_t)p;
  wave_barrier();
  const int g0 = l & ~15, i = l & 15;
  v4s v;
  for (int e = 0; e < 4; ++e) {
    const int f = e * 16 + i;
    v[e] = ((const short*)(uintptr_t)W.u64[g0 + (f >> 2)])[f & 3];
  }
  wave_barrier();
  return v;
}
}  // namespace emu
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu::ds_read_tr16_b64((uintptr_t)(p))
namespace emu {
inline unsigned raw_buffer_load_b32(rsrc_t r, unsigned voff, unsigned soff, int aux) {
  (void)aux;
  unsigned v = 0;
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 4 <= r.num_records) memcpy(&v, r.base + o, 4);
  return v;
}
inline void raw_buffer_store_b32(unsigned v, rsrc_t r, unsigned voff, unsigned soff, int aux) {
  (void)aux;
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 4 <= r.num_records) memcpy(const_cast<char*>(r.base) + o, &v, 4);      // out of range: dropped, as the hardware
}
}  // namespace emu
namespace emu {
inline void raw_buffer_store_b128(v4u v, rsrc_t r, unsigned voff, unsigned soff, int aux) {
  (void)aux;
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 16 <= r.num_records) memcpy(const_cast<char*>(r.base) + o, &v, 16);      // out of range: dropped, as the hardware
}
}  // namespace emu
#define __builtin_amdgcn_raw_buffer_store_b128 emu::raw_buffer_store_b128
#define __builtin_amdgcn_raw_buffer_load_b32 emu::raw_buffer_load_b32
#define __builtin_amdgcn_raw_buffer_store_b32 emu::raw_buffer_store_b32
#define __amdgpu_buffer_rsrc_t emu::rsrc_t
#define __builtin_amdgcn_make_buffer_rsrc emu::make_buffer_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128 emu::raw_buffer_load_b128
// LDS-DMA helpers of csrc/vfs_common.h (inline asm on the device): host versions.  Lane i's 16 bytes
// land at lds_wave_base + 16*i; the transfer completes immediately (no asynchrony to emulate).
#define VFS_EMU 1
typedef emu::rsrc_t vfs_rsrc_words;
inline vfs_rsrc_words vfs_make_rsrc_words(const void* p, unsigned bytes) { return emu::rsrc_t{(const char*)p, bytes}; }
inline void vfs_dma16_async(vfs_rsrc_words rsrc, void* lds_wave_base, unsigned voffset, unsigned soffset) {
  const emu::v4u v = emu::raw_buffer_load_b128(rsrc, voffset, soffset, 0);
  memcpy((char*)lds_wave_base + 16 * emu::lane_id(), &v, 16);
}
// LDS "addresses" of the host build: byte offsets are plain pointers carried in a uintptr_t
typedef uintptr_t vfs_lds_t;
inline vfs_lds_t vfs_lds_addr(const void* lds) { return (uintptr_t)lds; }
inline void vfs_dma16_async_at(vfs_rsrc_words rsrc, vfs_lds_t lds_addr, unsigned voffset, unsigned soffset) {
  vfs_dma16_async(rsrc, (void*)lds_addr, voffset, soffset);
}
inline void vfs_dma_wait_all() {}
// The HIP atomic intrinsics, scopes and fences csrc/vfs_common.h uses for inter-workgroup / inter-process hand-offs: blocks run on
// different host threads, two "ranks" of csrc/p2p.hip on two host threads of the test process.
// (__hip_atomic_load / _store / _fetch_add are clang builtins in host C++ too; the scope macros are only predefined in HIP mode -
// on the host every scope is the whole process)
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __builtin_amdgcn_s_sleep(n) std::this_thread::yield()

template <typename T>
inline T atomicAdd(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (!__atomic_compare_exchange_n(p, &old, old + v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
template <>
inline float atomicAdd<float>(float* p, float v) {
  uint32_t* ip = (uint32_t*)p;
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    uint32_t nu;
    memcpy(&nu, &f, 4);
    if (__atomic_compare_exchange_n(ip, &old, nu, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      float r;
      memcpy(&r, &old, 4);
      return r;
    }
  }
}
template <>
inline double atomicAdd<double>(double* p, double v) {
  uint64_t* ip = (uint64_t*)p;
  uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
  for (;;) {
    double f;
    memcpy(&f, &old, 8);
    f += v;
    uint64_t nu;
    memcpy(&nu, &f, 8);
    if (__atomic_compare_exchange_n(ip, &old, nu, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      double r;
      memcpy(&r, &old, 8);
      return r;
    }
  }
}

inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
using std::max;
using std::min;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

// The staging pipeline of labelprop_f32_kernel in isolation (no scoring / top-k): what does ONE structural change buy?
// Same geometry as the product kernel on a DAVIS res4 map (60 x 107, C = 1024, 21 key frames, radius 18): a workgroup =
// an 8x8 query tile x one key frame, 64-key blocks of the window, channels in stages of 32, one 32x32x2 MFMA tile per wave.
// Every structure computes the same fma chains: the checksums of V2.., coal, ring, ring3 must agree bit for bit (V0 / V1 zero-fill
// the rows past the window instead of clamping them; ring2 has a known race in super blocks of ONE chunk, its timings stand).
//   V0     the product kernel's loop (registers -> LDS stores -> barrier -> loads of the next stage -> operand reads + MFMAs -> barrier)
//   V1     V0 with the operand reads of MFMA pair p+1 issued before the MFMAs of pair p (__builtin_amdgcn_sched_group_barrier)
//   V2     V1 with unconditional loads (clamped rows; C % 32 == 0): no exec-mask branches around the loads
//   V3     V2 with TWO register stages in flight
//   coal   V2 with COALESCED loaders (8 consecutive lanes = one 128-byte line), LDS planes [s][row][2] (0) or [k][row] (1), padded
//   ring   LDS-DMA ring of R stages (K + Q per stage), lane = (16-byte group, row): conflict-free ds_read2_b32 operand fetch
//   ring2  micro-stage ring: the Q stage stays in LDS / registers for NA 64-key chunks, NA accumulators per wave
//   ring3  LDS-DMA ring with coalesced transfers (XOR-swizzled inside the line), ds_read_b128 operand fetch + select
// what-if bits (Args::skip) switch parts of a loop off; results: MEASUREMENTS.md (round 3), profiles/r03_probe_lp_stage_*.txt
// hipcc --offload-arch=gfx950 -O3 -w tools/probe_lp_stage.hip -o tools/_build/probe_lp_stage && tools/_build/probe_lp_stage [suite]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct Args {
  const float* fbank;
  float* out;
  int qframe, nkeys, H, W, C, radius;
  int nsub;   // ring2: the 64-key chunks of a window dealt to nsub workgroups
  int skip;   // what-if bits: 1 no global loads, 2 no LDS stores, 4 no operand reads / MFMAs, 8 no barriers, 16 every stage loads the channels of stage 0,
              // 32 MFMA operands from registers (no LDS reads), 64 every workgroup runs 20 key blocks (no imbalance between tiles)
};

__device__ __forceinline__ f32x4 ldf4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int V>
__global__ __launch_bounds__(256) void stage_kernel(Args a) {
  constexpr int BQ = 64, BKEY = 64, BC = 32;
  __shared__ __attribute__((aligned(16))) float sK[BC / 2][BKEY][2];
  __shared__ __attribute__((aligned(16))) float sQ[BC / 2][BQ][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nst = C / BC;
  const int kh = wave & 1, qh = wave >> 1;
  const int li = lane & 31, lk = lane >> 5;
  const int lrow = t & 63, lq = t >> 6;
  const int lqy = min(H - 1, qy0 + (lrow >> 3)), lqx = min(W - 1, qx0 + (lrow & 7));
  const float* qsrc = a.fbank + ((size_t)a.qframe * HW + (size_t)(lqy * W + lqx)) * C;
  const int f = blockIdx.y;
  const int r = a.radius;
  const int wy0 = max(0, qy0 - (r - 1)), wy1 = min(H - 1, qy0 + 7 + (r - 1));
  const int wx0 = max(0, qx0 - (r - 1)), wx1 = min(W - 1, qx0 + 7 + (r - 1));
  const int ww = wx1 - wx0 + 1, nwin = (wy1 - wy0 + 1) * ww;
  const int nkb = (a.skip & 64) ? 20 : (nwin + BKEY - 1) / BKEY;
  float check = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const int kk = min(nwin - 1, kb * BKEY + lrow);
    const int ky = wy0 + kk / ww, kx = wx0 + kk % ww;
    const float* ksrc = a.fbank + ((size_t)f * HW + (size_t)(ky * W + kx)) * C;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 rk[2], rq[2];
    auto load = [&](int st, f32x4 (&k)[2], f32x4 (&q)[2]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = st * BC + (lq + 4 * i) * 4;
        if (V >= 2) {
          k[i] = ldf4(ksrc + c);
          q[i] = ldf4(qsrc + c);
        } else {
          k[i] = (kb * BKEY + lrow < nwin && c < C) ? ldf4(ksrc + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
          q[i] = (qy0 + (lrow >> 3) < H && qx0 + (lrow & 7) < W && c < C) ? ldf4(qsrc + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    auto store = [&](const f32x4 (&k)[2], const f32x4 (&q)[2]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int g = lq + 4 * i;
        *reinterpret_cast<f32x2*>(&sK[2 * g][lrow][0]) = (f32x2){k[i][0], k[i][1]};
        *reinterpret_cast<f32x2*>(&sK[2 * g + 1][lrow][0]) = (f32x2){k[i][2], k[i][3]};
        *reinterpret_cast<f32x2*>(&sQ[2 * g][lrow][0]) = (f32x2){q[i][0], q[i][1]};
        *reinterpret_cast<f32x2*>(&sQ[2 * g + 1][lrow][0]) = (f32x2){q[i][2], q[i][3]};
      }
    };
    auto mfmas = [&]() {
      if (a.skip & 32) {
        const float x = rk[0][0], y = rq[0][0];
#pragma unroll
        for (int s = 0; s < BC / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
      } else if (V == 0) {
#pragma unroll
        for (int s = 0; s < BC / 2; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[s][kh * 32 + li][lk], sQ[s][qh * 32 + li][lk], acc, 0, 0, 0);
      } else {
        float ka[2][2], qa[2][2];
        ka[0][0] = sK[0][kh * 32 + li][lk]; ka[0][1] = sK[1][kh * 32 + li][lk];
        qa[0][0] = sQ[0][qh * 32 + li][lk]; qa[0][1] = sQ[1][qh * 32 + li][lk];
#pragma unroll
        for (int p = 0; p < BC / 4; ++p) {
          const int cur = p & 1, nxt = cur ^ 1;
          if (p + 1 < BC / 4) {
            ka[nxt][0] = sK[2 * p + 2][kh * 32 + li][lk]; ka[nxt][1] = sK[2 * p + 3][kh * 32 + li][lk];
            qa[nxt][0] = sQ[2 * p + 2][qh * 32 + li][lk]; qa[nxt][1] = sQ[2 * p + 3][qh * 32 + li][lk];
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[cur][0], qa[cur][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[cur][1], qa[cur][1], acc, 0, 0, 0);
        }
        // pin the issue order: operand reads of pair p+1 BEFORE the MFMAs of pair p (0x100 = DS read, 0x008 = MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int p = 0; p < BC / 4 - 2; ++p) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    };
    if (V == 2 && a.skip) {
      const int sk = a.skip;
      load(0, rk, rq);
      for (int st = 0; st < nst; ++st) {
        if (!(sk & 2)) store(rk, rq);
        if (!(sk & 8)) __syncthreads();
        if (st + 1 < nst && !(sk & 1)) load((sk & 16) ? 0 : st + 1, rk, rq);
        if (!(sk & 4)) mfmas();
        if (!(sk & 8)) __syncthreads();
      }
    } else if (V <= 2) {
      load(0, rk, rq);
      for (int st = 0; st < nst; ++st) {
        store(rk, rq);
        __syncthreads();
        if (st + 1 < nst) load(st + 1, rk, rq);
        mfmas();
        __syncthreads();
      }
    } else {
      f32x4 rk2[2], rq2[2];
      load(0, rk, rq);
      load(1, rk2, rq2);
      for (int st = 0; st < nst; st += 2) {
        store(rk, rq);
        __syncthreads();
        load(min(st + 2, nst - 1), rk, rq);
        mfmas();
        __syncthreads();
        store(rk2, rq2);
        __syncthreads();
        load(min(st + 3, nst - 1), rk2, rq2);
        mfmas();
        __syncthreads();
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) check += acc[i];
    check += rk[0][0] + rk[1][1] + rq[0][2] + rq[1][3];
  }
  a.out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = check;
}


// ---- V5..: LDS-DMA ring (no register staging, no ds_write, one barrier per stage) -----------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return (u32x4){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
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
template <int N>
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS block j of an operand (8 rows x 128 B, written by ONE DMA instruction as [16-byte group g][row % 8]): the four blocks a wave's
// 32 operand rows live in start at dword offsets = 0, 2, 32, 34 (mod 64), which makes the 64-lane operand read conflict-free
__device__ __forceinline__ constexpr int blk_dw(int j) { return (j >> 2) * 1088 + ((j & 3) == 0 ? 0 : (j & 3) == 1 ? 258 : (j & 3) == 2 ? 544 : 802); }

template <int R, bool GMINOR>
__global__ __launch_bounds__(256) void ring_kernel(Args a) {
  constexpr int BKEY = 64, BC = 32;
  constexpr int OPER_DW = 2176, STAGE_DW = 2 * OPER_DW;
  __shared__ __attribute__((aligned(16))) float ring[R * STAGE_DW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nst = C / BC;
  const int kh = wave & 1, qh = wave >> 1;
  const int li = lane & 31, lk = lane >> 5;
  const int f = blockIdx.y;
  const int r = a.radius;
  const int wy0 = max(0, qy0 - (r - 1)), wy1 = min(H - 1, qy0 + 7 + (r - 1));
  const int wx0 = max(0, qx0 - (r - 1)), wx1 = min(W - 1, qx0 + 7 + (r - 1));
  const int ww = wx1 - wx0 + 1, nwin = (wy1 - wy0 + 1) * ww;
  const int nkb = (a.skip & 64) ? 20 : (nwin + BKEY - 1) / BKEY;
  const int total = nkb * nst;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned ring_addr = (unsigned)(size_t)(lds_ptr)ring;
  // DMA roles: wave w issues blocks {2w, 2w+1} of K and of Q; lane -> (16-byte group g, row r8) of the block
  const int dg = GMINOR ? (lane & 7) : (lane >> 3), dr = GMINOR ? (lane >> 3) : (lane & 7);
  const u32x4 krs = make_rsrc(a.fbank + (size_t)f * HW * C, (unsigned)((size_t)HW * C * 4));
  const u32x4 qrs = make_rsrc(a.fbank + (size_t)a.qframe * HW * C, (unsigned)((size_t)HW * C * 4));
  unsigned qvo[2], kvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (2 * wave + i) * 8 + dr;
    const int y = min(H - 1, qy0 + (row >> 3)), x = min(W - 1, qx0 + (row & 7));
    qvo[i] = (unsigned)(((size_t)(y * W + x) * C + dg * 4) * 4);
  }
  auto kaddr = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = min(nwin - 1, kb * BKEY + (2 * wave + i) * 8 + dr);
      const int ky = wy0 + kk / ww, kx = wx0 + kk % ww;
      kvo[i] = (unsigned)(((size_t)(ky * W + kx) * C + dg * 4) * 4);
    }
  };
  int ikb = 0, ist = 0;        // (key block, stage) of the next stage to issue
  kaddr(0);
  auto issue = [&](int n) {
    const unsigned base = ring_addr + (unsigned)((n % R) * STAGE_DW * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = 2 * wave + i;
      const unsigned so = (a.skip & 16) ? 0u : (unsigned)ist * 128u;
      dma16(krs, __builtin_amdgcn_readfirstlane(base + blk_dw(j) * 4), (a.skip & 128) ? qvo[i] : kvo[i], so);
      dma16(qrs, __builtin_amdgcn_readfirstlane(base + (OPER_DW + blk_dw(j)) * 4), qvo[i], so);
    }
    if (++ist == nst) { ist = 0; ++ikb; kaddr(min(ikb, nkb - 1)); }
  };
  // operand fragments: row kh*32 + li of K, row qh*32 + li of Q; dword (li%8)*4 + lk inside the 128-byte group line
  const int kblk = blk_dw(4 * kh + (li >> 3)) + (li & 7) * 4 + lk;
  const int qblk = OPER_DW + blk_dw(4 * qh + (li >> 3)) + (li & 7) * 4 + lk;
  float check = 0.f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int n = 0; n < R - 1; ++n) issue(n);
  int st = 0;
  for (int n = 0; n < total; ++n) {
    if (n + R - 1 <= total) {
      if (R == 2) dma_wait<0>(); else if (R == 3) dma_wait<4>(); else dma_wait<8>();
    } else {
      dma_wait<0>();
    }
    __syncthreads();
    if (n + R - 1 < total && !(a.skip & 1)) issue(n + R - 1);
    if (a.skip & 4) continue;
    const float* sk = ring + (n % R) * STAGE_DW + kblk;
    const float* sq = ring + (n % R) * STAGE_DW + qblk;
    float ka[2][2], qa[2][2];
    ka[0][0] = sk[0]; ka[0][1] = sk[2]; qa[0][0] = sq[0]; qa[0][1] = sq[2];
#pragma unroll
    for (int p = 0; p < BC / 4; ++p) {
      const int cur = p & 1, nxt = cur ^ 1;
      if (p + 1 < BC / 4) {
        ka[nxt][0] = sk[(p + 1) * 32]; ka[nxt][1] = sk[(p + 1) * 32 + 2];
        qa[nxt][0] = sq[(p + 1) * 32]; qa[nxt][1] = sq[(p + 1) * 32 + 2];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[cur][0], qa[cur][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[cur][1], qa[cur][1], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int p = 0; p < BC / 4 - 2; ++p) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    if (++st == nst) {
      st = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) { check += acc[i]; acc[i] = 0.f; }
    }
  }
  a.out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = check;
}

template <int R, bool GMINOR>
static double run_ring(const Args& a, int tiles, std::vector<float>& host) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ring_kernel<R, GMINOR>), dim3(tiles, a.nkeys), dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((ring_kernel<R, GMINOR>), dim3(tiles, a.nkeys), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  hipMemcpy(host.data(), a.out, host.size() * 4, hipMemcpyDeviceToHost);
  double s = 0.0;
  for (float v : host) s += (double)v;
  printf("ring R=%d %s nkeys %2d skip %2d: %.3f ms   checksum %.9e\n", R, GMINOR ? "lane=(row,g)" : "lane=(g,row)", a.nkeys, a.skip, best, s);
  return best;
}

// ---- micro-stage ring: a stage of Q (32 channels x 64 queries) stays in LDS (double-buffered) and in the waves' registers for NA
// 64-key chunks; only the K chunks stream through a 3-slot ring.  One barrier and 16 MFMAs per wave per micro-stage, NA accumulators.
template <int NA>
__global__ __launch_bounds__(256) void ring2_kernel(Args a) {
  constexpr int BKEY = 64, BC = 32, R = 3;
  constexpr int OPER_DW = 2176;
  __shared__ __attribute__((aligned(16))) float ringK[R * OPER_DW];
  __shared__ __attribute__((aligned(16))) float bufQ[2 * OPER_DW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nst = C / BC;
  const int kh = wave & 1, qh = wave >> 1;
  const int li = lane & 31, lk = lane >> 5;
  const int nsub = a.nsub;
  const int f = blockIdx.y / nsub, sub = blockIdx.y % nsub;
  const int r = a.radius;
  const int wy0 = max(0, qy0 - (r - 1)), wy1 = min(H - 1, qy0 + 7 + (r - 1));
  const int wx0 = max(0, qx0 - (r - 1)), wx1 = min(W - 1, qx0 + 7 + (r - 1));
  const int ww = wx1 - wx0 + 1, nwin = (wy1 - wy0 + 1) * ww;
  const int nkb_all = (a.skip & 64) ? 20 : (nwin + BKEY - 1) / BKEY;
  const int cpb = (nkb_all + nsub - 1) / nsub;
  const int kb0 = sub * cpb, nkb = max(0, min(nkb_all, kb0 + cpb) - kb0);        // this workgroup's 64-key chunks
  const int nsb = (nkb + NA - 1) / NA;                                            // super blocks of NA chunks (the last one shorter)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned k_addr = (unsigned)(size_t)(lds_ptr)ringK, q_addr = (unsigned)(size_t)(lds_ptr)bufQ;
  const int dg = lane >> 3, dr = lane & 7;
  const u32x4 krs = make_rsrc(a.fbank + (size_t)f * HW * C, (unsigned)((size_t)HW * C * 4));
  const u32x4 qrs = make_rsrc(a.fbank + (size_t)a.qframe * HW * C, (unsigned)((size_t)HW * C * 4));
  unsigned qvo[2], kvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (2 * wave + i) * 8 + dr;
    const int y = min(H - 1, qy0 + (row >> 3)), x = min(W - 1, qx0 + (row & 7));
    qvo[i] = (unsigned)(((size_t)(y * W + x) * C + dg * 4) * 4);
  }
  auto kaddr = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = min(nwin - 1, (kb0 + kb) * BKEY + (2 * wave + i) * 8 + dr);
      const int ky = wy0 + kk / ww, kx = wx0 + kk % ww;
      kvo[i] = (unsigned)(((size_t)(ky * W + kx) * C + dg * 4) * 4);
    }
  };
  // the issue cursor walks micro-stages in execution order: super block isb, stage ist, chunk ic (na_i chunks in that super block)
  int isb = 0, ist = 0, ic = 0, islot = 0;
  auto na_of = [&](int sb) { return min(NA, nkb - sb * NA); };
  auto issue = [&]() {
    kaddr(isb * NA + ic);
    const unsigned kb_ = k_addr + (unsigned)(islot * OPER_DW * 4);
    const unsigned qb_ = q_addr + (unsigned)((ist & 1) * OPER_DW * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(krs, __builtin_amdgcn_readfirstlane(kb_ + blk_dw(2 * wave + i) * 4), kvo[i], (unsigned)ist * 128u);
    if (ic == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16(qrs, __builtin_amdgcn_readfirstlane(qb_ + blk_dw(2 * wave + i) * 4), qvo[i], (unsigned)ist * 128u);
    }
    islot = islot == R - 1 ? 0 : islot + 1;
    if (++ic == na_of(isb)) { ic = 0; if (++ist == nst) { ist = 0; ++isb; } }
  };
  const int kfrag = blk_dw(4 * kh + (li >> 3)) + (li & 7) * 4 + lk;
  const int qfrag = blk_dw(4 * qh + (li >> 3)) + (li & 7) * 4 + lk;
  float check = 0.f;
  f32x16 acc[NA];
  float qa[BC / 2];
  int total = 0;
  for (int sb = 0; sb < nsb; ++sb) total += na_of(sb) * nst;
  int issued = 0;
  for (; issued < R - 1 && issued < total; ++issued) issue();
  int slot = 0, m = 0;
  for (int sb = 0; sb < nsb; ++sb) {
    const int na = na_of(sb);
#pragma unroll
    for (int c = 0; c < NA; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int st = 0; st < nst; ++st) {
#pragma unroll
      for (int c = 0; c < NA; ++c) {
        if (c < na) {
          // the group issued after this micro-stage's own: 2 K pieces, + 2 Q pieces when it opens a stage
          const bool next_opens = (c + 1 == na);
          if (m + 1 >= total) dma_wait<0>(); else if (next_opens) dma_wait<4>(); else dma_wait<2>();
          if (!(a.skip & 8)) __syncthreads();
          if (issued < total && !(a.skip & 1)) { issue(); ++issued; }
          ++m;
          if (!(a.skip & 4)) {
            if (c == 0) {
              const float* sq = bufQ + (st & 1) * OPER_DW + qfrag;
#pragma unroll
              for (int p = 0; p < BC / 4; ++p) { qa[2 * p] = sq[p * 32]; qa[2 * p + 1] = sq[p * 32 + 2]; }
            }
            const float* sk = ringK + slot * OPER_DW + kfrag;
#pragma unroll
            for (int p = 0; p < BC / 4; ++p) {
              const float k0 = (a.skip & 32) ? qa[2 * p + 1] : sk[p * 32], k1 = (a.skip & 32) ? qa[2 * p] : sk[p * 32 + 2];
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0, qa[2 * p], acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1, qa[2 * p + 1], acc[c], 0, 0, 0);
            }
          }
          slot = slot == R - 1 ? 0 : slot + 1;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NA; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) check += acc[c][i];
  }
  a.out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = check;
}

template <int NA>
static double run_ring2(const Args& a, int tiles, float* out, size_t nout) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(out, 0, nout * 4);
  hipLaunchKernelGGL((ring2_kernel<NA>), dim3(tiles, a.nkeys * a.nsub), dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((ring2_kernel<NA>), dim3(tiles, a.nkeys * a.nsub), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  std::vector<float> host(nout);
  hipMemcpy(host.data(), out, nout * 4, hipMemcpyDeviceToHost);
  double s = 0.0;
  for (float v : host) s += (double)v;
  printf("ring2 NA=%d nsub %d nkeys %2d skip %2d: %.3f ms   checksum %.9e\n", NA, a.nsub, a.nkeys, a.skip, best, s);
  return best;
}

// ---- ring with COALESCED transfers: lane (row p, j) of a DMA instruction fetches the 16-byte group j ^ p of row p (8 consecutive
// lanes = one 128-byte line, XOR-swizzled inside the line); LDS block = 8 rows x 128 B, block stride 1152 B (bases alternate 0 / 128
// mod 256): the operand fetch is ONE ds_read_b128 per two MFMA steps (the lane uses 2 of the 4 channels), conflict-free in all four
// 16-lane groups (brute-forced: tools/README).
template <int R>
__global__ __launch_bounds__(256) void ring3_kernel(Args a) {
  constexpr int BKEY = 64, BC = 32;
  constexpr int BLK_B = 1152, OPER_B = 8 * BLK_B, STAGE_B = 2 * OPER_B;
  __shared__ __attribute__((aligned(16))) char ring[R * STAGE_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nst = C / BC;
  const int kh = wave & 1, qh = wave >> 1;
  const int li = lane & 31, lk = lane >> 5;
  const int nsub = a.nsub;
  const int f = blockIdx.y / nsub, sub = blockIdx.y % nsub;
  const int r = a.radius;
  const int wy0 = max(0, qy0 - (r - 1)), wy1 = min(H - 1, qy0 + 7 + (r - 1));
  const int wx0 = max(0, qx0 - (r - 1)), wx1 = min(W - 1, qx0 + 7 + (r - 1));
  const int ww = wx1 - wx0 + 1, nwin = (wy1 - wy0 + 1) * ww;
  const int nkb_all = (a.skip & 64) ? 20 : (nwin + BKEY - 1) / BKEY;
  const int cpb = (nkb_all + nsub - 1) / nsub;
  const int kb0 = sub * cpb, nkb = max(0, min(nkb_all, kb0 + cpb) - kb0);
  const int total = nkb * nst;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned ring_addr = (unsigned)(size_t)(lds_ptr)ring;
  const int dp = lane >> 3, dg = (lane & 7) ^ dp;       // this lane's row of the block and the (swizzled) group it fetches
  const u32x4 krs = make_rsrc(a.fbank + (size_t)f * HW * C, (unsigned)((size_t)HW * C * 4));
  const u32x4 qrs = make_rsrc(a.fbank + (size_t)a.qframe * HW * C, (unsigned)((size_t)HW * C * 4));
  unsigned qvo[2], kvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (2 * wave + i) * 8 + dp;
    const int y = min(H - 1, qy0 + (row >> 3)), x = min(W - 1, qx0 + (row & 7));
    qvo[i] = (unsigned)(((size_t)(y * W + x) * C + dg * 4) * 4);
  }
  int ikb = 0, ist = 0;
  auto kaddr = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = min(nwin - 1, (kb0 + kb) * BKEY + (2 * wave + i) * 8 + dp);
      const int ky = wy0 + kk / ww, kx = wx0 + kk % ww;
      kvo[i] = (unsigned)(((size_t)(ky * W + kx) * C + dg * 4) * 4);
    }
  };
  kaddr(0);
  auto issue = [&](int n) {
    const unsigned base = ring_addr + (unsigned)((n % R) * STAGE_B);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = 2 * wave + i;
      dma16(krs, __builtin_amdgcn_readfirstlane(base + j * BLK_B), kvo[i], (unsigned)ist * 128u);
      dma16(qrs, __builtin_amdgcn_readfirstlane(base + OPER_B + j * BLK_B), qvo[i], (unsigned)ist * 128u);
    }
    if (++ist == nst) { ist = 0; ++ikb; kaddr(min(ikb, nkb - 1)); }
  };
  // operand row kh*32 + li (K) / qh*32 + li (Q): block (row / 8), position p = li % 8; group g sits at byte ((g ^ p) * 16) of the row
  const int p8 = li & 7;
  const int krow = (4 * kh + (li >> 3)) * BLK_B + p8 * 128;
  const int qrow = OPER_B + (4 * qh + (li >> 3)) * BLK_B + p8 * 128;
  float check = 0.f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int n = 0; n < R - 1 && n < total; ++n) issue(n);
  int st = 0;
  for (int n = 0; n < total; ++n) {
    if (n + R - 1 <= total && R > 1) {
      if (R == 2) dma_wait<0>(); else if (R == 3) dma_wait<4>(); else dma_wait<8>();
    } else {
      dma_wait<0>();
    }
    if (!(a.skip & 8)) __syncthreads();
    if (n + R - 1 < total && !(a.skip & 1)) issue(n + R - 1);
    if (!(a.skip & 4)) {
      const char* sb = ring + (n % R) * STAGE_B;
      f32x4 kv[2], qv[2];
      kv[0] = *reinterpret_cast<const f32x4*>(sb + krow + ((0 ^ p8) * 16));
      qv[0] = *reinterpret_cast<const f32x4*>(sb + qrow + ((0 ^ p8) * 16));
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < 8) {
          kv[nxt] = *reinterpret_cast<const f32x4*>(sb + krow + (((g + 1) ^ p8) * 16));
          qv[nxt] = *reinterpret_cast<const f32x4*>(sb + qrow + (((g + 1) ^ p8) * 16));
        }
        // (the scalars pass through an empty asm: otherwise the selects become dynamic vector indexing, 3 v_cndmask each)
        float a0 = kv[cur][0], a1 = kv[cur][1], a2 = kv[cur][2], a3 = kv[cur][3];
        float b0 = qv[cur][0], b1 = qv[cur][1], b2 = qv[cur][2], b3 = qv[cur][3];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        const float k0 = lk ? a1 : a0, k1 = lk ? a3 : a2;
        const float q0 = lk ? b1 : b0, q1 = lk ? b3 : b2;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k0, q0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k1, q1, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
    if (++st == nst) {
      st = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) { check += acc[i]; acc[i] = 0.f; }
    }
  }
  a.out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = check;
}

template <int R>
static double run_ring3(const Args& a, int tiles, float* out, size_t nout) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(out, 0, nout * 4);
  hipLaunchKernelGGL((ring3_kernel<R>), dim3(tiles, a.nkeys * a.nsub), dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((ring3_kernel<R>), dim3(tiles, a.nkeys * a.nsub), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  std::vector<float> host(nout);
  hipMemcpy(host.data(), out, nout * 4, hipMemcpyDeviceToHost);
  double s = 0.0;
  for (float v : host) s += (double)v;
  printf("ring3 (coalesced, b128 reads) R=%d nsub %d nkeys %2d skip %2d: %.3f ms   checksum %.9e\n", R, a.nsub, a.nkeys, a.skip, best, s);
  return best;
}

// ---- register-staged loop with COALESCED loaders: thread (row = t / 8 (+32), group = t % 8) - 8 consecutive lanes read one 128-byte
// line.  LAYOUT 0: planes [s][row][2] with one row of padding per plane (conflict-free ds_write_b64, operand reads 2-way as in V0);
// LAYOUT 1: planes [s][k][row] with one dword of padding (4 ds_write_b32 per float4, conflict-free stores AND operand reads).
// 36 KB of LDS ballast keeps it at the product kernel's 3 workgroups per CU.
template <int LAYOUT>
__global__ __launch_bounds__(256) void coal_kernel(Args a) {
  constexpr int BKEY = 64, BC = 32;
  constexpr int PL = LAYOUT == 0 ? (BKEY + 1) * 2 : 2 * (BKEY + 1);     // dwords per k-pair plane
  __shared__ __attribute__((aligned(16))) float sK[(BC / 2) * PL];
  __shared__ __attribute__((aligned(16))) float sQ[(BC / 2) * PL];
  __shared__ float ballast[9000];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nst = C / BC;
  const int kh = wave & 1, qh = wave >> 1;
  const int li = lane & 31, lk = lane >> 5;
  const int lrow = t >> 3, lg = t & 7;                  // loader: rows lrow and lrow + 32, 16-byte group lg
  const float* qsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = lrow + 32 * i;
    const int y = min(H - 1, qy0 + (row >> 3)), x = min(W - 1, qx0 + (row & 7));
    qsrc[i] = a.fbank + ((size_t)a.qframe * HW + (size_t)(y * W + x)) * C + lg * 4;
  }
  const int f = blockIdx.y;
  const int r = a.radius;
  const int wy0 = max(0, qy0 - (r - 1)), wy1 = min(H - 1, qy0 + 7 + (r - 1));
  const int wx0 = max(0, qx0 - (r - 1)), wx1 = min(W - 1, qx0 + 7 + (r - 1));
  const int ww = wx1 - wx0 + 1, nwin = (wy1 - wy0 + 1) * ww;
  const int nkb = (a.skip & 64) ? 20 : (nwin + BKEY - 1) / BKEY;
  if (a.skip == 12345) ballast[t] = 1.f;
  float check = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const float* ksrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = min(nwin - 1, kb * BKEY + lrow + 32 * i);
      ksrc[i] = a.fbank + ((size_t)f * HW + (size_t)((wy0 + kk / ww) * W + wx0 + kk % ww)) * C + lg * 4;
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 rk[2], rq[2];
    auto load = [&](int st) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { rk[i] = ldf4(ksrc[i] + st * BC); rq[i] = ldf4(qsrc[i] + st * BC); }
    };
    auto store = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = lrow + 32 * i;
        if (LAYOUT == 0) {
          *reinterpret_cast<f32x2*>(&sK[(2 * lg) * PL + row * 2]) = (f32x2){rk[i][0], rk[i][1]};
          *reinterpret_cast<f32x2*>(&sK[(2 * lg + 1) * PL + row * 2]) = (f32x2){rk[i][2], rk[i][3]};
          *reinterpret_cast<f32x2*>(&sQ[(2 * lg) * PL + row * 2]) = (f32x2){rq[i][0], rq[i][1]};
          *reinterpret_cast<f32x2*>(&sQ[(2 * lg + 1) * PL + row * 2]) = (f32x2){rq[i][2], rq[i][3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sK[(2 * lg + (e >> 1)) * PL + (e & 1) * (BKEY + 1) + row] = rk[i][e];
            sQ[(2 * lg + (e >> 1)) * PL + (e & 1) * (BKEY + 1) + row] = rq[i][e];
          }
        }
      }
    };
    const int kfr = LAYOUT == 0 ? (kh * 32 + li) * 2 + lk : lk * (BKEY + 1) + kh * 32 + li;
    const int qfr = LAYOUT == 0 ? (qh * 32 + li) * 2 + lk : lk * (BKEY + 1) + qh * 32 + li;
    load(0);
    for (int st = 0; st < nst; ++st) {
      if (!(a.skip & 2)) store();
      __syncthreads();
      if (st + 1 < nst && !(a.skip & 1)) load(st + 1);
      if (!(a.skip & 4)) {
        float ka[2][2], qa[2][2];
        ka[0][0] = sK[kfr]; ka[0][1] = sK[PL + kfr]; qa[0][0] = sQ[qfr]; qa[0][1] = sQ[PL + qfr];
#pragma unroll
        for (int p = 0; p < BC / 4; ++p) {
          const int cur = p & 1, nxt = cur ^ 1;
          if (p + 1 < BC / 4) {
            ka[nxt][0] = sK[(2 * p + 2) * PL + kfr]; ka[nxt][1] = sK[(2 * p + 3) * PL + kfr];
            qa[nxt][0] = sQ[(2 * p + 2) * PL + qfr]; qa[nxt][1] = sQ[(2 * p + 3) * PL + qfr];
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[cur][0], qa[cur][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[cur][1], qa[cur][1], acc, 0, 0, 0);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) check += acc[i];
  }
  a.out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = check;
}

template <int LAYOUT>
static double run_coal(const Args& a, int tiles, std::vector<float>& host) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((coal_kernel<LAYOUT>), dim3(tiles, a.nkeys), dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((coal_kernel<LAYOUT>), dim3(tiles, a.nkeys), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  hipMemcpy(host.data(), a.out, host.size() * 4, hipMemcpyDeviceToHost);
  double s = 0.0;
  for (float v : host) s += (double)v;
  printf("coalesced loaders, layout %d, nkeys %2d skip %2d: %.3f ms   checksum %.9e\n", LAYOUT, a.nkeys, a.skip, best, s);
  return best;
}

// ballast: dynamic LDS bytes that only lower the occupancy (36000: 3 workgroups per CU like the product kernel; 0: 8 per CU)
template <int V>
static double run(const Args& a, int tiles, std::vector<float>& host, unsigned ballast = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(stage_kernel<V>, dim3(tiles, a.nkeys), dim3(256), ballast, 0, a);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(stage_kernel<V>, dim3(tiles, a.nkeys), dim3(256), ballast, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  hipMemcpy(host.data(), a.out, host.size() * 4, hipMemcpyDeviceToHost);
  double s = 0.0;
  for (float v : host) s += (double)v;
  printf("V%d nkeys %2d skip %2d%s: %.3f ms   checksum %.9e\n", V, a.nkeys, a.skip, ballast ? " (3 workgroups per CU)" : "", best, s);
  return best;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int H = 60, W = 107, C = 1024, T = 22;
  const size_t n = (size_t)T * H * W * C;
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) % 2001 - 1000) * 3.0e-5f; }
  float* bank; hipMalloc(&bank, n * 4); hipMemcpy(bank, h.data(), n * 4, hipMemcpyHostToDevice);
  const int tiles = ((H + 7) / 8) * ((W + 7) / 8);
  std::vector<float> host((size_t)tiles * 21 * 256);
  float* out; hipMalloc(&out, host.size() * 4);
  // suites: "summary" (default) every structure once, "v2" / "coal" / "ring" / "ring2" / "ring3": the what-if runs of one structure
  const std::string suite = argc > 1 ? argv[1] : "summary";
  Args a{bank, out, T - 1, 21, H, W, C, 18, 1, 0};
  float* big; hipMalloc(&big, host.size() * 8 * 4);       // per-thread checksums of the sub-split runs (up to 8 workgroups per key frame)
  const size_t nbig = host.size() * 8;
  if (suite == "summary") {
    run<0>(a, tiles, host); run<1>(a, tiles, host); run<2>(a, tiles, host); run<3>(a, tiles, host);
    run_coal<0>(a, tiles, host); run_coal<1>(a, tiles, host);
    run_ring<2, false>(a, tiles, host); run_ring<3, false>(a, tiles, host); run_ring<4, false>(a, tiles, host);
    a.out = big;
    run_ring3<2>(a, tiles, big, nbig); run_ring2<2>(a, tiles, big, nbig); run_ring2<4>(a, tiles, big, nbig);
    a.nsub = 4;
    run_ring3<2>(a, tiles, big, nbig); run_ring2<2>(a, tiles, big, nbig); run_ring2<4>(a, tiles, big, nbig);
  } else if (suite == "occ") {
    for (unsigned b : {0u, 36000u, 60000u}) { run<0>(a, tiles, host, b); run<2>(a, tiles, host, b); }
    run_coal<0>(a, tiles, host); run_coal<1>(a, tiles, host);
  } else if (suite == "v2") {
    for (int sk : {0, 1, 2, 4, 8, 16, 3, 5, 6, 7, 11, 11 | 32, 64, 64 | 3, 64 | 11 | 32}) { a.skip = sk; run<2>(a, tiles, host); }
  } else if (suite == "coal") {
    for (int sk : {0, 1 | 2, 2 | 4, 64}) { a.skip = sk; run_coal<0>(a, tiles, host); run_coal<1>(a, tiles, host); }
  } else if (suite == "ring") {
    for (int sk : {0, 1, 4, 16, 128, 128 | 16, 4 | 128 | 16, 64, 65, 68}) { a.skip = sk; run_ring<3, false>(a, tiles, host); run_ring<3, true>(a, tiles, host); }
  } else if (suite == "ring2" || suite == "ring3") {
    a.out = big;
    for (int ns : {1, 4}) {
      a.nsub = ns;
      for (int sk : {0, 1, 4, 64, 64 | 1, 64 | 1 | 8 | 32}) {
        a.skip = sk;
        if (suite == "ring2") { run_ring2<2>(a, tiles, big, nbig); run_ring2<4>(a, tiles, big, nbig); }
        else { run_ring3<2>(a, tiles, big, nbig); run_ring3<3>(a, tiles, big, nbig); }
      }
    }
  }
  return 0;
}

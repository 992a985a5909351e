// Two-pass EXACT label propagation for gfx950: the top-10 of masked_attention_efficient (mmaction/models/common/local_attention.py:
// 277-335) with the bits of the dense fp32 kernel (exact_f32.hip / oracle/exact_oracle.c), at a fraction of its fp32 work.
//
//   pass 1  (lp2_score_kernel)   every in-window candidate is scored on the bf16 MATRIX path (v_mfma_f32_32x32x16_bf16, 16x the fp32
//           rate) from a SPLIT copy of the bank: x = hi + lo + e, hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-16 |x|, and
//               s~ = hi_q.hi_k + hi_q.lo_k + lo_q.hi_k          (three MFMAs per 16 channels; lo.lo <= 2^-16 is dropped)
//           For unit vectors |s~ - s| <= EPS = 3 * 2^-16 + 4 * C * 2^-24 (representation: Cauchy-Schwarz on the dropped terms;
//           accumulation: C roundings of the exact chain + the matrix unit's internal sums, both bounded by their worst case).
//           Every candidate with s~ >= (running lower bound of the query's 10th-best s~) - 2 EPS is appended to the query's list.
//           The bound starts from a SEED (lp2_seed_kernel: the query's own position in every key frame - in a video the best
//           matches - scored exactly, 10th best minus EPS), follows the query's true running 10th best inside a workgroup and is
//           shared between the workgroups that split a query's key frames through a device-scope atomic max: the lists hold a
//           few dozen entries, not the thousands a cold start lists before its threshold has risen.
//   pass 2  (lp2_refine_kernel)  per query: t = the 10th largest s~ of its list (= of ALL its candidates: the ten best are always
//           listed), survivors = {s~ >= t - 2 EPS}.  A candidate outside the survivors has ten candidates whose s~ exceeds its own
//           by more than 2 EPS, hence whose EXACT score is strictly larger: it cannot be in the exact top 10 under any tie rule.
//           The survivors (a few dozen) are rescored with the defining arithmetic - one ascending chain acc = fma(k_c, q_c, acc)
//           per pair, bitwise what v_mfma_f32_32x32x2_f32 computes in the dense kernel - and go through the same total order
//           (score desc, candidate id asc), softmax (vexp) and value sum as labelprop_f32_merge_kernel: identical bits, by
//           construction, whatever pass 1's rounding did.
//   Anything the lists cannot hold (capacity overflow), a bank that is not unit-norm (test_cfg.with_norm=False) or a channel count
//   the register-resident query tile does not cover falls back to the dense kernel; the overflow case WITHOUT a host round trip
//   (the dense launches read a device flag and exit at once when it is clear).
//
// Pass-1 work shape (what round 3's probes asked for, MEASUREMENTS.md "What bounds the fp32 evaluation kernels"): the 8x8 query
// tile lives in REGISTERS for the whole workgroup - wave w holds channels [w C/4, (w+1) C/4) of all 64 queries as MFMA B
// operands (2 x C/64 x 8 VGPRs) - so only KEY rows stream: every wave pulls ITS channel quarter of a 64-key block through its own
// three-stage LDS ring by LDS-DMA (buffer_load ... lds, whole 128-byte lines: the split bank interleaves hi / lo per 16 channels),
// no barrier and no cross-wave traffic inside the channel loop; the four partial 64x64 score tiles meet in LDS once per key block.
#include "vfs_lpx.h"

#pragma clang fp contract(off)

#ifndef LP2_PF
#define LP2_PF 0      // 1: key fragments requested half a stage ahead into a second register set (round 5: measured level, MEASUREMENTS.md)
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---------------------------------------------------------------------------------------------
// x [P][C] fp32 -> hl [P][C/16][hi k0..7 | hi k8..15 | lo k0..7 | lo k8..15] bf16: one thread per 8 channels
__global__ __launch_bounds__(256) void split_rows_bf16x2_kernel(const float* __restrict__ x, bf16_t* __restrict__ hl, long long P, int C) {
  const int c8n = C >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P * c8n) return;
  const long long row = i / c8n;
  const int c8 = (int)(i - row * c8n);
  const float* src = x + (size_t)row * C + c8 * 8;
  const f32x4 v0 = lpx_ldf4(src), v1 = lpx_ldf4(src + 4);
  float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, lo[8];
  u32x4 hv, lv;
#pragma unroll
  for (int j = 0; j < 8; ++j) lo[j] = v[j] - round_bf(v[j]);      // exact in fp32
  hv = pack8(v);
  lv = pack8(lo);
  bf16_t* dst = hl + (size_t)row * 2 * C + (c8 >> 1) * 32 + (c8 & 1) * 8;
  st16(dst, hv);
  st16(dst + 16, lv);
}
int vfs_split_rows_bf16x2_launch(const float* x, bf16_t* hl, long long P, int C, hipStream_t s) {
  if (C % 16) return vfs_set_error(VFS_ERR_SHAPE, "split_rows_bf16x2: C % 16");
  const long long total = P * (C / 8);
  hipLaunchKernelGGL(split_rows_bf16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, hl, P, C);
  return vfs_check_launch("split_rows_bf16x2");
}

// ---------------------------------------------------------------------------------------------
// monotone float <-> int (for an LDS atomic max over floats of either sign)
__device__ __forceinline__ int lp2_enc(float f) {
  const int b = __builtin_bit_cast(int, f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float lp2_dec(int e) { return __builtin_bit_cast(float, e >= 0 ? e : e ^ 0x7fffffff); }

// sorted insertion of a value into a descending list of 10 (pass 1 only needs thresholds, not ids)
__device__ __forceinline__ void lp2_insert_val(float (&tv)[LPX_TOPK], float s) {
  if (s > tv[LPX_TOPK - 1]) tv[LPX_TOPK - 1] = s;
#pragma unroll
  for (int j = LPX_TOPK - 1; j > 0; --j) {
    const float a = tv[j - 1], b = tv[j];
    tv[j - 1] = b > a ? b : a;
    tv[j] = b > a ? a : b;
  }
}

// the window of one key frame for this query tile, and the centre-out order of its 64-key blocks (the best matches of a video
// sit near the query's own position: visiting those rows first lets the running threshold rise at once, so few candidates
// are listed before it has)
struct Lp2Window {
  int slot, r, wy0, wx0, ww, nwin, nkb;
  float rww;      // 1 / ww
  bool tab;       // keys enumerated by the workgroup's table of a TRIMMED window (below) instead of the rectangle
};
// Trimmed window of a masked key frame: row y of the rectangle keeps only the columns SOME query of the 8 x 8 tile can reach,
// [qx0 - dx(y), qx0 + 7 + dx(y)] with dx(y) = the largest dx with dx^2 + dmin(y)^2 < r^2, dmin(y) = the distance from y to the
// tile's rows.  At radius 18 the corners it drops are 12.5 % of the 42 x 42 rectangle - key rows that were streamed, multiplied and
// masked for every query.  The geometry is the same for every masked key frame of a tile: ONE table per workgroup, key index ->
// (y << 16 | x) in LDS (which also replaces the integer divisions of the rectangle's index arithmetic).
__device__ __forceinline__ int lp2_row_reach(int dmin, int r) {      // dx(y) above; dmin <= r - 1
  const int lim = r * r - 1 - dmin * dmin;      // dx^2 <= lim
  int dx = (int)__builtin_sqrtf((float)lim);
  while ((dx + 1) * (dx + 1) <= lim) ++dx;
  while (dx * dx > lim) --dx;
  return dx;
}
__device__ __forceinline__ Lp2Window lp2_window(const Lp2Args& a, int f, int qy0, int qx0, int tab_nwin = 0) {
  Lp2Window w;
  w.slot = a.kslot[f];
  w.r = f < a.non_mask_len ? 0 : a.radius;
  w.tab = false;
  int wy1 = a.H - 1, wx1 = a.W - 1;
  w.wy0 = 0; w.wx0 = 0;
  if (w.r > 0) {
    w.wy0 = max(0, qy0 - (w.r - 1)); wy1 = min(a.H - 1, qy0 + 7 + (w.r - 1));
    w.wx0 = max(0, qx0 - (w.r - 1)); wx1 = min(a.W - 1, qx0 + 7 + (w.r - 1));
  }
  w.ww = wx1 - w.wx0 + 1;
  w.nwin = (wy1 - w.wy0 + 1) * w.ww;
  if (w.r > 0 && tab_nwin > 0) { w.tab = true; w.nwin = tab_nwin; }
  w.nkb = (w.nwin + 63) >> 6;
  w.rww = 1.0f / (float)w.ww;
  return w;
}
__device__ __forceinline__ int lp2_block_of(int i, int nkb, int stagger = 0) {      // i-th block in centre-out order
  i += stagger;                    // (XCD-aware order: neighbouring tiles run `stagger` blocks apart, so that one FETCHES a key row
  if (i >= nkb) i -= nkb;          //  and the next HITS it instead of both waiting for the same line in flight)
  const int mid = (nkb - 1) >> 1, off = (i + 1) >> 1;
  return (i & 1) ? mid + off : mid - off;
}

// Seed of the running threshold: one wave per query scores the query's OWN pixel in every key frame (plus its four neighbours while
// there are fewer than ten key frames) with an fp32 dot product; gthr[q] = (10th best - EPS) is a valid lower bound of the 10th
// best s~ pass 1 will see (|dot - s~| <= EPS for every candidate, and the 10th best over a subset never exceeds the 10th best
// over all).  -inf when fewer than ten seeds exist.
__global__ __launch_bounds__(256) void lp2_seed_kernel(Lp2Args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int q = blockIdx.x * 4 + wave;
  if (q >= HW) return;
  const int qy = q / W, qx = q - qy * W;
  const float* qrow = a.fbank + ((size_t)a.qframe * HW + q) * C;
  f32x4 qv[4];      // C <= 1024: channels 256 j + 4 lane .. + 3
#pragma unroll
  for (int j = 0; j < 4; ++j) qv[j] = 256 * j < C ? lpx_ldf4(qrow + 256 * j + 4 * lane) : (f32x4){0.f, 0.f, 0.f, 0.f};
  float tv[LPX_TOPK];
#pragma unroll
  for (int i = 0; i < LPX_TOPK; ++i) tv[i] = -INFINITY;
  const int noff = a.nkeys < LPX_TOPK ? 5 : 1;
  for (int f = 0; f < a.nkeys; ++f) {
    const int r = f < a.non_mask_len ? 0 : a.radius;
    const float* frame = a.fbank + (size_t)a.kslot[f] * HW * C;
    for (int o = 0; o < noff; ++o) {
      const int dy = o == 3 ? 1 : (o == 4 ? -1 : 0), dx = o == 1 ? 1 : (o == 2 ? -1 : 0);
      const int ky = qy + dy, kx = qx + dx;
      if (ky < 0 || ky >= H || kx < 0 || kx >= W) continue;
      if (r > 0 && dy * dy + dx * dx >= r * r) continue;
      const float* krow = frame + (size_t)(ky * W + kx) * C;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (256 * j < C) {
          const f32x4 kv = lpx_ldf4(krow + 256 * j + 4 * lane);
          acc = __builtin_fmaf(kv[0], qv[j][0], acc); acc = __builtin_fmaf(kv[1], qv[j][1], acc);
          acc = __builtin_fmaf(kv[2], qv[j][2], acc); acc = __builtin_fmaf(kv[3], qv[j][3], acc);
        }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc = acc + __shfl_xor(acc, d);
      lp2_insert_val(tv, acc);
    }
  }
  if (lane == 0) a.gthr[q] = lp2_enc(tv[LPX_TOPK - 1] - 0.5f * a.margin);
}

struct Lp2Off { unsigned v[8]; };      // byte offsets of the eight DMA pieces of a key block (by value: stays in registers)
__device__ __forceinline__ Lp2Off lp2_offsets_tab(const Lp2Window& w, const int* keytab, int kb, int lane, int W, unsigned rowb, unsigned lane_off) {
  Lp2Off o;      // eight independent LDS reads; rows past the window: its last key (their scores are masked)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = 8 * p + (lane >> 3);
    const int pk = keytab[min(kb * 64 + r, w.nwin - 1)];
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    o.v[p] = (unsigned)((pk >> 16) * W + (pk & 0xffff)) * rowb + lane_off + (unsigned)c * 16u;
  }
  return o;
}
__device__ __forceinline__ Lp2Off lp2_offsets(const Lp2Window& w, int kb, int lane, int W, unsigned rowb, unsigned lane_off) {
  // the lane's rows are 8 window positions apart: ONE division, then steps of 8 columns (with one wave per SIMD every VALU
  // instruction of the skeleton is on the critical path - sixteen integer divisions per key block were ~4 k cycles of it)
  Lp2Off o;
  const int kk0 = kb * 64 + (lane >> 3), last = w.nwin - 1;
  int ky = kk0 / w.ww, kx = kk0 - ky * w.ww;
  const int ly = last / w.ww, lx = last - ly * w.ww;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = 8 * p + (lane >> 3);
    const bool past = kk0 + 8 * p > last;      // rows past the window: its last key (their scores are masked)
    const int yy = w.wy0 + (past ? ly : ky), xx = w.wx0 + (past ? lx : kx);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    o.v[p] = (unsigned)(yy * W + xx) * rowb + lane_off + (unsigned)c * 16u;
    kx += 8;
    while (kx >= w.ww) { kx -= w.ww; ky += 1; }      // (one iteration unless the window is narrower than 8 columns)
  }
  return o;
}
// Key stages travel by LDS-DMA (buffer_load ... lds): RING - 1 stages of 8 KB in flight per wave, no VGPR staging.  (Measured on
// the MI355X against buffer_load_dwordx4 -> VGPR -> ds_write_b128 with one 8 KB stage in flight per wave: 2.15 vs 2.6 ms for the
// ResNet-50 frame - with one wave per SIMD the loop is bound by bytes in flight x memory latency, not by the issue path.)
__device__ __forceinline__ vfs_rsrc_words lp2_frame_rsrc(const bf16_t* hl, int slot, int HW, unsigned rowb) {
  return vfs_make_rsrc_words(reinterpret_cast<const unsigned char*>(hl) + (size_t)slot * HW * rowb, (unsigned)HW * rowb);
}
// the eight pieces of a stage as ONE burst: M0 (the LDS base of a piece) is saved and restored once and stepped by 1 KB between the
// pieces - vfs_dma16_async saves / sets / restores it around every piece (5 scalar instructions and two M0 reads per KB), and in
// this kernel the issue path of the DMA pieces, not the memory behind them, is what the matrix pipe waits for (what-if with
// cache-hot key rows: 1.23 vs 1.34 ms)
__device__ __forceinline__ void lp2_issue(const vfs_rsrc_words& rs, const Lp2Off& o, unsigned char* dst, unsigned soff) {
#ifndef VFS_EMU
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)dst);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %4, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %5, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %6, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %7, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %8, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %9, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %10, %2, %3 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %11, %2, %3 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "s"(rs), "s"(soff), "v"(o.v[0]), "v"(o.v[1]), "v"(o.v[2]), "v"(o.v[3]), "v"(o.v[4]), "v"(o.v[5]), "v"(o.v[6]), "v"(o.v[7])
      : "memory", "scc");
#else
#pragma unroll
  for (int p = 0; p < 8; ++p) vfs_dma16_async(rs, dst + p * 1024, o.v[p], soff);
#endif
}
template <int RH>
__device__ __forceinline__ void lp2_park(float* dst, const f32x16& v, int r0) {      // dst -> [RH / 4][64 lanes][4]: 16-byte stores
#pragma unroll
  for (int r = 0; r < RH; r += 4) *reinterpret_cast<f32x4*>(dst + r * 64) = (f32x4){v[r0 + r], v[r0 + r + 1], v[r0 + r + 2], v[r0 + r + 3]};
}

// NG = 16-channel groups per wave = C / 64 (4: C = 256, 8: 512, 16: 1024)
template <int NG>
__global__ __launch_bounds__(256, 1) void lp2_score_kernel(Lp2Args a) {
  // (a fourth stage - three in flight, the reduction in two rounds to make room - measured the same 2.28 ms: kept at three)
  constexpr int RING = LP2_RING, NST = NG / 2, SBYTES = 64 * 128;
  constexpr int RH = RING == 4 ? 8 : 16;      // accumulator registers parked per reduction round (4 stages of ring: LDS for half a tile)
  static_assert(RING >= 3 && RING - 1 <= NST, "the requests run at most one key block ahead");      // a stage = 32 channels (hi + lo) of 64 key rows = 8 KB per wave
  __shared__ __attribute__((aligned(16))) unsigned char sRing[4][RING][SBYTES];
  __shared__ __attribute__((aligned(16))) float sRed[4][3][RH / 4][64][4];      // [owner wave][source rank][register quad (of a round)][lane][4]
  __shared__ int sKC[64], sThr[64], sCnt[64], sEn[64];
  __shared__ float sEq[64][LP2_BLOCK_QUEUE];      // scores listed for a query in the current key block (feed its running top 10)
  __shared__ float sTop[LPX_TOPK][64];            // the queries' running top 10 of s~ (in LDS: the 512 registers of a lane are taken)
  __shared__ int sKeyTab[LP2_KEYTAB];             // trimmed window: key index -> (y << 16 | x)
  __shared__ int sTabN;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, kgrp = lane >> 5;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  // Work order (a speed matter only): the key rows a workgroup streams are wanted by the ~5 x 5 query tiles whose windows contain
  // them.  Hardware workgroup b runs on XCD b % 8, each XCD behind its own 4 MB L2: every XCD gets a CONTIGUOUS range of the
  // logical order (tile row, key-frame split, tile column), so the workgroups resident on an XCD are the neighbouring tiles of ONE
  // tile row on the SAME key frames, walking their windows (centre-out, same order) in step - PMC: 7.6 GB fetched per ResNet-50
  // frame instead of 12.4 GB (bijective remap, any grid).
  int lb = blockIdx.x;
  if (a.xcd_order) {
    const int nb = gridDim.x, qn = nb >> 3, rn = nb & 7, xcd = lb & 7;
    lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (lb >> 3);
  }
  const int trow = lb / (a.nsplit * tiles_x), rem = lb - trow * (a.nsplit * tiles_x);
  const int split = rem / tiles_x, tcol = rem - split * tiles_x;
  const int qy0 = trow * 8, qx0 = tcol * 8;
  const int stag = a.xcd_order >= 2 ? min(a.xcd_order - 1, 3) * (tcol & 1) : 0;      // (nkb >= 4 for every window this path sees)
  const int fpb = (a.nkeys + a.nsplit - 1) / a.nsplit;
  const int f_begin = split * fpb, f_end = min(a.nkeys, f_begin + fpb);
  const unsigned rowb = (unsigned)C * 4u;      // bytes of one split row (hi + lo)

  // ---- the query tile, resident: B fragments (16 channels x 32 queries) of this wave's channel quarter, hi and lo
  // Every wave names the tiles RELATIVE to the one it owns in the epilogue (keys of half kh x queries of half qhh): accumulator a00
  // is always the own tile (first key rows = half kh, first queries = half qhh), so the reduction starts from a register set known
  // at compile time (selecting one of four accumulators by the wave index was 48 v_cndmask per key block, all exposed)
  const int kh = wave & 1, qhh = wave >> 1;
  bf16x8 qh0[NG], qh1[NG], ql0[NG], ql1[NG];
  {
    const int qa = 32 * qhh + li, qb = 32 * (qhh ^ 1) + li;
    const int ya = min(qy0 + (qa >> 3), H - 1), xa = min(qx0 + (qa & 7), W - 1);      // rows past the map: a valid row, masked below
    const int yb = min(qy0 + (qb >> 3), H - 1), xb = min(qx0 + (qb & 7), W - 1);
    const bf16_t* ba = a.hl + ((size_t)a.qframe * HW + (size_t)(ya * W + xa)) * 2 * C + (size_t)wave * NG * 32 + kgrp * 8;
    const bf16_t* bb = a.hl + ((size_t)a.qframe * HW + (size_t)(yb * W + xb)) * 2 * C + (size_t)wave * NG * 32 + kgrp * 8;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      qh0[g] = *reinterpret_cast<const bf16x8*>(ba + g * 32);
      ql0[g] = *reinterpret_cast<const bf16x8*>(ba + g * 32 + 16);
      qh1[g] = *reinterpret_cast<const bf16x8*>(bb + g * 32);
      ql1[g] = *reinterpret_cast<const bf16x8*>(bb + g * 32 + 16);
    }
  }
  // The query fragments have ARRIVED before the loop starts, and the compiler must know it: its wait-count pass otherwise keeps a
  // `s_waitcnt vmcnt(n)` in front of the first use of every fragment INSIDE the loop body (n counting down to 0 over the unrolled
  // stages) - and those waits also cover the untracked LDS-DMA pieces in flight: the key pipeline was drained once per key block
  // (found in the ISA; the kernel took 2.1 instead of ~1.3 ms for the 21-key ResNet-50 frame).
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), nothing else
  // the query's running top 10 of s~ (sTop, maintained by lane t of wave 0); its 10th entry, the seed and what the other
  // key-frame splits of this tile have reached (a.gthr, device-scope atomic max) give the listing threshold sThr
  int tq_pix = -1, gseen = lp2_enc(-INFINITY);
  if (t < 64) {
    const int y = qy0 + (t >> 3), x = qx0 + (t & 7);
    if (y < H && x < W) { tq_pix = y * W + x; gseen = a.gthr[tq_pix]; }
    sThr[t] = gseen; sCnt[t] = 0; sEn[t] = 0;
#pragma unroll
    for (int i = 0; i < LPX_TOPK; ++i) sTop[i][t] = -INFINITY;
  }

  // ---- the trimmed window of this tile's masked key frames (lp2_row_reach above): row extents by the lanes of wave 0, prefix sums
  // over the rows, then every row writes its keys' positions into the table
  int tab_nwin = 0;
  if (a.trim && a.radius > 0) {
    const int r = a.radius;
    const int ty0 = max(0, qy0 - (r - 1)), ty1 = min(H - 1, qy0 + 7 + (r - 1));
    const int rows = ty1 - ty0 + 1;
    if (rows <= 64) {      // (uniform over the workgroup)
      if (t < 64) {
        int wdt = 0, x0 = 0;
        if (t < rows) {
          const int y = ty0 + t;
          const int dx = lp2_row_reach(max(0, max(qy0 - y, y - (qy0 + 7))), r);
          x0 = max(0, qx0 - dx);
          wdt = min(W - 1, qx0 + 7 + dx) - x0 + 1;
        }
        int incl = wdt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        const int total = __shfl(incl, 63);
        if (t == 0) sTabN = total;
        if (total <= LP2_KEYTAB)
          for (int x = 0; x < wdt; ++x) sKeyTab[incl - wdt + x] = ((ty0 + t) << 16) | (x0 + x);
      }
      __syncthreads();
      tab_nwin = sTabN <= LP2_KEYTAB ? sTabN : 0;
    }
  }

  // the wave's part in the epilogue: scores of 32 keys (half kh) x 32 queries (half qhh); the lane's query
  const int myq = qhh * 32 + li;
  const int qy = qy0 + (myq >> 3), qx = qx0 + (myq & 7);
  const bool q_in = qy < H && qx < W;
  const int qpix = q_in ? qy * W + qx : 0;
  unsigned long long* mylist = a.lists + ((size_t)split * HW + qpix) * a.cap;

  // ---- block sequence: key frames newest first, blocks centre-out.  (cf, ci) = the block being computed, (nf, ni) = the next one
  int total_blocks = 0;
  for (int f = f_begin; f < f_end; ++f) total_blocks += lp2_window(a, f, qy0, qx0, tab_nwin).nkb;
  if (total_blocks == 0) return;      // (uniform; cannot happen for a non-empty split)
  // DMA addressing of a block: lane l of piece p fetches chunk c = (l & 7) ^ ((r >> 1) & 7) of key row r = 8 p + (l >> 3), so that
  // the piece lands linearly (piece base + 16 l) in the XOR-swizzled layout the conflict-free fragment reads below expect
  const unsigned lane_off = (unsigned)wave * NG * 64u;
  int cf = f_end - 1, ci = 0, nf = cf, ni = 1;
  Lp2Window cw = lp2_window(a, cf, qy0, qx0, tab_nwin), nw = cw;
  // ONE set of row offsets (`off`, `rs`): it names the block whose stages are being REQUESTED - the current block until its last
  // stage has been requested (two stages before its end), the next block from then on
  Lp2Off off = cw.tab ? lp2_offsets_tab(cw, sKeyTab, lp2_block_of(0, cw.nkb, stag), lane, W, rowb, lane_off)
                      : lp2_offsets(cw, lp2_block_of(0, cw.nkb, stag), lane, W, rowb, lane_off);
  vfs_rsrc_words rs = lp2_frame_rsrc(a.hl, cw.slot, HW, rowb);
  bool has_next = true;
  if (ni >= nw.nkb) {
    nf -= 1; ni = 0;
    if (nf < f_begin) has_next = false;
    else nw = lp2_window(a, nf, qy0, qx0, tab_nwin);
  }
  unsigned char* ring = &sRing[wave][0][0];
  // Pipeline per wave (its own channel quarter, no cross-wave traffic, no barrier): flat stage counter S, stage S lives in ring slot
  // S % RING, RING - 1 stages are in flight: while stage s of a block is multiplied, stage s + RING - 1 is requested - of this block,
  // or (the last RING - 1 stages) of the next one.  Everything about a stage but its ring slot is known at compile time: with ONE
  // wave per SIMD the wave's own instruction issue is what the stage loop is bound by (phase timers, MEASUREMENTS.md round 4: 1230
  // clk per stage for 768 clk of MFMAs, 1 % of it waiting for data), so the bookkeeping is kept out of the instruction stream.
#pragma unroll
  for (int d = 0; d < RING - 1; ++d)      // (the prologue stays inside the first block: RING - 1 <= NST is asserted above)
    lp2_issue(rs, off, ring + d * SBYTES, (unsigned)d * 128u);
  int slot = 0;                      // ring slot of the stage about to be consumed
  __syncthreads();                   // sThr / sCnt initialised

#if LP2_PF
  // Fragment pipeline (round 5, opt-in build -DLP2_PF=1): the compiler reads every key fragment just in time - two register quads, an `s_waitcnt lgkmcnt(0)`
  // in front of every second MFMA.  Here the fragments of a
  // HALF stage (16 channels: hi and lo rows of both key halves, four quads) are requested one half stage ahead into the other of two
  // register sets, across stage and key-block boundaries (the first half of the next block waits through the epilogue).
  const int R0 = 32 * kh + li, R1 = 32 * (kh ^ 1) + li, sw0 = (R0 >> 1) & 7, sw1 = (R1 >> 1) & 7;
  bf16x8 F[2][4];
  auto read_half = [&](bf16x8 (&f)[4], const unsigned char* st, int p) {
    f[0] = *reinterpret_cast<const bf16x8*>(st + R0 * 128 + (((4 * p + kgrp) ^ sw0) << 4));
    f[1] = *reinterpret_cast<const bf16x8*>(st + R0 * 128 + (((4 * p + 2 + kgrp) ^ sw0) << 4));
    f[2] = *reinterpret_cast<const bf16x8*>(st + R1 * 128 + (((4 * p + kgrp) ^ sw1) << 4));
    f[3] = *reinterpret_cast<const bf16x8*>(st + R1 * 128 + (((4 * p + 2 + kgrp) ^ sw1) << 4));
  };
  static_assert(RING == 3, "the fragment pipeline is written for the three-stage ring");
  vfs_dma_wait<8>();                 // stage 0 of the first block has landed (stage 1 in flight)
  read_half(F[0], ring, 0);
#endif

  for (int blk = 0; blk < total_blocks; ++blk) {
    f32x16 a00, a01, a10, a11;      // [key half][query half] partial scores over this wave's channels
#pragma unroll
    for (int r = 0; r < 16; ++r) { a00[r] = 0.f; a01[r] = 0.f; a10[r] = 0.f; a11[r] = 0.f; }
#if LP2_PF
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      // ---- first half: request the second half's fragments, refill the slot stage s - 1 used, multiply
      const int pslot = slot == 0 ? RING - 1 : slot - 1, nslot = slot == RING - 1 ? 0 : slot + 1;
      read_half(F[1], ring + slot * SBYTES, 1);
      bool dma = true;
      unsigned soff = (unsigned)(s + RING - 1) * 128u;
      if (s + RING - 1 >= NST) {      // (compile time) a stage of the NEXT block
        if (s + RING - 1 == NST && has_next && !(a.dbg & 2)) {
          off = nw.tab ? lp2_offsets_tab(nw, sKeyTab, lp2_block_of(ni, nw.nkb, stag), lane, W, rowb, lane_off)
                       : lp2_offsets(nw, lp2_block_of(ni, nw.nkb, stag), lane, W, rowb, lane_off);
          rs = lp2_frame_rsrc(a.hl, nw.slot, HW, rowb);
        }
        dma = has_next;
        soff = (unsigned)(s + RING - 1 - NST) * 128u;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dma) lp2_issue(rs, off, ring + pslot * SBYTES, soff);
      {
        const int g = 2 * s;
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][0], qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][0], qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][2], qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][2], qh1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][0], ql0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][0], ql1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][2], ql0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][2], ql1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][1], qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][1], qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][3], qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0][3], qh1[g], a11, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- second half: the NEXT stage has landed (the one requested above may stay in flight); request its first half, multiply
      if (s + 1 < NST || has_next) {
        if (s + 2 <= NST - 1 || has_next) vfs_dma_wait<8>(); else vfs_dma_wait<0>();
        read_half(F[0], ring + nslot * SBYTES, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int g = 2 * s + 1;
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][0], qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][0], qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][2], qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][2], qh1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][0], ql0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][0], ql1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][2], ql0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][2], ql1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][1], qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][1], qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][3], qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1][3], qh1[g], a11, 0, 0, 0);
      }
      slot = nslot;
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      // stage (blk, s) has landed once at most the stages requested after it are still in flight: RING - 2 of them, fewer at the end
      // of the last block
      if (s + RING - 2 <= NST - 1 || has_next) vfs_dma_wait<8 * (RING - 2)>();
      else if (RING > 3 && s == NST - 2) vfs_dma_wait<8>();
      else vfs_dma_wait<0>();
      // refill the slot the previous stage used (its fragment reads fed that stage's MFMAs) with stage s + RING - 1
      bool dma = true;
      unsigned soff = (unsigned)(s + RING - 1) * 128u;
      if (s + RING - 1 >= NST) {      // (compile time) a stage of the NEXT block
        if (s + RING - 1 == NST && has_next && !(a.dbg & 2)) {      // its first: name it (dbg 2, what-if with WRONG results: every block re-reads the first block's rows)
          off = nw.tab ? lp2_offsets_tab(nw, sKeyTab, lp2_block_of(ni, nw.nkb, stag), lane, W, rowb, lane_off)
                       : lp2_offsets(nw, lp2_block_of(ni, nw.nkb, stag), lane, W, rowb, lane_off);
          rs = lp2_frame_rsrc(a.hl, nw.slot, HW, rowb);
        }
        dma = has_next;
        soff = (unsigned)(s + RING - 1 - NST) * 128u;
      }
      unsigned char* pdst = ring + (slot == 0 ? RING - 1 : slot - 1) * SBYTES;
      const unsigned char* st = ring + slot * SBYTES;
      const int R0 = 32 * kh + li, R1 = 32 * (kh ^ 1) + li, sw0 = (R0 >> 1) & 7, sw1 = (R1 >> 1) & 7;
      bf16x8 ka0 = *reinterpret_cast<const bf16x8*>(st + R0 * 128 + ((kgrp ^ sw0) << 4));
      bf16x8 kl0 = *reinterpret_cast<const bf16x8*>(st + R0 * 128 + (((2 + kgrp) ^ sw0) << 4));
      bf16x8 ka1 = *reinterpret_cast<const bf16x8*>(st + R1 * 128 + ((kgrp ^ sw1) << 4));
      bf16x8 kl1 = *reinterpret_cast<const bf16x8*>(st + R1 * 128 + (((2 + kgrp) ^ sw1) << 4));
      if (dma) lp2_issue(rs, off, pdst, soff);      // (one burst in front of the MFMAs; a piece between each of the first MFMAs measured the same or worse, twice)
      {
        const int g = 2 * s;
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qh1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, ql0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, ql1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, ql0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, ql1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl1, qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl1, qh1[g], a11, 0, 0, 0);
      }
      {
        const int g = 2 * s + 1;
        ka0 = *reinterpret_cast<const bf16x8*>(st + R0 * 128 + (((4 + kgrp) ^ sw0) << 4));
        kl0 = *reinterpret_cast<const bf16x8*>(st + R0 * 128 + (((6 + kgrp) ^ sw0) << 4));
        ka1 = *reinterpret_cast<const bf16x8*>(st + R1 * 128 + (((4 + kgrp) ^ sw1) << 4));
        kl1 = *reinterpret_cast<const bf16x8*>(st + R1 * 128 + (((6 + kgrp) ^ sw1) << 4));
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qh1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, ql0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, ql1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, ql0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, ql1[g], a11, 0, 0, 0);
        a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qh0[g], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qh1[g], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl1, qh0[g], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl1, qh1[g], a11, 0, 0, 0);
      }
      slot = slot == RING - 1 ? 0 : slot + 1;
      __builtin_amdgcn_sched_barrier(0);      // no motion across stages: the register file is full (Q tile 256 + scores 64 + the stage in flight 32)
    }

#endif

    // ---- the four channel quarters of the 64 x 64 score block meet: tile (kt, qt) belongs to wave kt + 2 qt
    const int kb = lp2_block_of(ci, cw.nkb, stag);
    if (t < 64) {
      // key kk of the window -> (row, column): kk / ww through a float reciprocal, exact here (kk < 2^13, ww <= W: (kk + 0.5) / ww is
      // at least 0.5 / ww away from an integer, the reciprocal's error moves it by < 1e-3) - the integer division the compiler
      // emits is ~45 vector instructions, and the other three waves wait for this one at the barrier below
      const int kk = kb * 64 + t;
      const int ky = (int)(((float)kk + 0.5f) * cw.rww);
      sKC[t] = kk >= cw.nwin ? -1 : cw.tab ? sKeyTab[kk] : (((cw.wy0 + ky) << 16) | (cw.wx0 + kk - ky * cw.ww));
    }
    f32x16 tot = a00;
    const int o10 = (kh ^ 1) + 2 * qhh, o01 = kh + 2 * (qhh ^ 1), o11 = (kh ^ 1) + 2 * (qhh ^ 1);      // owners of the other three tiles
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RH) {
      if (r0 > 0) __syncthreads();      // the previous round has been read
      lp2_park<RH>(&sRed[o10][wave < o10 ? wave : wave - 1][0][lane][0], a10, r0);      // [owner][rank of this wave among the other three]
      lp2_park<RH>(&sRed[o01][wave < o01 ? wave : wave - 1][0][lane][0], a01, r0);
      lp2_park<RH>(&sRed[o11][wave < o11 ? wave : wave - 1][0][lane][0], a11, r0);
      __syncthreads();
#pragma unroll
      for (int src = 0; src < 3; ++src)
#pragma unroll
        for (int r = 0; r < RH; r += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&sRed[wave][src][r >> 2][lane][0]);
          tot[r0 + r] = tot[r0 + r] + v[0]; tot[r0 + r + 1] = tot[r0 + r + 1] + v[1];
          tot[r0 + r + 2] = tot[r0 + r + 2] + v[2]; tot[r0 + r + 3] = tot[r0 + r + 3] + v[3];
        }
    }
    // ---- candidates: list everything inside the circle that may still be in the query's top 10.  Once the threshold is warm almost
    // no score passes it: the sixteen threshold tests come FIRST (two instructions each) and the geometry - key coordinates, circle
    // mask, candidate id - is worked out only for the scores that passed, only in waves that have one (with one wave per SIMD and
    // no MFMA in flight here, every vector instruction of this epilogue is exposed: the mask-first form spent ~190 per key block)
    const float thr_e = (a.dbg & 16) ? -INFINITY : lp2_dec(sThr[myq]) - a.margin;      // (dbg 16, tests: list EVERY in-mask candidate with its s~)
    unsigned long long hot = 0;      // lanes with a score above the threshold: sixteen compares, OR-ed on the scalar side
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) hot |= __ballot(q_in && tot[rg] >= thr_e);      // (queries past the map have no threshold: they must not keep the slow path alive in every edge tile)
    if (hot != 0 && !(a.dbg & 4)) {      // (dbg 4: what-if timing without the lists)
      const int fid = cf * HW;
      // every passing candidate is listed; the query's running top 10 is fed with this lane's BEST TWO of them (four lanes hold a
      // query's 64 scores of the block: at most eight values per query and block, the queue never overflows).  Feeding the first
      // sixteen in arrival order let the threshold of a cold clip creep up by a few percentiles per block: ~500 listed candidates
      // per query in the first nine steps of a clip, and a refinement as long as pass 1 itself.
      float b1 = -INFINITY, b2 = -INFINITY;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        if (q_in && tot[rg] >= thr_e) {
          const int pk = sKC[kh * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kgrp];
          const int cy = pk >> 16, cx = pk & 0xffff;
          const int dy = cy - qy, dx = cx - qx;
          if (pk >= 0 && (cw.r <= 0 || dy * dy + dx * dx < cw.r * cw.r)) {
            const float sc = tot[rg];
            const int idx = atomicAdd(&sCnt[myq], 1);
            if (idx < a.cap) mylist[idx] = ((unsigned long long)(unsigned)(fid + cy * W + cx) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, sc);
            if (sc > b1) { b2 = b1; b1 = sc; }
            else if (sc > b2) b2 = sc;
          }
        }
      }
      if (b1 > -INFINITY) {
        const int e = atomicAdd(&sEn[myq], 1);
        if (e < LP2_BLOCK_QUEUE) sEq[myq][e] = b1;      // (a dropped value would only delay the threshold: it stays a lower bound)
      }
      if (b2 > -INFINITY) {
        const int e = atomicAdd(&sEn[myq], 1);
        if (e < LP2_BLOCK_QUEUE) sEq[myq][e] = b2;
      }
    }
    __syncthreads();      // sRed / sKC are rewritten by the next block; the block's listed scores are complete
    if (t < 64 && tq_pix >= 0) {
      const int n = min(sEn[t], LP2_BLOCK_QUEUE);
      if (n > 0) {
        float tv[LPX_TOPK];
#pragma unroll
        for (int i = 0; i < LPX_TOPK; ++i) tv[i] = sTop[i][t];
        for (int e = 0; e < n; ++e) lp2_insert_val(tv, sEq[t][e]);
#pragma unroll
        for (int i = 0; i < LPX_TOPK; ++i) sTop[i][t] = tv[i];
        sEn[t] = 0;
      }
      const int mine = lp2_enc(sTop[LPX_TOPK - 1][t]);
      const int best = max(mine, gseen);
      // share with the workgroups of the other key-frame splits; what they had reached comes back for the NEXT block (the
      // returned value is first used one iteration later: the atomic's round trip hides behind a block of MFMAs)
      if (!(a.dbg & 8)) {      // (dbg 8: what-if without the exchange between the splits)
        if (mine > gseen) gseen = max(best, atomicMax(&a.gthr[tq_pix], mine));
        else gseen = max(gseen, __hip_atomic_load(&a.gthr[tq_pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      sThr[t] = best;      // (ordered before the next epilogue's read by the barrier behind the next block's reduction stores)
    }

    // ---- advance: (cf, ci) <- (nf, ni), then name the block after it
    if (blk + 1 < total_blocks) {
      cf = nf; ci = ni; cw = nw;
      ni += 1;
      if (ni >= nw.nkb) {
        nf -= 1; ni = 0;
        if (nf < f_begin) has_next = false;
        else nw = lp2_window(a, nf, qy0, qx0, tab_nwin);
      }
    }
  }
  if (t < 64) {
    const int y = qy0 + (t >> 3), x = qx0 + (t & 7);
    if (y < H && x < W) {
      const int n = sCnt[t];
      a.counts[(size_t)split * HW + y * W + x] = min(n, a.cap);
      if (n > a.cap) atomicOr(a.flags, 1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pass 2: one WAVE per query (no workgroup barrier: the four waves of a workgroup are independent)
#define LP2_SURV_CAP 512      // survivors of one query the refinement stages in LDS
__device__ __forceinline__ void lp2_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(256) void lp2_refine_kernel(Lp2Args a) {
  __shared__ __attribute__((aligned(16))) float sQ[4][1024];      // the query row (C <= 1024)
  __shared__ unsigned long long sSurv[4][LP2_SURV_CAP];      // the query's survivors
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HW = a.H * a.W, C = a.C;
  const int q = blockIdx.x * 4 + wave;
  if (q >= HW) return;      // wave-uniform
  // ---- 1. t = the 10th largest listed s~ (ties count as separate candidates).  The lists of the key-frame splits are walked as ONE
  // flat sequence, element i by lane i % 64: coalesced 8-byte loads, ceil(N / 64) rounds.  (One lane per split walking its own list
  // with a dependent load per entry was fine for the ~20 entries of a warm clip and ~250 us for the ~500 entries a query has in
  // the first nine steps of a clip, when pass 1 starts every split from the seeds' weak threshold.)
  __shared__ int sPre[4][LP2_MAX_SPLIT + 1];      // exclusive prefix of the splits' list lengths
  const int cnt = lane < a.nsplit ? min(a.counts[(size_t)lane * HW + q], a.cap) : 0;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }      // (nsplit <= 24 < 32)
  if (lane < a.nsplit) sPre[wave][lane] = incl - cnt;
  const int N = __shfl(incl, a.nsplit - 1);
  lp2_wave_sync();
  const bool staged = N <= LP2_SURV_CAP;      // every entry fits in LDS: the survivors are filtered in place, no second walk
  auto entry = [&](int i) -> unsigned long long {      // element i < N of the flat sequence
    int sp = 0;
    for (int k = 1; k < a.nsplit; ++k) sp += i >= sPre[wave][k] ? 1 : 0;
    return a.lists[((size_t)sp * HW + q) * a.cap + (i - sPre[wave][sp])];
  };
  float tv[LPX_TOPK];
#pragma unroll
  for (int i = 0; i < LPX_TOPK; ++i) tv[i] = -INFINITY;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    if (i < N) {
      const unsigned long long ent = entry(i);
      lp2_insert_val(tv, __builtin_bit_cast(float, (unsigned)ent));
      if (staged) sSurv[wave][i] = ent;
    }
  }
  float t10 = -INFINITY;
  for (int k = 0; k < LPX_TOPK; ++k) {
    float m = tv[0];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d); m = o > m ? o : m; }
    t10 = m;
    const unsigned long long holders = __ballot(tv[0] == m && m > -INFINITY);
    const bool pop = holders != 0 && lane == (int)__ffsll((unsigned long long)holders) - 1;
#pragma unroll
    for (int i = 0; i < LPX_TOPK - 1; ++i) tv[i] = pop ? tv[i + 1] : tv[i];
    tv[LPX_TOPK - 1] = pop ? -INFINITY : tv[LPX_TOPK - 1];
  }
  const float thr = t10 - a.margin;      // (fewer than ten candidates: -inf, everything survives)
  // ---- 2. survivors = {s~ >= t - margin} -> the front of sSurv (wave-wide compaction, 64 elements per step; in place when the
  // entries were staged: a step writes positions <= the ones it has read)
  lp2_wave_sync();
  int nsurv = 0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    const unsigned long long ent = i < N ? (staged ? sSurv[wave][i] : entry(i)) : 0ull;
    const bool keep = i < N && __builtin_bit_cast(float, (unsigned)ent) >= thr;
    const unsigned long long m = __ballot(keep);
    const int pos = nsurv + __popcll(m & ((1ull << lane) - 1ull));
    lp2_wave_sync();      // (in place: every lane has read its element before any lane overwrites one)
    if (keep && pos < LP2_SURV_CAP) sSurv[wave][pos] = ent;
    nsurv += __popcll(m);
  }
  if (nsurv > LP2_SURV_CAP) {      // more near-ties than the refinement holds: the dense kernel redoes the frame
    if (lane == 0) atomicOr(a.flags, 1);
    return;
  }
  lp2_wave_sync();
  // ---- 3. exact scores.  EIGHT lanes share a survivor: lane j of a group reads channels 32 p + 4 j .. + 3 of piece p, so a wave load
  // touches 8 rows x one whole 128-byte line.  (One row per lane - 64 different lines per load instruction, each line looked up by
  // eight instructions - kept the CU's vector-memory address path busy for ~14 k clk per query: the kernel took 0.2-0.28 ms, as long
  // as a sixth of pass 1, for a few dozen rows per query.)  The defining chain acc = fma(k_c, q_c, acc), c ascending, then walks the
  // lanes of the group: lane j continues from lane j - 1 (DPP row_shr:1), lane 0 of the next piece from lane 7 (row_shl:7).  Every
  // lane executes every step; only the lane whose turn it is holds the true accumulator, the other values are never used.
  float ev[LPX_TOPK];
  int ei[LPX_TOPK];
#pragma unroll
  for (int i = 0; i < LPX_TOPK; ++i) { ev[i] = -INFINITY; ei[i] = LPX_NONE; }
  const float* qrow = a.fbank + ((size_t)a.qframe * HW + q) * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<f32x4*>(&sQ[wave][c]) = lpx_ldf4(qrow + c);
  lp2_wave_sync();
  const int grp = lane >> 3, sub = lane & 7;
  for (int b0 = 0; b0 < nsurv; b0 += 8) {
    const bool on = b0 + grp < nsurv;
    const int id = on ? (int)(sSurv[wave][b0 + grp] >> 32) : (int)(sSurv[wave][0] >> 32);      // (idle groups: survivor 0's row, result dropped)
    const int fr = id / HW, px = id - fr * HW;
    const float* krow = a.fbank + ((size_t)a.kslot[fr] * HW + px) * C + 4 * sub;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += 256) {      // 8 pieces of 32 channels in flight per lane (C = 256, 512, 1024: C % 256 == 0)
      f32x4 kv[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) kv[p] = lpx_ldf4(krow + c0 + 32 * p);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(&sQ[wave][c0 + 32 * p + 4 * sub]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float t = __builtin_fmaf(kv[p][0], qv[0], acc);
          t = __builtin_fmaf(kv[p][1], qv[1], t);
          t = __builtin_fmaf(kv[p][2], qv[2], t);
          t = __builtin_fmaf(kv[p][3], qv[3], t);
          acc = s < 7 ? dpp_mov<0x111>(t) : dpp_mov<0x107>(t);      // row_shr:1 / row_shl:7 (the wrap to lane 0 of the group)
        }
      }
    }
    if (on && sub == 0) lpx_insert(ev, ei, acc / a.temperature, id);
  }
  // ---- 4. the query's top 10 under the total order (score desc, candidate id asc): ten rounds of a wave-wide arg-best over
  // the lanes' heads (every lane ends up holding the whole sorted list)
  float bv[LPX_TOPK];
  int bi[LPX_TOPK];
#pragma unroll
  for (int k = 0; k < LPX_TOPK; ++k) {
    float mv = ev[0];
    int mi = ei[0];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ov = __shfl_xor(mv, d);
      const int oi = __shfl_xor(mi, d);
      if (lpx_better(ov, oi, mv, mi)) { mv = ov; mi = oi; }
    }
    bv[k] = mv; bi[k] = mi;
    const bool pop = mi != LPX_NONE && ev[0] == mv && ei[0] == mi;      // candidate ids are unique: exactly one lane pops
#pragma unroll
    for (int i = 0; i < LPX_TOPK - 1; ++i) { ev[i] = pop ? ev[i + 1] : ev[i]; ei[i] = pop ? ei[i + 1] : ei[i]; }
    if (pop) { ev[LPX_TOPK - 1] = -INFINITY; ei[LPX_TOPK - 1] = LPX_NONE; }
  }
  // ---- 5. softmax over the top-k in sorted order, weighted sum of the values (the arithmetic of labelprop_f32_merge_kernel)
  float e[LPX_TOPK], z = 0.f;
#pragma unroll
  for (int k = 0; k < LPX_TOPK; ++k) {
    e[k] = (k < a.topk && bi[k] != LPX_NONE && bv[k] > -INFINITY) ? vexp(bv[k] - bv[0]) : 0.f;
    z = z + e[k];
  }
  float* o = a.out + (size_t)q * a.CO;
  for (int c = lane; c < a.CO; c += 64) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LPX_TOPK; ++k) {
      if (e[k] > 0.f) {
        const int fr = bi[k] / HW, px = bi[k] - fr * HW;
        s = __builtin_fmaf(e[k] / z, a.sbank[((size_t)a.kslot[fr] * HW + px) * a.CO + c], s);
      }
    }
    o[c] = s;
  }
}

// ---------------------------------------------------------------------------------------------
int vfs_option_lp2 = 1;            // 0: always the dense kernel (A/B knob)
int vfs_option_lp2_dbg = 0;        // what-if timing (WRONG results): 2 = cache-hot key traffic, 4 = no lists; 16 = list every in-mask candidate (results unchanged: tests read the s~ of the lists)
int vfs_option_lp2_fpb = 0;        // pass 1: key frames per workgroup; 0 = chosen per launch (vfs_lp2_splits)
int vfs_option_lp2_trim = 1;       // pass 1: windows of masked key frames trimmed to the columns the tile can reach (0: rectangles, A/B knob)
int vfs_option_lp2_cap = 0;        // list entries per (key-frame split, query); 0 = the workspace shared out among the splits in use
int vfs_option_lp2_xcd = -1;       // pass 1 work order: 1 / 2 = XCD-aware (/ staggered), 0 = dispatch order, -1 = by bank width: XCD-aware for
                                   // C = 1024 (ResNet-50: level in time, 2.986 vs 3.004 ms per frame, 4.7 instead of 8.3 GB fetched per launch), dispatch order for
                                   // narrower banks (ResNet-18, C = 256: 1.015 vs 1.054 ms per frame - short key blocks, the tiles of an XCD wait for the same lines)

// Key frames per workgroup: about four workgroups per CU and launch (pass 1 runs ONE workgroup per CU at a time: the query tile fills
// the register file).  Measured on the MI355X, 21 key frames: 1 / 2 / 3 frames per workgroup = 2.21 / 2.24 / 2.27 ms (ResNet-50) and
// 0.78 / 0.75 / 0.77 (ResNet-18) - flat; two workgroups of 11 frames per tile: 0.92.
int vfs_lp2_splits(int H, int W, int nkeys) {
  const int tiles = ((H + 7) / 8) * ((W + 7) / 8);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  }
  int nsplit = (4 * cus + tiles - 1) / tiles;
  nsplit = max(1, min(nsplit, min(nkeys, LP2_MAX_SPLIT)));
  int fpb = (nkeys + nsplit - 1) / nsplit;
  if (vfs_option_lp2_fpb > 0) fpb = min(nkeys, max(vfs_option_lp2_fpb, (nkeys + LP2_MAX_SPLIT - 1) / LP2_MAX_SPLIT));
  return (nkeys + fpb - 1) / fpb;
}

bool vfs_lp2_eligible(int C) { return vfs_option_lp2 && (C == 256 || C == 512 || C == 1024); }

int vfs_labelprop_f32_2pass_launch(Lp2Args a, hipStream_t s) {
  if (!vfs_lp2_eligible(a.C)) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32_2pass: C must be 256, 512 or 1024");
  if (a.nkeys < 1 || a.nkeys > LP_MAX_KEYS) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32_2pass: 1 <= nkeys <= 64");
  if (a.topk < 1 || a.topk > LPX_TOPK) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32_2pass: 1 <= topk <= 10");
  if (a.H >= 32768 || a.W >= 65536 || (long long)a.nkeys * a.H * a.W >= 0x7fffffffLL || (long long)a.H * a.W * a.C * 4 >= 0xFFFFFFF0LL)
    return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32_2pass: map too large");
  if (!(a.temperature > 0.f)) return vfs_set_error(VFS_ERR_ARG, "labelprop_f32_2pass: temperature > 0");
  // |s~ - s| <= EPS for unit rows: 3 * 2^-16 (the dropped lo.lo product and the two representation remainders, Cauchy-Schwarz) +
  // 4 * C * 2^-24 (C roundings of the exact chain, <= 3 C of the matrix unit's partial sums); margin = 2 EPS
  // + 2^-21: the total order is applied to fl(s / temperature), and a correctly rounded division keeps a STRICT order only across a
  // gap of more than one ulp of the quotient (2^-23 relative): with the extra 2^-21 (|s| <= 1) a pruned candidate's exact score
  // lies more than four ulps below each of its ten betters, so it cannot tie with one of them after the division either
  // (round 4 advisor finding: the proof was stated on s, the order on s / temperature)
  a.margin = 2.0f * (3.0f / 65536.0f + 4.0f * (float)a.C / 16777216.0f) + 1.0f / 2097152.0f;
  a.nsplit = vfs_lp2_splits(a.H, a.W, a.nkeys);
  // the list workspace (LP2_MAX_SPLIT x LP2_MAX_CAP entries per query) is shared out among the splits in use: the first steps of a
  // clip have few key frames (few splits) and cold thresholds (long lists)
  a.cap = min(vfs_option_lp2_cap > 0 ? vfs_option_lp2_cap : LP2_LIST_MAX, min(LP2_LIST_MAX, a.entries / a.nsplit));
  if (a.cap < 1) return vfs_set_error(VFS_ERR_ARG, "labelprop_f32_2pass: the workspace holds less than one list entry per (key-frame split, query)");
  const int tiles = ((a.H + 7) / 8) * ((a.W + 7) / 8);
  if (hipMemsetAsync(a.flags, 0, sizeof(int), s) != hipSuccess) return vfs_set_error(VFS_ERR_LAUNCH, "labelprop_f32_2pass: hipMemsetAsync");
  hipLaunchKernelGGL(lp2_seed_kernel, dim3((a.H * a.W + 3) / 4), dim3(256), 0, s, a);
  int rcs = vfs_check_launch("lp2_seed");
  if (rcs) return rcs;
  a.xcd_order = vfs_option_lp2_xcd >= 0 ? vfs_option_lp2_xcd : (a.C >= 1024 ? 1 : 0);
  a.trim = vfs_option_lp2_trim;
  a.dbg = vfs_option_lp2_dbg;
  if (a.C == 256) hipLaunchKernelGGL(lp2_score_kernel<4>, dim3(tiles * a.nsplit), dim3(256), 0, s, a);
  else if (a.C == 512) hipLaunchKernelGGL(lp2_score_kernel<8>, dim3(tiles * a.nsplit), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(lp2_score_kernel<16>, dim3(tiles * a.nsplit), dim3(256), 0, s, a);
  int rc = vfs_check_launch("lp2_score");
  if (rc) return rc;
  hipLaunchKernelGGL(lp2_refine_kernel, dim3((a.H * a.W + 3) / 4), dim3(256), 0, s, a);
  return vfs_check_launch("lp2_refine");
}

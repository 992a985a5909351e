// 3x3 / stride-1 / pad-1 convolution (forward and dgrad) with the input HALO TILE resident in LDS.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-fetches a pixel's 128-byte channel chunk
// once per tap: 9x the L1/L2 traffic and 9x the LDS stores of the activation operand.  Here one
// workgroup owns a SPATIAL tile of output pixels, stages the (TH+2)x(TW+2) halo patch of a
// 64-channel chunk ONCE, and the nine taps read it with shifted row indices:
//     B-fragment row of tap (r,s), pixel (py,px)  =  patch[(py + r') * (TW+2) + px + s']
// (r' = r forward, 2-r dgrad).  Per tap only the BC x 64 weight tile is fetched.
//
// Tiling (v_mfma_f32_16x16x32_bf16, rows = channels, cols = pixels): every wave owns 64 channels x
// 64 pixels (4x4 MFMA tiles, 0.5 KB of LDS fragment reads per MFMA - the main loop is bound by the
// LDS pipe as much as by the matrix pipe, measured with per-phase s_memtime stamps).  The four waves
// are arranged to cover all BC output channels of the block:
//     BC = 128: 2 (channels) x 2 (pixels)  -> 8x16 pixel tile, or two whole 8x8 images
//     BC =  64: 1 (channels) x 4 (pixels)  -> 16x16 pixel tile (less halo, weights amortised 2x)
// Fragment reads are software-pipelined across the k-steps and across the per-tap barrier (the patch
// does not change at a tap boundary), so a wave's LDS reads overlap its own and its neighbours' MFMAs.
#include <type_traits>
#include "vfs_conv.h"

// any offset >= num_records reads as zero; 2^31 leaves room for a scalar offset on top without wrapping
#define OOB_OFFSET 0x80000000u

// LDS rows are 64 bf16 + 16 pad = 160 bytes (32 B x odd): the ds_read_b128 fragment reads of 16
// consecutive rows are bank-conflict free on gfx950 and, being linear, every tap / k-step / MFMA
// tile is an IMMEDIATE offset from one per-lane base register (no address VALU in the main loop).
#define HALO_RS 80
// what-if builds (tools/make_variant_lib.sh; results are garbage, only the time is read): 1 no weight DMA in the tap loop,
// 2 no barrier per tap, 4 no MFMAs (the fragment reads stay), 8 no fragment reads in the deep tap loop, 16 no epilogue, 32 no tap loop (deep)
#ifndef HALO_WHATIF
#define HALO_WHATIF 0
#endif

// lane column lr of 16-pixel group tn of pixel-wave wp -> pixel of the spatial tile.  For the 8-wide
// tile a group is two 8-pixel rows, 10 patch rows apart; the second row is permuted so that the patch
// rows of each ds_read_b128 lane group stay distinct mod 8 (conflict free).
template <bool SMALLW>
__device__ __forceinline__ void halo_pixel(int wp, int tn, int lr, int& ti, int& py, int& px) {
  if (SMALLW) {
    const int c = lr & 7;
    ti = wp;
    py = tn * 2 + (lr >> 3);
    px = lr < 8 ? c : (c < 2 ? c : (c < 4 ? c + 4 : c - 2));
  } else {
    ti = 0;
    py = wp * 4 + tn;
    px = lr;
  }
}

// ---------------- epilogue shared by the halo kernels ----------------
// acc[tm][tn]: wave tile of 64 channels (wc) x 64 pixels (wp).  F = compile-time superset of
// {1: BatchNorm partial statistics, 2: residual add, 4: bias, 8: fused BatchNorm-BACKWARD statistics of
// the output (BnBwdFuse, dgrad)}: the combinations the training step uses get branch-free bodies,
// anything else the runtime-checked one.  sRed: [WAVES_P][2][BC] floats.
//
// The MFMA accumulator layout gives a lane 4 channels (8 bytes) of one pixel per tile; storing that
// directly is 16 dwordx2 stores per lane in 32-byte fragments - store-ISSUE bound (measured: the
// waves of a workgroup queue ~9k cycles behind one another).  Instead each wave transposes its
// 64 x 64 outputs through a private LDS slab `stage` (64 pixel rows of 144 bytes; the caller
// guarantees nobody still reads that memory) and writes whole 128-byte pixel rows: 8 dwordx4 stores
// per lane, 1 KB contiguous per wave instruction.  LDS operations of one wave execute in order, so
// the write -> read hand-over inside the wave needs no barrier.
#define HALO_STAGE_ROW 72   // bf16 elements per staged pixel row (64 + 8 pad)
#define HALO_STAGE_WAVE (64 * HALO_STAGE_ROW)
template <int F, int BC, bool SMALLW, bool RAGGED>
__device__ __forceinline__ void halo_epilogue(const ConvArgs& a, const f32x4 (&acc)[4][4], float* sRed, bf16_t* stage,
                                              int tile, int tn0, int y0, int x0, int c0, int wc, int wp, int lr, int lq, int t) {
  constexpr int TM = 4, TN = 4, WAVES_P = 4 / (BC / 64);
  const ConvGeom& g = a.g;
  const bool do_stats = (F & 1) && a.stats != nullptr;
  const bool do_add = (F & 2) && a.add != nullptr;
  const bool do_bias = (F & 4) && a.bias != nullptr;
  const bool do_bn = (F & 8) && a.bn.partial != nullptr;
  const int lane = t & 63;
  bf16_t* slab = stage + (t >> 6) * HALO_STAGE_WAVE;
  float s1[TM][4], s2[TM][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[tm][r] = 0.f; s2[tm][r] = 0.f; }
  // ragged tiles (right / bottom edge of maps that are not multiples of the tile, e.g. 56 x 56): validity of this
  // lane's pixels in the accumulator layout (bit tn) and in the row-store layout (bit i)
  // (RAGGED is a separate instantiation: exact tilings keep constant masks and pay nothing)
  unsigned okt = RAGGED ? 0u : 0xFu, oki = RAGGED ? 0u : 0xFFu;
  if (RAGGED) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      int ti, py, px;
      halo_pixel<SMALLW>(wp, tn, lr, ti, py, px);
      if (y0 + py < g.H && x0 + px < g.W && tn0 + ti < g.N) okt |= 1u << tn;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = i * 8 + (lane >> 3);
      int ti, py, px;
      halo_pixel<SMALLW>(wp, p >> 4, p & 15, ti, py, px);
      if (y0 + py < g.H && x0 + px < g.W && tn0 + ti < g.N) oki |= 1u << i;
    }
  }
  // fused BatchNorm-backward statistics: this lane's eight (pixel, 8-channel chunk) operands are
  // requested NOW, before the accumulators are converted and staged, and consumed in the row-store loop
  // (straight-line batches, pixels outside the map clamped to the tile's first pixel - always inside: loads inside per-pixel
  // conditionals made the compiler wait for each one separately, see conv_igemm.hip)
  u32x4 bxv[8];
  unsigned bym[8];
  if (do_bn) {
    const int cch = c0 + wc * 64 + (lane & 7) * 8;
    const size_t mfirst = ((size_t)tn0 * g.H + y0) * g.W + x0;
    size_t mrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = i * 8 + (lane >> 3);
      int ti, py, px;
      halo_pixel<SMALLW>(wp, p >> 4, p & 15, ti, py, px);
      mrow[i] = ((oki >> i) & 1u) ? ((size_t)(tn0 + ti) * g.H + (y0 + py)) * g.W + (x0 + px) : mfirst;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) bxv[i] = ld16(a.bn.x + mrow[i] * a.Cout + cch);
    const long long rows = (long long)g.N * g.H * g.W;
    if (a.bn.y && a.bn.relu == VFS_MASK_BITS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bym[i] = mask8_load(a.bn.y, (long long)mrow[i], cch, rows, a.Cout);
    } else if (a.bn.y) {
#pragma unroll
      for (int h = 0; h < 8; h += 4) {
        u32x4 yv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) yv[i] = ld16(a.bn.y + mrow[h + i] * a.Cout + cch);
#pragma unroll
        for (int i = 0; i < 4; ++i) bym[h + i] = mask8_of(yv[i]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // the residual operand is requested for TWO pixel tiles at a time (8 loads in flight per lane), as in conv_igemm.hip
#pragma unroll
  for (int th = 0; th < TN; th += 2) {
  u32x2 adv[2][TM];
  unsigned long long amwv[2] = {0, 0};
  if (do_add) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int tn = th + t2;
      int ti, py, px;
      halo_pixel<SMALLW>(wp, tn, lr, ti, py, px);
      // pixels outside the map: clamped to the tile's first pixel (inside), so that the loads carry no per-lane condition
      const size_t mdst = ((okt >> tn) & 1u) ? ((size_t)(tn0 + ti) * g.H + (y0 + py)) * g.W + (x0 + px) : ((size_t)tn0 * g.H + y0) * g.W + x0;
      const size_t obase = mdst * a.Cout + c0 + wc * 64 + lq * 4;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) adv[t2][tm] = ld8(a.add + obase + tm * 16);
      if (a.add_mask) amwv[t2] = addmask_word<64>(a.add_mask, (long long)mdst, c0 + wc * 64, a.add_rows, a.Cout);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int tn = th + t2;
    const u32x2 (&ad)[TM] = adv[t2];
    const unsigned long long amw = amwv[t2];
    const bool okp = (okt >> tn) & 1u;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
      if (do_bias) {
        const int c = c0 + wc * 64 + tm * 16 + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += a.bias[c + r];
      }
      if (do_add && a.add_mask) {       // the add operand gated by its own ReLU mask (vfs_conv.h)
        const unsigned nib = (unsigned)(amw >> (tm * 16 + lq * 4));
        v[0] += (nib & 1u) ? bflo(ad[tm].x) : 0.f; v[1] += (nib & 2u) ? bfhi(ad[tm].x) : 0.f;
        v[2] += (nib & 4u) ? bflo(ad[tm].y) : 0.f; v[3] += (nib & 8u) ? bfhi(ad[tm].y) : 0.f;
      } else if (do_add) {
        v[0] += bflo(ad[tm].x); v[1] += bfhi(ad[tm].x); v[2] += bflo(ad[tm].y); v[3] += bfhi(ad[tm].y);
      }
      u32x2 pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      st8(&slab[(tn * 16 + lr) * HALO_STAGE_ROW + tm * 16 + lq * 4], pk);
      if (do_stats && okp) {   // statistics of the STORED (bf16) values
        const float q0 = bflo(pk.x), q1 = bfhi(pk.x), q2 = bflo(pk.y), q3 = bfhi(pk.y);
        s1[tm][0] += q0; s2[tm][0] += q0 * q0;
        s1[tm][1] += q1; s2[tm][1] += q1 * q1;
        s1[tm][2] += q2; s2[tm][2] += q2 * q2;
        s1[tm][3] += q3; s2[tm][3] += q3 * q3;
      }
    }
  }
  }
  __builtin_amdgcn_wave_barrier();   // no code: in-order LDS pipe; keeps the compiler (and the CPU emulator) honest
  // whole pixel rows out: lane = (pixel p = 8 i + lane/8, 16-byte chunk lane%8)
  BnFuseLane bl;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = i * 8 + (lane >> 3), ch = lane & 7;
    int ti, py, px;
    halo_pixel<SMALLW>(wp, p >> 4, p & 15, ti, py, px);
    const size_t mdst = ((size_t)(tn0 + ti) * g.H + (y0 + py)) * g.W + (x0 + px);
    const size_t o = mdst * a.Cout + c0 + wc * 64 + ch * 8;
    const u32x4 gv = ld16(&slab[p * HALO_STAGE_ROW + ch * 8]);
    const bool ok = (oki >> i) & 1u;
    if (ok) st16(a.out + o, gv);
    if (do_bn) {
      if (i == 0) {   // statistics group of this wave's pixels: from the tile's first pixel (always inside the map)
        const int img = min(tn0 + (SMALLW ? wp : 0), g.N - 1);      // the second image of a small-map tile may not exist (odd N)
        const size_t m0 = ((size_t)img * g.H + y0) * g.W + x0;
        bnfuse_init(bl, a.bn, a.Cout, (int)(m0 / a.bn.mpg), c0 + wc * 64 + ch * 8);
      }
      if (ok) bnfuse_accum(bl, a.bn, gv, bxv[i], bym[i]);
    }
  }
  if (do_bn) {
    // lanes lane%8 == ch hold partial sums of the same 8 channels for 8 different pixel groups: meet in
    // LDS behind the output slabs, then one row of {S1, S2} per 128 pixels (= per pair of pixel waves)
    float* sB = reinterpret_cast<float*>(stage + 4 * HALO_STAGE_WAVE);
    bnfuse_spill(bl, sB, t >> 6, lane >> 3, 8, 64, lane & 7);
    __syncthreads();
    for (int e = t; e < WAVES_P * BC; e += 256) {
      const int h = e / (2 * BC), rem = e - h * 2 * BC, st = rem / BC, cl = rem - st * BC;
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const int wave = (cl >> 6) * WAVES_P + 2 * h + w2;
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) sum += sB[((size_t)(wave * 8 + grp) * 2 + st) * 64 + (cl & 63)];
      }
      a.bn.partial[((size_t)tile * (WAVES_P / 2) + h) * 2 * a.Cout + st * a.Cout + c0 + cl] = sum;
    }
  }
  if (do_stats) {
    // lanes of one DPP row hold the same channels for 16 different pixels: VALU row reduction, then
    // the pixel-waves of the tile meet in LDS ([pixel wave][stat][channel], 16-byte stores)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      f32x4 r1, r2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        r1[r] = row16_sum(s1[tm][r]);
        r2[r] = row16_sum(s2[tm][r]);
      }
      if (lr == 0) {
        const int cl = wc * 64 + tm * 16 + lq * 4;
        *reinterpret_cast<f32x4*>(&sRed[(wp * 2 + 0) * BC + cl]) = r1;
        *reinterpret_cast<f32x4*>(&sRed[(wp * 2 + 1) * BC + cl]) = r2;
      }
    }
    __syncthreads();
    // one partial row per 128 pixels = per pair of pixel-waves (rows tile*WAVES_P/2 + h)
    const bool coarse = a.coarse_log2 > 0 && a.stats_coarse != nullptr;      // uniform (vfs_conv.h: coarse statistics rows)
    for (int e = t; e < WAVES_P * BC; e += 256) {
      const int h = e / (2 * BC), rem = e - h * 2 * BC, st = rem / BC, cl = rem - st * BC;
      float* dst = a.stats + ((size_t)tile * (WAVES_P / 2) + h) * 2 * a.Cout;
      const float v = sRed[((2 * h) * 2 + st) * BC + cl] + sRed[((2 * h + 1) * 2 + st) * BC + cl];
      if (coarse) vfs_store_agent(dst + st * a.Cout + c0 + cl, v);
      else dst[st * a.Cout + c0 + cl] = v;
    }
    if (coarse) {
      constexpr int RW = WAVES_P / 2;                       // fine rows per workgroup
      const int ncb = a.Cout / BC, rows = (int)(gridDim.x / ncb) * RW;
      const int L = a.coarse_log2, grp = (tile * RW) >> L, row0 = grp << L, nrow = min(1 << L, rows - row0);
      vfs_stats_coarsen_tail(a, grp, row0, nrow, nrow / RW, grp * ncb + c0 / BC, c0, BC);
    }
  }
}
template <int BC, bool SMALLW, bool RAGGED>
__device__ __forceinline__ void halo_epilogue_dispatch(const ConvArgs& a, const f32x4 (&acc)[4][4], float* sRed, bf16_t* stage,
                                                       int tile, int tn0, int y0, int x0, int c0, int wc, int wp, int lr, int lq, int t) {
  const int flags = (a.stats ? 1 : 0) | (a.add ? 2 : 0) | (a.bias ? 4 : 0) | (a.bn.partial ? 8 : 0);
  if (flags == 1) halo_epilogue<1, BC, SMALLW, RAGGED>(a, acc, sRed, stage, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
  else if (flags == 2) halo_epilogue<2, BC, SMALLW, RAGGED>(a, acc, sRed, stage, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
  else if (flags == 0) halo_epilogue<0, BC, SMALLW, RAGGED>(a, acc, sRed, stage, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
  else if (flags == 8) halo_epilogue<8, BC, SMALLW, RAGGED>(a, acc, sRed, stage, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
  else if (flags == 10) halo_epilogue<10, BC, SMALLW, RAGGED>(a, acc, sRed, stage, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
  else halo_epilogue<15, BC, SMALLW, RAGGED>(a, acc, sRed, stage, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
}

// NW = weight stages.  2: the tap's weight tile is requested one tap ahead (two workgroups per CU hide the round trip for
// each other: the large maps).  4 (round 5): the deep stages - 16 x 16 / 8 x 8 maps, at most one workgroup per CU, 36 - 72 taps
// in ONE dependent chain - waited a full L2 round trip per tap (~1 us for 0.2 us of MFMAs: 37 / 65 us per launch for 19 GFLOP);
// with a ring of four tiles the DMA runs three taps ahead and a tap costs its MFMAs.
template <int BC, bool DGRAD, bool SMALLW, bool RAGGED, int NW = 2>
__global__ __launch_bounds__(256, NW > 2 ? 1 : 2) void conv3x3_halo_kernel(ConvArgs a) {
  constexpr int WAVES_C = BC / 64, WAVES_P = 4 / WAVES_C;      // wave grid: channels x pixels
  constexpr int TW = SMALLW ? 8 : 16, TH = SMALLW ? 8 : 4 * WAVES_P, TI = SMALLW ? WAVES_P : 1;
  constexpr int PW = TW + 2, PH = TH + 2;
  constexpr int PROWS = TI * PH * PW;            // patch rows of 64 channels: 180 / 200 / 324
  constexpr int PLD = (PROWS * 8 + 255) / 256;   // 16-byte patch loads per thread
  constexpr int TM = 4, TN = 4;
  // DEEP (NW > 2): the patch travels by LDS-DMA as well (lanes of padding pixels are sent out of range: the DMA writes zeros -
  // tools/probe_dma_oob.hip), into TWO patch buffers: no register stage, no compiler-tracked load in the loop (a tracked load's
  // s_waitcnt would drain the ring at every chunk boundary), no second barrier at a chunk boundary
  constexpr bool DEEP = NW > 2;
  // LDS row stride (bf16): 160 bytes for the 16-row fragments of the two-stage schedule, 144 bytes (an ODD multiple of 16: the
  // 32 rows of a 32x32x16 fragment fall into 16 distinct 16-byte bank groups per ds_read_b128 lane group) for the deep one
  constexpr int RS = DEEP ? 72 : HALO_RS;
  constexpr int PPQ = ((PROWS * RS * 2 + 4095) / 4096) * 4, PQ = PPQ / 4;   // DMA pieces (1 KB) per patch buffer / per wave
  constexpr int PSZ = DEEP ? PPQ * 512 : PROWS * RS;                       // bf16 elements of one patch buffer
  constexpr int NPB = DEEP ? 2 : 1;
  // one arena: [patch buffer(s) | NW weight tiles]; after the last tap the epilogue re-uses it as output stage
  constexpr int NQD = ((BC * RS * 2 + 4095) / 4096) * 4;          // DEEP: DMA pieces of a weight stage, padded so that every wave moves the same number
  constexpr int WST = DEEP ? NQD * 512 : BC * RS;                 // bf16 elements between weight stages
  __shared__ __attribute__((aligned(16))) bf16_t smem[NPB * PSZ + NW * WST];
  __shared__ __attribute__((aligned(16))) float sRed[WAVES_P][2][BC];
  static_assert((NPB * PSZ + NW * WST) * 2 >= 4 * HALO_STAGE_WAVE * 2 + 4 * 8 * 2 * 64 * 4, "output stage + statistics do not fit");
  bf16_t* const sP = smem;
  bf16_t* const sW = smem + NPB * PSZ;

  const ConvGeom g = a.g;                        // FWD: H,W,C = input; DGRAD: H,W,C = dY (same H,W)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wc = wave / WAVES_P, wp = wave - wc * WAVES_P;
  const int lr = lane & 15, lq = lane >> 4;
  const int ncb = a.Cout / BC;
  // XCD-aware tile order (as in conv_igemm.hip): hardware workgroup b runs on XCD b % 8 behind its own L2; neighbouring spatial
  // tiles share halo rows (an 8x16 tile reads a 10x18 patch) and the channel blocks of a tile share the whole patch - in plain
  // order they sit on eight different XCDs.  Every XCD gets a contiguous range of logical blocks (bijective remap).
  int bx = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bx >> 3);
  }
  const int tile = bx / ncb, cb = bx - tile * ncb;
  const int c0 = cb * BC;
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;     // ragged edge tiles are masked in the epilogue
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tn0 = (tile / (tiles_x * tiles_y)) * TI;
  const int y0 = ty * TH, x0 = tx * TW;
  const int j = t & 7, row0 = t >> 3;
  const int nchunk = g.C >> 6;

  // ---- patch loads: slot k covers patch row (t>>3) + 32k, chunk j; offset fixed for the block
  unsigned poff[PLD];
#pragma unroll
  for (int k = 0; k < PLD; ++k) {
    const int pr = row0 + 32 * k;
    unsigned off = OOB_OFFSET;
    if (pr < PROWS) {
      const int ti = pr / (PH * PW), rem = pr - ti * (PH * PW);
      const int py = rem / PW, px = rem - py * PW;
      const int y = y0 - 1 + py, x = x0 - 1 + px, n = tn0 + ti;
      if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W && n < g.N)
        off = (unsigned)((((size_t)(n * g.H + y) * g.W + x) * g.C + j * 8) * 2);
    }
    poff[k] = off;
  }
  // weights by LDS-DMA: the padded BC x 160-byte tile is NQ consecutive 1-KB pieces; wave w moves
  // pieces w, w+4, ...; lane l of piece q fills bytes q*1024 + 16*l = (row, col) of the tile (lanes
  // that land in the 32-byte row padding fetch column 0 - never read).  Cout % BC == 0: in range.
  const vfs_rsrc_words wrs = vfs_make_rsrc_words(a.wgt, (unsigned)((size_t)a.Cout * g.Ktot * 2));
  constexpr int NQ = BC * RS * 2 / 1024, WQ = (NQ + 3) / 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned wvoff[WQ];
#pragma unroll
  for (int i = 0; i < WQ; ++i) {
    const int pos = (i * 4 + wave_u) * 1024 + lane * 16;
    const int row = pos / (RS * 2), col = pos - row * (RS * 2);
    wvoff[i] = (unsigned)(((size_t)(c0 + (row < BC ? row : 0)) * g.Ktot) * 2 + (col < 128 ? col : 0));
    if (DEEP && row >= BC) wvoff[i] = OOB_OFFSET;      // the padding pieces of the deep schedule's stage: zero fill
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.src, 0, (unsigned)((size_t)g.N * g.H * g.W * g.C * 2), 0x00020000);

  u32x4 pr_[PLD];
  // folded BatchNorm-apply + ReLU of the input (ConvArgs::in_bnp): this thread's 8 channels of the chunk
  const bool bn_in = !DGRAD && a.in_bnp != nullptr;
  const float* bn_in_p = bn_in ? a.in_bnp + (size_t)(tn0 / a.in_npg) * 4 * g.C + j * 8 : nullptr;
  f32x4 isc0, isc1, ish0, ish1;
  auto load_patch = [&](int cc) {
#pragma unroll
    for (int k = 0; k < PLD; ++k) pr_[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, poff[k], cc * 128, 0);
    if (bn_in) {
      const float* p = bn_in_p + cc * 64;
      isc0 = *reinterpret_cast<const f32x4*>(p); isc1 = *reinterpret_cast<const f32x4*>(p + 4);
      ish0 = *reinterpret_cast<const f32x4*>(p + g.C); ish1 = *reinterpret_cast<const f32x4*>(p + g.C + 4);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int k = 0; k < PLD; ++k) {
      const int pr = row0 + 32 * k;
      if (pr < PROWS) {
        u32x4 v = pr_[k];
        if (bn_in && poff[k] != OOB_OFFSET) v = bn_relu_vec(v, isc0, isc1, ish0, ish1);   // padding stays zero
        st16(&sP[pr * RS + j * 8], v);
      }
    }
  };
  auto dma_w = [&](int cc, int tap, int buf) {
    const int wcol = (tap * g.C + cc * 64) * 2;
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
      const int q = i * 4 + wave_u;
      if (DEEP || NQ % 4 == 0 || q < NQ) vfs_dma16_async(wrs, sW + buf * WST + q * 512, wvoff[i], wcol);
    }
  };

  // DEEP: the patch as PQ DMA pieces per wave: lane l of piece q fills bytes q * 1024 + 16 l = (patch row, column) of the padded
  // buffer; padding pixels, the 32-byte row padding and the rows past the patch are sent out of range (zeros)
  const vfs_rsrc_words xrw = vfs_make_rsrc_words(a.src, (unsigned)((size_t)g.N * g.H * g.W * g.C * 2));
  unsigned pdoff[DEEP ? PQ : 1];
  if (DEEP) {
#pragma unroll
    for (int i = 0; i < PQ; ++i) {
      const int pos = (i * 4 + wave_u) * 1024 + lane * 16;
      const int pr = pos / (RS * 2), col = pos - pr * (RS * 2);
      unsigned off = OOB_OFFSET;
      if (pr < PROWS && col < 128) {
        const int ti = pr / (PH * PW), rem = pr - ti * (PH * PW);
        const int py = rem / PW, px = rem - py * PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px, n = tn0 + ti;
        if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W && n < g.N)
          off = (unsigned)((((size_t)(n * g.H + y) * g.W + x) * g.C) * 2 + col);
      }
      pdoff[i] = off;
    }
  }
  auto dma_patch = [&](int cc, int buf) {
#pragma unroll
    for (int i = 0; i < PQ; ++i) vfs_dma16_async(xrw, sP + buf * PSZ + (i * 4 + wave_u) * 512, pdoff[i], cc * 128);
  };

  // ---- per-lane fragment bases: everything else in the main loop is an immediate offset
  int pbase[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int ti, py, px;
    halo_pixel<SMALLW>(wp, tn, lr, ti, py, px);
    pbase[tn] = (ti * (PH * PW) + py * PW + px) * RS + lq * 8;
  }
  const int abase = (wc * 64 + lr) * RS + lq * 8;
  auto tap_shift = [](int tap) {
    const int r = tap / 3, s = tap - 3 * r;
    return (DGRAD ? (2 - r) * PW + (2 - s) : r * PW + s) * RS;
  };
  auto read_b = [&](bf16x8* f, int tap, int kk) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) f[tn] = *reinterpret_cast<const bf16x8*>(&sP[pbase[tn] + tap_shift(tap) + kk * 32]);
  };
  auto read_a = [&](bf16x8* f, const bf16_t* wsrc, int kk) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) f[tm] = *reinterpret_cast<const bf16x8*>(wsrc + tm * 16 * RS + kk * 32);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if HALO_WHATIF & 64
  typedef __attribute__((ext_vector_type(16))) float wf32x16;
  wf32x16 wacc[2][2] = {};
#endif
  auto mma = [&](const bf16x8* af, const bf16x8* bf) {
#if HALO_WHATIF & 64      // timing only (WRONG results): the same FLOP on eight 32x32x16 MFMAs per k-step instead of sixteen 16x16x32
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) wacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2 * i + h], bf[2 * j + h], wacc[i][j], 0, 0, 0);
    return;
#endif
#if HALO_WHATIF & 4
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) asm volatile("" ::"v"(af[tm]), "v"(bf[tm]));
#else
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
#endif
  };

  const int S = 9 * nchunk;      // steps of the flat (chunk, tap) sequence; step s = 9 cc + tap
  if constexpr (DEEP) {
    // ---------------- deep schedule (round 5) ----------------
    // What-if builds of the two-stage loop on the 16 x 16 / 8 x 8 layers (one workgroup per CU, one wave per SIMD) showed its
    // phases ADDING UP instead of overlapping - fragment reads + MFMAs + weight DMA + barriers, an in-order wave that reads,
    // waits, multiplies, then issues DMA - and tools/probe_mfma_rate.hip showed the 16x16x32 MFMA sustaining 1.47 PFLOP/s
    // chip-wide in this pattern where the 32x32x16 one sustains 2.43 (MEASUREMENTS.md).  Here
    //   * v_mfma_f32_32x32x16_bf16, wave tile 64 channels x 64 pixels = 2 x 2 tiles; a tap's 16 MFMAs carry, in their issue
    //     shadows, the 16 fragment reads of the NEXT tap (second register set) and the DMA pieces of the step FOUR taps ahead
    //   * stage s % NW holds step s; at the top of tap s that stage is free (its fragments were read during tap s - 1, behind a
    //     barrier) and takes step s + NW; stage s + 1 is being read, s + 2 must land by the end of the tap, s + 3 is in flight
    //   * the patch of chunk cc + 1 is requested at the top of chunk cc (other buffer) and has landed by the end of its tap 2
    //   * the accumulators meet the shared epilogue through one LDS transposition into the 16x16 layout it is written for
    // K is summed 16 at a time instead of 32: results equal the two-stage kernel's to fp32 rounding, not bit for bit.
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    // The tap body is BRANCH-FREE (round 5 what-ifs: with the wait ladder and the tail / last-chunk conditions compiled into ~15
    // scalar branches per tap, a tap of MFMAs alone took 0.40 us where the bare MFMA stream takes 0.22): every tap issues exactly
    // WQ weight pieces per wave and every chunk PQ patch pieces - past the end of the sequence they are sent out of range
    // (zero fill into a stage / buffer nobody reads any more) - so the wait counts are compile-time constants of the tap index.
    static_assert(NW == 4 && WQ * 4 == NQD, "the deep schedule's wait counts assume a four-stage ring of equal pieces");
    const vfs_lds_t lds_w = vfs_lds_addr(sW), lds_p = vfs_lds_addr(sP);
    dma_patch(0, 0);
#pragma unroll
    for (int d = 0; d < NW; ++d) dma_w(d / 9, d % 9, d);          // S >= 9 > NW
    vfs_dma_wait_all();
    __syncthreads();
    // fragment addresses of the 32-row layout: lane l = row l % 32 (channel / pixel), k offset 8 (l / 32) inside a 16-deep step
    const int l32 = lane & 31, lk = (lane >> 5) * 8;
    const int abase32 = (wc * 64 + l32) * RS + lk;
    int pbase32[2];
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {      // MFMA column l % 32 of column tile tq = pixel (tn = 2 tq + (l % 32) / 16, lr = l % 16) of the epilogue's layout
      int ti, py, px;
      halo_pixel<SMALLW>(wp, 2 * tq + (l32 >> 4), l32 & 15, ti, py, px);
      pbase32[tq] = (ti * (PH * PW) + py * PW + px) * RS + lk;
    }
    bf16x8 fa[2][4][2], fb[2][4][2];                               // [register set][16-deep k step][tile]
    const bf16_t* sPc = sP;                                        // patch buffer of the current chunk
    auto rd_a = [&](int buf, int ks, int tq) { return *reinterpret_cast<const bf16x8*>(sW + buf * WST + abase32 + tq * 32 * RS + ks * 16); };
    auto rd_b = [&](const bf16_t* patch, int tap, int ks, int tq) { return *reinterpret_cast<const bf16x8*>(&patch[pbase32[tq] + tap_shift(tap) + ks * 16]); };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int tq = 0; tq < 2; ++tq) { fa[0][ks][tq] = rd_a(0, ks, tq); fb[0][ks][tq] = rd_b(sPc, 0, ks, tq); }
    __syncthreads();              // everybody holds step 0's fragments: tap 0 may overwrite stage 0 with step NW
    f32x16 acc32[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    int wbuf = 0, pbuf = 0;
    auto chunk = [&](auto parity, int cc) {
      constexpr int P = decltype(parity)::value;
      // the next chunk's patch into the other buffer (past the last chunk: out of range, zero fill)
      {
        const unsigned pso = cc + 1 < nchunk ? (unsigned)(cc + 1) * 128u : OOB_OFFSET;
        const vfs_lds_t dst = lds_p + (vfs_lds_t)((pbuf ^ 1) * PSZ * 2);
#pragma unroll
        for (int i = 0; i < PQ; ++i) vfs_dma16_async_at(xrw, dst + (vfs_lds_t)((i * 4 + wave_u) * 1024), pdoff[i], pso);
      }
      const bf16_t* sPn = sP + (pbuf ^ 1) * PSZ;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = (P * 9 + tap) & 1, nxt = cur ^ 1;
        const int s = cc * 9 + tap;
        const int ntap = (tap + NW) % 9, ncc = cc + (tap + NW) / 9;
        const unsigned wcol = s + NW < S ? (unsigned)((ntap * g.C + ncc * 64) * 2) : OOB_OFFSET;   // past the end: zero fill
        const int nstage = wbuf + 1 == NW ? 0 : wbuf + 1;            // stage of step s + 1
        const vfs_lds_t wdst = lds_w + (vfs_lds_t)(wbuf * WST * 2 + wave_u * 1024);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (!(HALO_WHATIF & 8)) {
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {      // (the last tap of the last chunk reads a stage / buffer of zeros: never used)
              fa[nxt][ks][tq] = rd_a(nstage, ks, tq);
              fb[nxt][ks][tq] = tap < 8 ? rd_b(sPc, tap + 1, ks, tq) : rd_b(sPn, 0, ks, tq);
            }
          }
          if (!(HALO_WHATIF & 1)) {               // pieces 2 ks, 2 ks + 1 of this wave's WQ
#pragma unroll
            for (int i = 2 * ks; i < 2 * ks + 2 && i < WQ; ++i) vfs_dma16_async_at(wrs, wdst + (vfs_lds_t)(i * 4096), wvoff[i], wcol);
          }
#if !(HALO_WHATIF & 4)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][ks][i], fb[cur][ks][j], acc32[i][j], 0, 0, 0);
#else
          asm volatile("" ::"v"(fa[cur][ks][0]), "v"(fb[cur][ks][0]), "v"(fa[cur][ks][1]), "v"(fb[cur][ks][1]));
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        // step s + 2 has landed (this wave's pieces; everybody's behind the barrier): steps s + 3, s + 4 - and, in a chunk's first
        // two taps, the next chunk's patch pieces, issued behind step s + 3's - stay in flight
        if (tap <= 1) vfs_dma_wait<2 * WQ + PQ>(); else vfs_dma_wait<2 * WQ>();
        if (!(HALO_WHATIF & 2)) __syncthreads();
        wbuf = nstage;
      }
      pbuf ^= 1;
      sPc = sPn;
    };
    for (int cc = 0; cc < ((HALO_WHATIF & 32) ? 0 : nchunk); cc += 2) {
      chunk(std::integral_constant<int, 0>{}, cc);
      if (cc + 1 < nchunk) chunk(std::integral_constant<int, 1>{}, cc + 1);
    }
    vfs_dma_wait_all();                // the zero-fill pieces of the last taps
    __syncthreads();
    // ---- accumulators -> the 16x16 layout of the epilogue, through LDS (the ring is idle: the last tap waited for everything)
    // 32x32 tile (i, j), register 4 q + r of lane l: channel 32 i + 8 q + 4 (l / 32) + r, pixel index 32 j + l % 32
    constexpr int TR = 68;                                         // floats per pixel row of the transposition slab (64 + 4: odd multiple of 16 bytes)
    static_assert((NPB * PSZ + NW * WST) * 2 >= 4 * 64 * TR * 4, "transposition slab does not fit");
    float* slab32 = reinterpret_cast<float*>(smem) + wave * (64 * TR);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(&slab32[(32 * j + l32) * TR + 32 * i + 8 * q + 4 * (lane >> 5)]) =
              (f32x4){acc32[i][j][4 * q], acc32[i][j][4 * q + 1], acc32[i][j][4 * q + 2], acc32[i][j][4 * q + 3]};
    __builtin_amdgcn_wave_barrier();   // wave-private slab, in-order LDS pipe
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = *reinterpret_cast<const f32x4*>(&slab32[(tn * 16 + lr) * TR + tm * 16 + lq * 4]);
    __syncthreads();                   // the epilogue's output stage overlaps the other waves' slabs
  } else {
  // ---------------- two-stage schedule: the tap's weight tile is requested one tap ahead ----------------
  load_patch(0);
  dma_w(0, 0, 0);
  store_patch();
  vfs_dma_wait_all();
  __syncthreads();
  int wbuf = 0;
  bf16x8 b0[TN], b1[TN], a0[TM], a1[TM];
  read_b(b0, 0, 0);
  for (int cc = 0; cc < nchunk; ++cc) {
    const bool more_chunks = cc + 1 < nchunk;
    if (more_chunks) load_patch(cc + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const bool last_tap = tap == 8;
      const bool more = !last_tap || more_chunks;
      // next tap's weights: DMA issued FIRST so that the L2 round trip hides under this tap's MFMAs
      if (more && !(HALO_WHATIF & 1)) dma_w(last_tap ? cc + 1 : cc, last_tap ? 0 : tap + 1, wbuf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const bf16_t* wsrc = sW + wbuf * (BC * RS) + abase;
      // k-step 0 operands: b0 was read before the barrier; k-step 1 operands are read now and
      // consumed after the first MFMA block
      read_a(a0, wsrc, 0);
      read_a(a1, wsrc, 1);
      read_b(b1, tap, 1);
      mma(a0, b0);
      if (!last_tap) read_b(b0, tap + 1, 0);      // next tap, same patch: in flight across the barrier
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (last_tap && more_chunks) {
        __syncthreads();                           // every wave is done with this chunk's patch
        store_patch();
      }
      vfs_dma_wait_all();                          // this wave's pieces of the next weight tile landed
      if (!(HALO_WHATIF & 2)) __syncthreads();
      if (last_tap && more_chunks) read_b(b0, 0, 0);
      wbuf ^= 1;
    }
  }
  }

  // the loop's final barrier has passed: no wave reads the patch / weight tiles any more
  if (HALO_WHATIF & 16) return;
#if HALO_WHATIF & 64
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[2 * i + (e >> 3)][2 * j + ((e >> 2) & 1)][e & 3] += wacc[i][j][e];
#endif
  halo_epilogue_dispatch<BC, SMALLW, RAGGED>(a, acc, &sRed[0][0][0], smem, tile, tn0, y0, x0, c0, wc, wp, lr, lq, t);
}

int vfs_option_halo_deep_max = 256;   // the four-stage weight ring (one workgroup per CU) for launches of at most this many workgroups (0: never)

template <int BC, bool DGRAD, bool SMALLW>
static int launch_halo(const ConvArgs& a0, hipStream_t stream) {
  ConvArgs a = a0;
  a.xcd_swizzle = vfs_option_halo_xcd;
  const int WAVES_P = 4 / (BC / 64);
  const int TW = SMALLW ? 8 : 16, TH = SMALLW ? 8 : 4 * WAVES_P, TI = SMALLW ? WAVES_P : 1;
  const int tiles = ((a.g.N + TI - 1) / TI) * ((a.g.H + TH - 1) / TH) * ((a.g.W + TW - 1) / TW);
  const int ncb = a.Cout / BC;
  const bool ragged = a.g.H % TH != 0 || a.g.W % TW != 0;
  if (BC == 128 && !a.in_bnp && tiles * ncb <= vfs_option_halo_deep_max) {
    constexpr int NWD = BC == 128 ? 4 : 2;     // (only the 128-channel tile has equal DMA piece counts in every wave)
    if (ragged) hipLaunchKernelGGL((conv3x3_halo_kernel<BC, DGRAD, SMALLW, true, NWD>), dim3(tiles * ncb), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<BC, DGRAD, SMALLW, false, NWD>), dim3(tiles * ncb), dim3(256), 0, stream, a);
    return vfs_check_launch("conv3x3_halo");
  }
  if (ragged) hipLaunchKernelGGL((conv3x3_halo_kernel<BC, DGRAD, SMALLW, true>), dim3(tiles * ncb), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv3x3_halo_kernel<BC, DGRAD, SMALLW, false>), dim3(tiles * ncb), dim3(256), 0, stream, a);
  return vfs_check_launch("conv3x3_halo");
}

int vfs_option_halo_xcd = 1;          // XCD-aware tile order of the halo kernels (A/B knob)
int vfs_option_halo_min_fill = 70;    // percent of a ragged tiling that must be real pixels (100: exact tilings only)

// eligibility: 3x3 / stride 1 / pad 1, 64-channel granularity, and a spatial tiling that keeps the
// per-128-pixel statistics rows aligned with the two halves of the batch:
//   Cout % 128 == 0:  8x16 tiles (H % 8, W % 16), or whole 8x8 images in pairs (N even)
//   otherwise      : 16x16 tiles (H % 16, W % 16)
bool vfs_conv_halo_eligible(const ConvArgs& a, int mode) {
  const ConvGeom& g = a.g;
  if (mode != GATHER_FWD && mode != GATHER_DGRAD) return false;
  if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1 || g.dil != 1) return false;
  if (a.Cout % 64 || g.C % 64) return false;
  if (g.H != g.Ho || g.W != g.Wo) return false;
  if (vfs_small_map(g.H, g.W) && a.Cout % 128 == 0) return g.N % 2 == 0;
  // other maps: 8x16 (Cout % 128 == 0) or 16x16 tiles; edge tiles may be ragged (masked stores / statistics) as long
  // as the tiles are mostly full: 56 x 56 -> 87 / 77 %, 28 x 28 and 14 x 14 -> 77 %, 7 x 7 -> 38 % (not taken)
  const int th = a.Cout % 128 == 0 ? 8 : 16, tw = 16;
  const long long cover = (long long)((g.H + th - 1) / th * th) * ((g.W + tw - 1) / tw * tw);
  return (long long)g.H * g.W * 100 >= cover * vfs_option_halo_min_fill;
}

int vfs_conv_halo_dispatch(const ConvArgs& a, int mode, hipStream_t stream) {
  const bool wide = (a.Cout % 128 == 0), smallw = vfs_small_map(a.g.H, a.g.W), dg = mode == GATHER_DGRAD;
  if (wide) {
    if (dg) return smallw ? launch_halo<128, true, true>(a, stream) : launch_halo<128, true, false>(a, stream);
    return smallw ? launch_halo<128, false, true>(a, stream) : launch_halo<128, false, false>(a, stream);
  }
  return dg ? launch_halo<64, true, false>(a, stream) : launch_halo<64, false, false>(a, stream);
}

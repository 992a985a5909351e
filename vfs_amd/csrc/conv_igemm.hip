// Implicit-GEMM convolution for gfx950: forward, dgrad (stride 1 and 2) and the 7x7 stem share
// one kernel.
//
//   out[pixel][chan] = sum_k Wt[chan][k] * G[pixel][k]   (+bias[chan]) (+add[pixel][chan])
//
// Replaces the torch conv2d / linear calls of the reference's hot path
// (mmaction/models/backbones/resnet.py:51-73,163-191,267-277,425-434 via mmcv ConvModule;
// mmaction/models/heads/sim_siam_head.py:78-111) and their autograd dgrad.
//
// Tiling (one workgroup = 4 waves of 64 lanes):
//   * 128 pixels x BC channels (BC = 64|128) per workgroup, K-steps of 64
//   * MFMA v_mfma_f32_16x16x32_bf16, rows(i)=channels (A operand = packed weights),
//     cols(j)=pixels (B operand = gathered NHWC activations); each wave owns (BC/2) x 64
//   * operand rows are fetched with buffer_load_dwordx4 through a raw buffer descriptor: a
//     pixel row's byte offset is  base(row) + delta(tap, channel chunk)  with delta uniform per
//     workgroup, and padding taps are sent out of range (hardware returns zeros), so a K-step
//     costs ~4 VALU per load instead of a re-derivation of (n, h, w) and a divergent branch
//   * global -> VGPR -> XOR-swizzled LDS, double-buffered: tile kt+1 is in flight while tile kt
//     feeds the MFMAs; one barrier per K-step
//   * stride-2 dgrad runs per OUTPUT PARITY CLASS (blockIdx.y = (h&1, w&1)): a class only sees the
//     taps r = (h+pad) mod 2, s = (w+pad) mod 2 (1, 2, 2 or 4 of the 9 taps of a 3x3), so no MFMA
//     multiplies the zeros a gather over the dilated gradient would insert
//   * epilogue: lane holds 4 consecutive channels of one pixel -> one 8-byte NHWC store;
//     optional per-channel (sum, sum of squares) of the bf16-ROUNDED outputs, reduced
//     wave-wide with shuffles and written as one deterministic partial per pixel-block
//     (consumed by bn_reduce_rows: BatchNorm batch statistics without an extra pass)
#include "vfs_igemm_epi.h"

#define OOB_OFFSET 0xFFFFFFF0u   // >= any num_records: the load returns zeros
#define XCD_SWIZZLE 1
#ifndef IGEMM_WHATIF
#define IGEMM_WHATIF 0     // what-if builds of the PIPE 5 ring (WRONG results, timing only): 1 no MFMAs, 2 no DMA waits, 4 no DMA issue in the loop, 8 no epilogue
#endif
#ifndef IGEMM_RING5
#define IGEMM_RING5 3      // stages of the PIPE 5 ring (what-if builds: tools/make_variant_lib.sh ... -DIGEMM_RING5=5)
#endif

// ONEK ("single buffer"): ONE operand buffer instead of two - the arena shrinks from 64 to 37 KB and, with the
// smaller register budget, three to four workgroups share a CU instead of two.  The 1x1 convolutions are HBM-bound
// latency chains (load -> LDS -> a few MFMA -> stage -> store): more of them in flight hides the latency better
// than the double buffer does (measured on the 64->256 layer of ResNet-50: 60 -> 42 us).  With more than one K-step
// the price is a second barrier per step (the tile may only be overwritten once every wave has read it); the
// register prefetch of the next tile still overlaps the MFMAs.
// PIPE = 3: LDS ring of three stages (a fourth stage measured no faster) filled by LDS-DMA (buffer_load ... lds), for the pure-GEMM case (1x1, stride 1) with a
// long reduction and few workgroups (ResNet-50's 16x16 / 8x8 stages, the head's Linear layers).  Those launches
// are ONE dependent chain of K-steps per workgroup; with the register double buffer a step costs a full global
// round trip (~0.9 us measured, 32 steps), with two steps of DMA in flight it costs the MFMAs plus a barrier.
// Lane l of a DMA piece lands at piece_base + 16*l, so each lane FETCHES the (row, chunk) whose swizzled slot
// that is.  Rows past the ragged end re-fetch a valid row (their outputs are masked by the epilogue).
template <int BC, int MODE, int PIPE, bool FBN>
__global__ __launch_bounds__(256, PIPE >= 3 ? (BC == 64 ? 2 : 1) : (PIPE == 1 ? 3 : 2)) void conv_igemm_kernel(ConvArgs a) {
  constexpr bool ONEK = (PIPE == 1);
  constexpr int BP = 128;
  constexpr int WC = BC / 2;   // channels per wave
  constexpr int TM = WC / 16;  // 16-channel tiles per wave
  constexpr int TN = 4;        // 16-pixel tiles per wave
  constexpr int WLD = BC / 32; // weight rows loaded per thread
  constexpr bool CAN_BN = FBN;
  // one arena [weight tiles | pixel tiles]; after the K loop the epilogue re-uses it as output stage
  constexpr int SROW = WC + 8;                    // staged pixel row: WC channels + 16 bytes of padding
  constexpr int NBUF = PIPE == 5 ? (BC == 128 && IGEMM_RING5 > 4 ? 4 : IGEMM_RING5) : PIPE >= 3 ? 3 : (ONEK ? 1 : 2);     // PIPE 4 = PIPE 3 with mma_kstep_upfront
  constexpr int OPER = NBUF * (BC * 64 + BP * 64);
  constexpr int STAGE = 4 * 64 * SROW + (CAN_BN ? 4 * (64 / (WC / 8)) * 2 * WC * 2 : 0);   // + statistics rows (floats)
  constexpr int SMEM = OPER > STAGE ? OPER : STAGE;
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];
  constexpr bool HAS_STATS = (MODE == GATHER_FWD || MODE == GATHER_STEM);   // forward statistics rows: forward convs only
  __shared__ float sRed[2][BC][2];
  bf16_t (*sW)[BC * 64] = reinterpret_cast<bf16_t (*)[BC * 64]>(smem);
  bf16_t (*sX)[BP * 64] = reinterpret_cast<bf16_t (*)[BP * 64]>(smem + NBUF * BC * 64);

  const ConvGeom g = a.g;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wc = wave >> 1, wp = wave & 1;
  const int ncb = (a.Cout + BC - 1) / BC;
  // XCD-aware tile order: hardware workgroup b runs on XCD b % 8 (observed dispatch; a speed matter only), each XCD with its
  // own L2.  Give every XCD a CONTIGUOUS range of logical tiles so that the ncb channel tiles of a pixel block - which read
  // the same 128 input rows - and neighbouring pixel blocks of a 3x3 gather meet in one L2 instead of eight (bijective
  // remap for any grid size, cdna_hip_programming.md T1).
  int bx = blockIdx.x;
  if (XCD_SWIZZLE && a.xcd_swizzle) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bx >> 3);
  }
  const int pb = bx / ncb, cb = bx - pb * ncb;
  const int m0 = pb * BP, c0 = cb * BC;
  const int j = t & 7, row0 = t >> 3;

  // ---- destination pixel grid of this workgroup (a parity class for stride-2 dgrad)
  const int ph = (MODE == GATHER_DGRAD2) ? (int)(blockIdx.y >> 1) : 0;
  const int pw = (MODE == GATHER_DGRAD2) ? (int)(blockIdx.y & 1) : 0;
  const int Hc = (MODE == GATHER_DGRAD2) ? (g.Ho - ph + 1) / 2 : g.Ho;
  const int Wc = (MODE == GATHER_DGRAD2) ? (g.Wo - pw + 1) / 2 : g.Wo;
  const int Mc = g.N * Hc * Wc;
  if (m0 >= Mc) return;   // uniform: whole workgroup leaves before any barrier

  // ---- tap list (uniform)
  const int cpt = (MODE == GATHER_STEM) ? 1 : (g.C >> 6);
  int r0 = 0, s0 = 0, tstep = 1, nr = g.KH, ns = g.KW;
  if (MODE == GATHER_DGRAD2) {
    r0 = (ph + g.pad) & 1; s0 = (pw + g.pad) & 1; tstep = 2;
    nr = r0 < g.KH ? (g.KH - r0 + 1) / 2 : 0;
    ns = s0 < g.KW ? (g.KW - s0 + 1) / 2 : 0;
  }
  const int nk = (MODE == GATHER_STEM) ? 4 : nr * ns * cpt;

  // ---- per-row descriptors: byte offset base (mod 2^32) and a validity bit per tap
  // pure GEMM (1x1, stride 1, no padding: most launches of ResNet-50): pixel m of the destination is row m of the source -
  // no (n, h, w) decomposition (two integer divisions per row), no tap loop.  SQ counters of the 64 -> 256 layer: 563 of the
  // 1055 VALU and ~all 550 SALU instructions a wave executed were this prologue, and the kernel is instruction-issue bound
  // (3 workgroups per CU: VALU busy 78 %, not HBM: 3.4 TB/s).
  const bool pure = (MODE == GATHER_FWD || MODE == GATHER_DGRAD) && g.KH * g.KW == 1 && g.stride == 1 && g.pad == 0 &&
                    g.H == g.Ho && g.W == g.Wo;
  // descriptor of destination pixel m, 16-byte chunk jc of its gathered rows: byte offset base (mod 2^32) + tap validity bits
  auto row_desc = [&](int m, int jc, unsigned& xb, unsigned& xm) {
    unsigned mask = 0;
    long long base = 0;
    if (pure) {
      if (m < Mc) { base = (long long)m * g.C + jc * 8; mask = 1u; }
    } else if (m < Mc) {
      const int hw = Hc * Wc;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int hc = rem / Wc, wcx = rem - hc * Wc;
      if (MODE == GATHER_STEM) {
        const int hb = 2 * hc - 3 + (jc >> 2), wi = 2 * wcx - 4 + 2 * (jc & 3);
        base = ((long long)(n * g.H + hb) * g.W + wi) * 4;
        const bool wok = (unsigned)wi < (unsigned)g.W;
        for (int kt = 0; kt < 4; ++kt) {
          const int r = 2 * kt + (jc >> 2), hi = hb + 2 * kt;
          if (wok && r < 7 && (unsigned)hi < (unsigned)g.H) mask |= 1u << kt;
        }
      } else {
        int hb, wb;
        if (MODE == GATHER_FWD) { hb = hc * g.stride - g.pad; wb = wcx * g.stride - g.pad; }
        else if (MODE == GATHER_DGRAD) { hb = hc + g.pad; wb = wcx + g.pad; }
        else { hb = hc; wb = wcx; }
        base = ((long long)(n * g.H + hb) * g.W + wb) * g.C + jc * 8;
        for (int ri = 0; ri < nr; ++ri)
          for (int si = 0; si < ns; ++si) {
            const int r = r0 + tstep * ri, s = s0 + tstep * si;
            int hi, wi;
            if (MODE == GATHER_FWD) { hi = hb + r * g.dil; wi = wb + s * g.dil; }
            else if (MODE == GATHER_DGRAD) { hi = hb - r; wi = wb - s; }
            else { hi = hb + (ph + g.pad - r) / 2; wi = wb + (pw + g.pad - s) / 2; }
            if ((unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W) mask |= 1u << (ri * ns + si);
          }
      }
    }
    xb = (unsigned)(base * 2);
    xm = mask;
  };
  unsigned xbase[4], xmask[4];
  if (PIPE < 3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) row_desc(m0 + row0 + 32 * i, j, xbase[i], xmask[i]);
  }
  unsigned wbase[WLD];
  bool wok[WLD];
#pragma unroll
  for (int i = 0; i < WLD; ++i) {
    const int c = c0 + row0 + 32 * i;
    wok[i] = c < a.Cout;
    wbase[i] = (unsigned)(((size_t)c * g.Ktot + j * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.src, 0, (unsigned)((size_t)g.N * g.H * g.W * g.C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.wgt, 0, (unsigned)((size_t)a.Cout * g.Ktot * 2), 0x00020000);

  // split-K: this workgroup owns K-steps [kbeg, kend) of nk
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
  const int kslice = ksplit > 1 ? (int)blockIdx.z : 0;
  const int kper = (nk + ksplit - 1) / ksplit;
  const int kbeg = min(nk, kslice * kper), kend = min(nk, kbeg + kper);

  u32x4 xr[4], wr[WLD];
  // Folded BatchNorm-apply + ReLU of the INPUT (ConvArgs::in_bnp; round 6: the conv2 -> conv3 edge of a bottleneck block): x is
  // the RAW output of the producer unit, relu(x * scale + shift) is applied to this thread's 8 channels of every K-step between
  // the load and the LDS store, rounded to bf16 as bn_act_kernel rounds it - the activation tensor is never written or read.
  // Pure 1x1 forward on the register pipelines only (the dispatcher keeps such launches off the DMA rings); a 128-pixel tile
  // lies in one statistics group (the engine folds only when the groups are multiples of 128 pixels).
  const bool bn_in = MODE == GATHER_FWD && PIPE < 3 && a.in_bnp != nullptr;      // uniform
  const float* bn_in_p = bn_in ? a.in_bnp + (size_t)((m0 / (g.H * g.W)) / a.in_npg) * 4 * g.C + j * 8 : nullptr;
  f32x4 isc0 = {0.f, 0.f, 0.f, 0.f}, isc1 = isc0, ish0 = isc0, ish1 = isc0;
  int l_cc = 0, l_ri = 0, l_si = 0, l_ti = 0;   // K-step counters of the NEXT load (uniform, no divisions)
  if (MODE != GATHER_STEM && kbeg > 0) {
    l_ti = kbeg / cpt; l_cc = kbeg - l_ti * cpt;
    l_ri = l_ti / ns; l_si = l_ti - l_ri * ns;
  }
  // uniform decode of the NEXT K-step (called with kt = kbeg, kbeg + 1, ... in order): tap index, source delta, weight column (bytes)
  auto step_decode = [&](int kt, int& ti, int& delta, int& wcol) {
    if (MODE == GATHER_STEM) {
      ti = kt; delta = 2 * kt * g.W * 4 * 2; wcol = kt * 64 * 2;
    } else {
      ti = l_ti;
      const int cc = l_cc << 6;
      const int r = r0 + tstep * l_ri, s = s0 + tstep * l_si;
      if (++l_cc == cpt) {
        l_cc = 0; ++l_ti;
        if (++l_si == ns) { l_si = 0; ++l_ri; }
      }
      int dh, dw;
      if (MODE == GATHER_FWD) { dh = r * g.dil; dw = s * g.dil; }
      else if (MODE == GATHER_DGRAD) { dh = -r; dw = -s; }
      else { dh = (ph + g.pad - r) / 2; dw = (pw + g.pad - s) / 2; }
      delta = ((dh * g.W + dw) * g.C + cc) * 2;
      wcol = ((r * g.KW + s) * g.C + cc) * 2;
    }
  };
  auto load_tiles = [&](int kt) {                // called with kt = kbeg, kbeg + 1, ... in order
    int ti, delta, wcol;
    step_decode(kt, ti, delta, wcol);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = ((xmask[i] >> ti) & 1u) ? xbase[i] + (unsigned)delta : OOB_OFFSET;
      xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
    }
    if (bn_in) {      // pure 1x1: K-step kt is channel block kt
      const float* p = bn_in_p + kt * 64;
      isc0 = *reinterpret_cast<const f32x4*>(p); isc1 = *reinterpret_cast<const f32x4*>(p + 4);
      ish0 = *reinterpret_cast<const f32x4*>(p + g.C); ish1 = *reinterpret_cast<const f32x4*>(p + g.C + 4);
    }
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const unsigned off = wok[i] ? wbase[i] + (unsigned)wcol : OOB_OFFSET;
      wr[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0);
    }
  };
  auto store_tiles = [&](int buf) {
    if (bn_in) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (xmask[i] & 1u) xr[i] = bn_relu_vec(xr[i], isc0, isc1, ish0, ish1);      // rows past M stay zero
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st16(&sX[buf][lds_off(row0 + 32 * i, j)], xr[i]);
#pragma unroll
    for (int i = 0; i < WLD; ++i) st16(&sW[buf][lds_off(row0 + 32 * i, j)], wr[i]);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (PIPE >= 3) {
    constexpr int XQ = 4, WQ = BC / 32, NPW = XQ + WQ;          // DMA pieces (1 KB) per wave and K-step
    const vfs_rsrc_words xrw = vfs_make_rsrc_words(a.src, (unsigned)((size_t)g.N * g.H * g.W * g.C * 2));
    const vfs_rsrc_words wrw = vfs_make_rsrc_words(a.wgt, (unsigned)((size_t)a.Cout * g.Ktot * 2));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // lane l of a piece lands at piece_base + 16 l: it FETCHES the (row, chunk) whose swizzled slot that is.  Pure GEMM: rows
    // past the ragged end re-fetch the tile's first row (their outputs are masked by the epilogue); gathers (round 5: 3x3 /
    // strided layers, the stride-2 dgrad classes): a row's taps that fall on padding - and every tap of a row past M - are sent
    // out of range, where the DMA writes zeros (tools/probe_dma_oob.hip)
    unsigned xvo[XQ], xvm[XQ], wvo[WQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int slot = (wave_u * XQ + q) * 64 + lane, row = slot >> 3, chunk = (slot & 7) ^ ((row >> 1) & 7);
      if (pure) {
        const int m = m0 + row < Mc ? m0 + row : m0;
        xvo[q] = (unsigned)(((size_t)m * g.C) * 2 + chunk * 16);
        xvm[q] = 1u;
      } else {
        row_desc(m0 + row, chunk, xvo[q], xvm[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int slot = (wave_u * WQ + q) * 64 + lane, row = slot >> 3, chunk = (slot & 7) ^ ((row >> 1) & 7);
      const int c = c0 + row < a.Cout ? c0 + row : c0;
      wvo[q] = (unsigned)(((size_t)c * g.Ktot) * 2 + chunk * 16);
    }
    const vfs_lds_t lds_x = vfs_lds_addr(&sX[0][0]), lds_w = vfs_lds_addr(&sW[0][0]);
    auto issue = [&](int kt, int st) {
      if (pure && PIPE == 5) {      // lean issue (4 instructions per piece instead of 10: LDS address as a scalar, m0 not saved)
#pragma unroll
        for (int q = 0; q < XQ; ++q)
          vfs_dma16_async_at(xrw, lds_x + (vfs_lds_t)((st * (BP * 64) + (wave_u * XQ + q) * 512) * 2), xvo[q], (unsigned)kt * 128u);
#pragma unroll
        for (int q = 0; q < WQ; ++q)
          vfs_dma16_async_at(wrw, lds_w + (vfs_lds_t)((st * (BC * 64) + (wave_u * WQ + q) * 512) * 2), wvo[q], (unsigned)kt * 128u);
      } else if (pure) {
#pragma unroll
        for (int q = 0; q < XQ; ++q) vfs_dma16_async(xrw, sX[st] + (wave_u * XQ + q) * 512, xvo[q], (unsigned)kt * 128u);
#pragma unroll
        for (int q = 0; q < WQ; ++q) vfs_dma16_async(wrw, sW[st] + (wave_u * WQ + q) * 512, wvo[q], (unsigned)kt * 128u);
      } else {
        int ti, delta, wcol;
        step_decode(kt, ti, delta, wcol);
#pragma unroll
        for (int q = 0; q < XQ; ++q)
          vfs_dma16_async(xrw, sX[st] + (wave_u * XQ + q) * 512, ((xvm[q] >> ti) & 1u) ? xvo[q] + (unsigned)delta : OOB_OFFSET, 0u);
#pragma unroll
        for (int q = 0; q < WQ; ++q) vfs_dma16_async(wrw, sW[st] + (wave_u * WQ + q) * 512, wvo[q], (unsigned)wcol);
      }
    };
    constexpr int RING = NBUF;
#pragma unroll
    for (int d = 0; d < RING - 1; ++d)
      if (kbeg + d < kend) issue(kbeg + d, d);
    constexpr int QM = WC / 32, QN = 2;       // PIPE 5: 32 x 32 tiles of the wave (WC channels x 64 pixels)
    vfs_f32x16 acc32[PIPE == 5 ? QM : 1][PIPE == 5 ? QN : 1];
    if (PIPE == 5) {
#pragma unroll
      for (int i = 0; i < QM; ++i)
#pragma unroll
        for (int j = 0; j < QN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    }
    int st = 0;
    for (int kt = kbeg; kt < kend; ++kt) {
      // this wave's pieces of step kt have landed: at most the pieces of the later steps already issued stay in flight
      const int ahead = min(RING - 2, kend - 1 - kt);
      if (PIPE == 5 && (IGEMM_WHATIF & 2)) {} else if (ahead >= 4) vfs_dma_wait<4 * NPW>(); else if (ahead == 3) vfs_dma_wait<3 * NPW>(); else if (ahead == 2) vfs_dma_wait<2 * NPW>(); else if (ahead == 1) vfs_dma_wait<NPW>(); else vfs_dma_wait<0>();
      __syncthreads();                       // ... everybody's have, and everybody is done with step kt - 1
      const int nst = st == 0 ? RING - 1 : st - 1;   // the stage step kt - 1 used
      if (!(PIPE == 5 && (IGEMM_WHATIF & 4)) && kt + RING - 1 < kend) issue(kt + RING - 1, nst);
      __builtin_amdgcn_sched_barrier(0);
      if (PIPE == 5 && (IGEMM_WHATIF & 1)) {} else if (PIPE == 5) mma_kstep32<PIPE == 5 ? QM : 1, PIPE == 5 ? QN : 1>(sW[st], sX[st], wc * WC, wp * 64, lane, acc32);
      else if (PIPE == 4) mma_kstep_upfront<TM, TN>(sW[st], sX[st], wc * WC, wp * 64, lane, acc);     // opt-in schedule, own instantiation
      else mma_kstep<TM, TN, false>(sW[st], sX[st], wc * WC, wp * 64, lane, acc);
      __builtin_amdgcn_sched_barrier(0);
      st = st == RING - 1 ? 0 : st + 1;
    }
    __syncthreads();                         // the epilogue re-uses the arena
    if (PIPE == 5 && (IGEMM_WHATIF & 8)) {
      float sum = 0.f;
      for (int i = 0; i < QM; ++i) for (int j = 0; j < QN; ++j) for (int e = 0; e < 16; ++e) sum += acc32[PIPE == 5 ? i : 0][PIPE == 5 ? j : 0][e];
      if (sum == 12345.678f) a.out[0] = (bf16_t)0;
      vfs_dma_wait<0>();
      return;
    }
    if (PIPE == 5) {
      // 32 x 32 accumulators -> the 16 x 16 layout the epilogue is written for, through a wave-private fp32 slab in the idle ring:
      // tile (i, j), register 4 q + r of lane l = channel 32 i + 8 q + 4 (l / 32) + r, pixel 32 j + l % 32 of the wave's tile
      constexpr int TR = WC + 4;
      static_assert(PIPE != 5 || OPER * 2 >= 4 * 64 * TR * 4, "transposition slab does not fit the ring");
      float* slab32 = reinterpret_cast<float*>(smem) + wave * (64 * TR);
      const int l32 = lane & 31, h = lane >> 5;
#pragma unroll
      for (int i = 0; i < QM; ++i)
#pragma unroll
        for (int j = 0; j < QN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(&slab32[(32 * j + l32) * TR + 32 * i + 8 * q + 4 * h]) =
                (f32x4){acc32[PIPE == 5 ? i : 0][PIPE == 5 ? j : 0][4 * q], acc32[PIPE == 5 ? i : 0][PIPE == 5 ? j : 0][4 * q + 1],
                        acc32[PIPE == 5 ? i : 0][PIPE == 5 ? j : 0][4 * q + 2], acc32[PIPE == 5 ? i : 0][PIPE == 5 ? j : 0][4 * q + 3]};
      __builtin_amdgcn_wave_barrier();
      const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = *reinterpret_cast<const f32x4*>(&slab32[(tn * 16 + lr) * TR + tm * 16 + lq * 4]);
      __syncthreads();                       // the epilogue's output stage overlaps the other waves' slabs
    }
  } else if (kend > kbeg) {
    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    for (int kt = kbeg; kt < kend; ++kt) {
      const int cur = ONEK ? 0 : (kt - kbeg) & 1;
      const bool more = kt + 1 < kend;
      if (more) load_tiles(kt + 1);
      __builtin_amdgcn_sched_barrier(0);          // keep the loads ahead of the MFMAs (the scheduler sinks them)
      mma_kstep<TM, TN, false>(sW[cur], sX[cur], wc * WC, wp * 64, lane, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        if (ONEK) __syncthreads();                // single buffer: every wave is done reading the tile
        store_tiles(ONEK ? 0 : cur ^ 1);
      }
      __syncthreads();
    }
  }

  // ---------------- split-K hand-off ----------------
  // every slice parks its accumulators (accumulator layout: 16 coalesced bytes per lane) with device-coherent
  // stores and takes a ticket; the last one to arrive adds the slices up in slice order - the result does not
  // depend on the arrival order - and goes on to the epilogue.  Tickets return to zero for the next launch.
  if (ksplit > 1) {
    __shared__ unsigned s_ticket;
    const int tile = bx;
    unsigned* tickets = reinterpret_cast<unsigned*>(a.ks_ws);
    float* parts = a.ks_ws + KS_TICKETS + (size_t)tile * ksplit * (TM * TN * 1024);
    float* mine = parts + (size_t)kslice * (TM * TN * 1024);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) vfs_store_agent(mine + ((tm * TN + tn) * 256 + t) * 4, acc[tm][tn]);
    vfs_release_workgroup();
    __syncthreads();
    if (t == 0) s_ticket = vfs_ticket_agent(&tickets[tile % KS_TICKETS]);
    __syncthreads();
    if (s_ticket != (unsigned)ksplit - 1u) return;
    if (t == 0) vfs_store_agent(&tickets[tile % KS_TICKETS], 0u);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < ksplit; ++sl) {
      const float* p = parts + (size_t)sl * (TM * TN * 1024);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] += vfs_load_agent4(p + ((tm * TN + tn) * 256 + t) * 4);
    }
  }

  // ---------------- epilogue (vfs_igemm_epi.h) ----------------
  // The K loop ended with a barrier: nobody reads the operand tiles any more, the arena becomes the output stage.
  static_assert(SMEM >= igemm_stage_elems<BC, FBN>(), "stage does not fit the arena");
  igemm_epilogue<BC, MODE, FBN>(a, m0, c0, pb, Mc, ph, pw, Hc, Wc, acc, smem, sRed);
}

// ------------------------------------------------------------------ host launcher
int vfs_option_igemm_bc = 0;     // 64: force the 64-channel tile (A/B knob)
int vfs_option_igemm_xcd = 1;    // XCD-aware tile order (A/B knob)
int vfs_option_igemm_mfma_stats = 1;   // forward statistics rows by MFMA from the staged tile (A/B knob; 0: per-element VALU + DPP)
int vfs_option_igemm_narrow_below = 513;   // 64-channel tiles when the 128-channel tiling has fewer tiles than this (0: never); whole-step A/B: R50 9.45 -> 9.32 ms
int vfs_option_igemm_ring_upfront = 0;  // ring variant: all fragment reads of a K-step before its MFMAs (prepared, not yet measured)
int vfs_option_igemm_ring_fbn = 1;      // the DMA ring also for dgrads with fused BatchNorm-backward statistics (A/B knob)
int vfs_option_igemm_ring_tiles = 512;   // DMA-ring variant for 1x1 problems with at most this many tiles (0: off)
int vfs_option_igemm_ring_gather = 0;      // DMA ring for GATHERED problems (3x3 / strided forward, stride-1 and stride-2 dgrad classes) with at most this many tiles (0: off)
int vfs_option_igemm_ring_mfma32 = 1;   // the pure-GEMM DMA ring on 32x32x16 MFMAs with the lean DMA issue (PIPE 5; 0: PIPE 3, A/B knob)
int vfs_option_igemm_onek = 3;   // single-buffer variant: 0 never, 1 for one-K-step problems (Ktot == 64), 2 every 1x1, 3 always (round 6: default, was 2: the strided / 3x3 gathers too - R18 step 6.85 -> 6.80 ms, R50 7.96 -> 7.93)

template <int BC, int MODE, int PIPE = 0, bool FBN = false>
static int launch_igemm(const ConvArgs& a, hipStream_t stream) {
  int M = a.g.M, classes = 1;
  if (MODE == GATHER_DGRAD2) {   // largest parity class: ceil(Ho/2) x ceil(Wo/2)
    M = a.g.N * ((a.g.Ho + 1) / 2) * ((a.g.Wo + 1) / 2);
    classes = 4;
  }
  int npb = (M + 127) / 128;
  int ncb = (a.Cout + BC - 1) / BC;
  int ks = 1;
  if (a.ksplit > 1) {
    if (MODE == GATHER_DGRAD2 || MODE == GATHER_STEM || !a.ks_ws || npb * ncb > KS_TICKETS)
      return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: split-K needs a stride-1 problem with at most 1024 tiles and a workspace");
    ks = a.ksplit;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BC, MODE, PIPE, FBN>), dim3(npb * ncb, classes, ks), dim3(256), 0, stream, a);
  return vfs_check_launch("conv_igemm");
}

int vfs_conv_igemm_dispatch(const ConvArgs& a_in, int mode, hipStream_t stream) {
  ConvArgs a = a_in;
  a.xcd_swizzle = vfs_option_igemm_xcd;
  a.mfma_stats = vfs_option_igemm_mfma_stats;
  if (a.g.Ktot % 64 != 0 || a.Cout % 8 != 0) return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: K%64 or Cout%8");
  if (vfs_option_halo && a.g.C % 64 == 0 && (size_t)a.g.N * a.g.H * a.g.W * a.g.C * 2 < 0xFFFFFFF0ull &&
      vfs_conv_halo_eligible(a, mode))
    return vfs_conv_halo_dispatch(a, mode, stream);
  if (mode != GATHER_STEM && a.g.C % 64 != 0) return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: C%64");
  if (mode != GATHER_STEM && a.g.KH * a.g.KW > 32) return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: more than 32 taps");
  if ((size_t)a.g.N * a.g.H * a.g.W * a.g.C * 2 >= 0xFFFFFFF0ull || (size_t)a.Cout * a.g.Ktot * 2 >= 0xFFFFFFF0ull)
    return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: tensor >= 4 GiB (split the batch)");
  if (a.g.dil != 1 && mode != GATHER_FWD) return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: dilation is forward-only");
  if (a.bn.partial && (mode != GATHER_DGRAD || a.g.stride != 1))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: fused BatchNorm-backward statistics need a stride-1 dgrad");
  if (a.in_bnp && !(mode == GATHER_FWD && a.g.KH * a.g.KW == 1 && a.g.stride == 1 && a.g.pad == 0 && a.g.H == a.g.Ho && a.g.W == a.g.Wo &&
                    a.in_npg > 0 && ((long long)a.in_npg * a.g.H * a.g.W) % 128 == 0 && a.ksplit <= 1))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: the input BatchNorm folds into the 1x1 / stride-1 forward only, groups of whole 128-pixel tiles");
  if (!a.in_bnp && vfs_conv_skinny_eligible(a, mode)) return vfs_conv_skinny_dispatch(a, stream);   // at most 128 rows (conv_pw.hip)
  if (!a.in_bnp && vfs_conv_pw_eligible(a, mode)) return vfs_conv_pw_dispatch(a, mode, stream);   // persistent kernel (conv_pw.hip)
  // 128-channel tiles, unless that leaves the chip under-filled (deep stages: 16x16 / 8x8 maps, the head): with fewer than
  // igemm_narrow_below tiles the 64-channel tile doubles the workgroups - two latency-bound K chains per CU instead of one
  const long long tiles128 = (long long)((a.g.M + 127) / 128) * ((a.Cout + 127) / 128);
  const bool wide = (a.Cout % 128 == 0) && vfs_option_igemm_bc != 64 && tiles128 >= vfs_option_igemm_narrow_below;
  const bool onek = vfs_option_igemm_onek >= 3 || (vfs_option_igemm_onek == 2 && a.g.KH * a.g.KW == 1) ||
                    (vfs_option_igemm_onek == 1 && a.g.Ktot == 64);
  // DMA ring: pure GEMM (1x1, stride 1, no padding), at least 4 K-steps, at most vfs_option_igemm_ring_tiles tiles
  const int bc = wide ? 128 : 64;
  const long long tiles = (long long)((a.g.M + 127) / 128) * ((a.Cout + bc - 1) / bc);
  const bool ring = vfs_option_igemm_ring_tiles > 0 && a.g.KH * a.g.KW == 1 && a.g.stride == 1 && a.g.pad == 0 &&
                    a.g.Ktot >= 256 && tiles <= vfs_option_igemm_ring_tiles && (!a.bn.partial || vfs_option_igemm_ring_fbn) &&
                    (mode == GATHER_FWD || mode == GATHER_DGRAD) && !a.in_bnp;      // (the DMA ring cannot transform what it moves)
  if (ring && vfs_option_igemm_ring_upfront && !a.bn.partial) {
    if (mode == GATHER_FWD) return wide ? launch_igemm<128, GATHER_FWD, 4>(a, stream) : launch_igemm<64, GATHER_FWD, 4>(a, stream);
    return wide ? launch_igemm<128, GATHER_DGRAD, 4>(a, stream) : launch_igemm<64, GATHER_DGRAD, 4>(a, stream);
  }
  if (ring && vfs_option_igemm_ring_mfma32) {
    if (a.bn.partial) return wide ? launch_igemm<128, GATHER_DGRAD, 5, true>(a, stream) : launch_igemm<64, GATHER_DGRAD, 5, true>(a, stream);
    if (mode == GATHER_FWD) return wide ? launch_igemm<128, GATHER_FWD, 5>(a, stream) : launch_igemm<64, GATHER_FWD, 5>(a, stream);
    return wide ? launch_igemm<128, GATHER_DGRAD, 5>(a, stream) : launch_igemm<64, GATHER_DGRAD, 5>(a, stream);
  }
  if (ring && a.bn.partial)   // the deep-stage dgrads that also emit BatchNorm-backward statistics (round 2: they had been left on the register pipeline)
    return wide ? launch_igemm<128, GATHER_DGRAD, 3, true>(a, stream) : launch_igemm<64, GATHER_DGRAD, 3, true>(a, stream);
  if (ring) {
    if (mode == GATHER_FWD) return wide ? launch_igemm<128, GATHER_FWD, 3>(a, stream) : launch_igemm<64, GATHER_FWD, 3>(a, stream);
    return wide ? launch_igemm<128, GATHER_DGRAD, 3>(a, stream) : launch_igemm<64, GATHER_DGRAD, 3>(a, stream);
  }
  // gathered problems with a long reduction (round 5): 3x3 stride-2 forward / dgrad, strided 1x1, ragged-map 3x3 - one dependent
  // chain of 18-72 K-steps per workgroup, a full memory round trip per step on the register double buffer
  {
    int cls = mode == GATHER_DGRAD && a.g.stride == 2 ? 4 : 1;
    const long long mcls = cls == 4 ? (long long)a.g.N * ((a.g.Ho + 1) / 2) * ((a.g.Wo + 1) / 2) : a.g.M;
    const long long gtiles = ((mcls + 127) / 128) * ((a.Cout + bc - 1) / bc) * cls;
    const bool gring = vfs_option_igemm_ring_gather > 0 && !ring && a.g.KH * a.g.KW > 0 && a.g.Ktot >= 256 && a.ksplit <= 1 &&
                       gtiles <= vfs_option_igemm_ring_gather && !a.bn.partial &&
                       !(a.g.KH * a.g.KW == 1 && a.g.stride == 1 && a.g.pad == 0);
    if (gring && mode == GATHER_FWD) return wide ? launch_igemm<128, GATHER_FWD, 3>(a, stream) : launch_igemm<64, GATHER_FWD, 3>(a, stream);
    if (gring && mode == GATHER_DGRAD && a.g.stride == 1) return wide ? launch_igemm<128, GATHER_DGRAD, 3>(a, stream) : launch_igemm<64, GATHER_DGRAD, 3>(a, stream);
    if (gring && mode == GATHER_DGRAD && a.g.stride == 2) return wide ? launch_igemm<128, GATHER_DGRAD2, 3>(a, stream) : launch_igemm<64, GATHER_DGRAD2, 3>(a, stream);
  }
  switch (mode) {
    case GATHER_FWD:
      if (onek) return wide ? launch_igemm<128, GATHER_FWD, 1>(a, stream) : launch_igemm<64, GATHER_FWD, 1>(a, stream);
      return wide ? launch_igemm<128, GATHER_FWD>(a, stream) : launch_igemm<64, GATHER_FWD>(a, stream);
    case GATHER_DGRAD:
      if (a.g.stride == 1) {
        if (a.bn.partial) {
          if (onek) return wide ? launch_igemm<128, GATHER_DGRAD, 1, true>(a, stream) : launch_igemm<64, GATHER_DGRAD, 1, true>(a, stream);
          return wide ? launch_igemm<128, GATHER_DGRAD, 0, true>(a, stream) : launch_igemm<64, GATHER_DGRAD, 0, true>(a, stream);
        }
        if (onek) return wide ? launch_igemm<128, GATHER_DGRAD, 1>(a, stream) : launch_igemm<64, GATHER_DGRAD, 1>(a, stream);
        return wide ? launch_igemm<128, GATHER_DGRAD>(a, stream) : launch_igemm<64, GATHER_DGRAD>(a, stream);
      }
      if (a.g.stride == 2)
        return wide ? launch_igemm<128, GATHER_DGRAD2>(a, stream) : launch_igemm<64, GATHER_DGRAD2>(a, stream);
      return vfs_set_error(VFS_ERR_SHAPE, "conv_dgrad: stride must be 1 or 2");
    case GATHER_STEM:
      return launch_igemm<64, GATHER_STEM>(a, stream);
  }
  return vfs_set_error(VFS_ERR_ARG, "conv_igemm: bad mode");
}

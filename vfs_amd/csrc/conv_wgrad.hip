// Weight-gradient convolution for gfx950 (autograd wgrad of the reference's conv2d / linear).
//
//   dW[cout][k] = sum_pixels dY[pixel][cout] * G[pixel][k]        (G = implicit im2col of the
//                                                                   forward input, vfs_conv.h)
// GEMM view: rows(i) = k-columns (A operand), cols(j) = cout (B operand), reduction = pixels.
// Both operands live in memory pixel-major (NHWC), i.e. with the REDUCTION index slowest.  The stage
// copies them as they lie (16-byte stores into [pixel][64 channels + pad] LDS tiles) and the MFMA
// fragments are built with ds_read_b64_tr_b16, the LDS transpose read of gfx950 (as in
// conv_wgrad_halo.hip) - the first version transposed through VGPRs with 16 bit-permutes and
// 8-byte LDS stores per lane and ran at a third of the halo kernel's rate.  Split-K over pixel
// ranges; fp32 partials are summed in a fixed order by wgrad_reduce (deterministic), which also
// scatters into the reference's OIHW parameter layout.
#include "vfs_conv.h"
#include "vfs_ops.h"
#include "vfs_wgrad_tail.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
#define WG_RS 72   // LDS row pitch of a [pixel][64 channels] tile: 144 B (see conv_wgrad_halo.hip)
// 8 pixels x 16 channels, transposed: returns the 8 pixel values (MFMA k run) of channel (lane&15)
__device__ __forceinline__ bf16x8 wg_tr_frag(const bf16_t* tile, int lo_off, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + lo_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + hi_off));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

#define WG_OOB 0xFFFFFFF0u   // >= any num_records: the buffer load returns zeros
// LIN: 1x1, stride 1, no padding - the gather is the identity, pixel m of the input is row m of [M][C].  The generic
// path re-derives (n, h, w) -> address for every 16-byte load (~230 VALU instructions per 64-pixel step, more than twice
// the issue time of the step's 32 MFMAs, and with ONE wave per SIMD nothing hides them: SQ counters of the 2048 -> 512
// layer: VALU busy 30 %, MFMA busy 14 %, the rest waits); here a load is `buffer_load voffset` with voffset += one step.
// LIN = 2 (round 5): the same idea for 3x3 / 1x1 convolutions with stride 1 or 2 whose maps tile evenly (H = stride Ho, W = stride
// Wo, Wo | 64, 4 | Wo: the stride-2 layers of the ResNets - 6 launches of ResNet-50 that ran at 108 us each on the generic path).
// Output pixel (R, wo) of GLOBAL row R = n Ho + ho reads input pixel (stride R + r - pad, stride wo + s - pad) - linear in R
// across image boundaries because H = stride Ho - so a lane's byte offset advances by a CONSTANT per 64-pixel step; only the tap's
// validity changes: the column test is fixed per lane, the row test follows ho (one add per step).
template <int BCW, int MODE, int LIN = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int BKC = 128;        // k-columns per workgroup (two forward K-steps)
  // pixel-major tiles: per buffer two k-column halves [64 px][64 + pad] and BCW/64 cout tiles
  constexpr int TILE = 64 * WG_RS, DT = BCW / 64;
  __shared__ __attribute__((aligned(16))) bf16_t sA[2][2 * TILE];
  __shared__ __attribute__((aligned(16))) bf16_t sD[2][DT * TILE];

  const ConvGeom g = a.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nkb = (g.Ktot + BKC - 1) / BKC;
  const int ncb = a.Cout / BCW;
  int b = blockIdx.x;
  // XCD-aware block order (as in conv_igemm.hip): hardware workgroup b runs on XCD b % 8, each XCD behind its own L2.  The
  // nkb * ncb tiles of ONE pixel split read the same dY / X rows; in plain order they sit on eight different XCDs and every L2
  // fetches those rows again (PMC: 139 MB per launch against 75 MB of operands + 16-32 MB of partials).  Give every XCD a
  // contiguous range of logical blocks (bijective remap; which workgroup computes which tile changes, no number does).
  if (a.xcd_swizzle) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int kb = b % nkb; b /= nkb;
  const int cb = b % ncb; b /= ncb;
  const int split = b;
  const int nk = g.Ktot >> 6;

  // loader roles
  const int a_cj = t & 15, a_pg = t >> 4;                       // A tile: 16 chunks x 16 pixel groups
  const int a_kt = kb * 2 + (a_cj >> 3), a_j = a_cj & 7;
  const bool a_ok = a_kt < nk;
  const KStep a_ks = kstep_decode<MODE>(g, a_ok ? a_kt : 0);
  constexpr int DCH = BCW / 8;                                   // dY chunks per pixel
  const int d_cj = t % DCH, d_pg = (t / DCH) & 15;
  const bool d_active = t < DCH * 16;

  const int pix_begin = split * a.pix_per_split;
  const int iters = a.pix_per_split >> 6;

  // The 4 pixels a thread gathers advance by exactly 64 per iteration; their (n, ho, wo) are kept
  // incrementally (no integer division in the loop): 64 = dn*Ho*Wo + dh*Wo + dw.
  const int hw = g.Ho * g.Wo;
  const int dn = 64 / hw, dr = 64 - dn * hw, dh = dr / g.Wo, dw = dr - dh * g.Wo;
  int pn[4], ph[4], pw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = pix_begin + a_pg * 4 + i;
    pn[i] = m / hw;
    const int rem = m - pn[i] * hw;
    ph[i] = rem / g.Wo;
    pw[i] = rem - ph[i] * g.Wo;
  }
  auto pix_coord = [&](int i) {
    PixCoord pc;
    if (pn[i] >= g.N) {
      pc.nH = 0; pc.hb = -(1 << 24); pc.wb = -(1 << 24);
    } else {
      pc.nH = pn[i] * g.H;
      if (MODE == GATHER_STEM) { pc.hb = 2 * ph[i] - 3; pc.wb = 2 * pw[i] - 4; }
      else { pc.hb = ph[i] * g.stride - g.pad; pc.wb = pw[i] * g.stride - g.pad; }
    }
    return pc;
  };
  auto pix_advance = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pw[i] += dw;
      if (pw[i] >= g.Wo) { pw[i] -= g.Wo; ph[i] += 1; }
      ph[i] += dh;
      if (ph[i] >= g.Ho) { ph[i] -= g.Ho; pn[i] += 1; }
      pn[i] += dn;
    }
  };

  u32x4 av[4], dv[4];
  // LIN: per-lane byte offsets of the four rows this lane fetches; rows past M are past num_records (zero fill); lanes
  // without work (k-columns past Ktot, the idle half of the dY loaders) stay at WG_OOB (their step is 0)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, LIN ? (unsigned)((size_t)g.N * g.H * g.W * g.C * 2) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, LIN ? (unsigned)((size_t)g.M * a.Cout * 2) : 0u, 0x00020000);
  unsigned xvo[4], dvo[4];
  unsigned xstep = (LIN && a_ok) ? 64u * (unsigned)g.C * 2u : 0u;
  const unsigned dstep = (LIN && d_active) ? 64u * (unsigned)a.Cout * 2u : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xvo[i] = (LIN && a_ok) ? (unsigned)((((size_t)pix_begin + a_pg * 4 + i) * g.C + a_kt * 64 + a_j * 8) * 2) : WG_OOB;
    dvo[i] = (LIN && d_active) ? (unsigned)((((size_t)pix_begin + d_pg * 4 + i) * a.Cout + cb * BCW + d_cj * 8) * 2) : WG_OOB;
  }
  // LIN = 2: tap (r, s) of this lane's k-columns; the four pixels of a lane lie in one output row (4 | Wo)
  int l2_ho = 0, l2_dho = 0, l2_hb = 0;
  unsigned l2_wv = 0;         // bit i: column of pixel i inside the map
  if (LIN == 2 && a_ok) {
    const int m = pix_begin + a_pg * 4, R = m / g.Wo, wo0 = m - R * g.Wo;
    const int dh = 64 / g.Wo;
    l2_ho = R % g.Ho;
    l2_dho = dh % g.Ho;
    l2_hb = a_ks.r - g.pad;
    xstep = (unsigned)(dh * g.stride * g.W * g.C * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int wi = (wo0 + i) * g.stride + a_ks.s - g.pad;
      if ((unsigned)wi < (unsigned)g.W) l2_wv |= 1u << i;
      const long long pix = ((long long)R * g.stride + a_ks.r - g.pad) * g.W + wi;
      xvo[i] = (unsigned)((pix * g.C + a_ks.c0 + a_j * 8) * 2);
    }
  }
  auto load_tiles = [&](int it) {   // must be called with it = 0, 1, 2, ... in order
    if (LIN == 2) {
      const unsigned hv = (unsigned)(l2_ho * g.stride + l2_hb) < (unsigned)g.H ? l2_wv : 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ((hv >> i) & 1u) ? xvo[i] : WG_OOB, 0, 0); xvo[i] += xstep; }
      l2_ho += l2_dho;
      if (l2_ho >= g.Ho) l2_ho -= g.Ho;
#pragma unroll
      for (int i = 0; i < 4; ++i) { dv[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, dvo[i], 0, 0); dvo[i] += dstep; }
      return;
    }
    if (LIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[i], 0, 0); xvo[i] += xstep; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { dv[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, dvo[i], 0, 0); dvo[i] += dstep; }
      return;
    }
    const int p0 = pix_begin + it * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a_ok) {
        const PixCoord pc = pix_coord(i);
        av[i] = gather16<MODE>(g, a.x, pc, a_ks, a_j);
      } else {
        av[i] = zero16();
      }
    }
    pix_advance();
    if (d_active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = p0 + d_pg * 4 + i;
        dv[i] = (m < g.M) ? ld16(a.dy + (size_t)m * a.Cout + cb * BCW + d_cj * 8) : zero16();
      }
    }
  };
  // Folded BatchNorm-apply + ReLU of the input (WgradArgs::in_bnp; round 6, LIN = 1: the conv3 of a bottleneck block reading the RAW
  // output of conv2): relu(x * scale + shift) on the lane's 8 channels between the load and the LDS store, as bn_act_kernel rounds
  // it.  A 64-pixel step lies in one statistics group (groups are multiples of 128 pixels, splits of 64): the coefficients are
  // reloaded when the group changes.
  const bool bn_in = LIN == 1 && a.in_bnp != nullptr && a_ok;
  const int bn_mpg = bn_in ? a.in_npg * g.H * g.W : 1;
  int bn_gi = -1, st_it = 0;
  f32x4 isc0 = {0.f, 0.f, 0.f, 0.f}, isc1 = isc0, ish0 = isc0, ish1 = isc0;
  auto store_tiles = [&](int buf) {
    if (LIN == 1 && bn_in) {
      const int p0 = pix_begin + st_it * 64;
      if (p0 < g.M) {
        const int gi = p0 / bn_mpg;      // uniform
        if (gi != bn_gi) {
          const float* p = a.in_bnp + (size_t)gi * 4 * g.C + a_kt * 64 + a_j * 8;
          isc0 = *reinterpret_cast<const f32x4*>(p); isc1 = *reinterpret_cast<const f32x4*>(p + 4);
          ish0 = *reinterpret_cast<const f32x4*>(p + g.C); ish1 = *reinterpret_cast<const f32x4*>(p + g.C + 4);
          bn_gi = gi;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (p0 + a_pg * 4 + i < g.M) av[i] = bn_relu_vec(av[i], isc0, isc1, ish0, ish1);      // rows past M stay zero
      }
    }
    ++st_it;
#pragma unroll
    for (int i = 0; i < 4; ++i) st16(&sA[buf][(a_cj >> 3) * TILE + (a_pg * 4 + i) * WG_RS + a_j * 8], av[i]);
    if (d_active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) st16(&sD[buf][(d_cj >> 3) * TILE + (d_pg * 4 + i) * WG_RS + (d_cj & 7) * 8], dv[i]);
    }
  };
  // fragment geometry (conv_wgrad_halo.hip; round 5: v_mfma_f32_32x32x16_bf16 - the 16x16x32 shape sustains 1.47 PFLOP/s chip-wide
  // in this pattern, the 32x32x16 shape 2.43, tools/probe_mfma_rate.hip): a k-step is 16 pixels; MFMA row m = lane % 32 = channel,
  // lane / 32 = k half (k = 8 (lane / 32) + 0..7 <-> pixel 16 st + k).  The four 16-lane groups of a wave are (channels 0-15 |
  // 16-31) x (k half 0 | 1); inside a group one transpose read hands lane i the element of channel i at the four pixel rows the
  // group's lanes point at (lane i points at pixel row i / 4, channel run i % 4).
  const int l16 = lane & 15, h2 = lane >> 5;
  const int pl = 8 * h2 + (l16 >> 2), chq = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
  const int a_base = wm * TILE + pl * WG_RS + chq;                                   // wave = k-column half wm (64 columns: two 32-row tiles)
  const int d_base = (BCW == 128 ? wn * TILE : wn * 32) + pl * WG_RS + chq;          // 64 (or 32) couts of the wave
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  constexpr int QM = 2, QN = BCW / 64;    // 32 x 32 tiles of the wave: k-columns x couts

  f32x16 acc[QM][QN];
#pragma unroll
  for (int tm = 0; tm < QM; ++tm)
#pragma unroll
    for (int tn = 0; tn < QN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < iters;
    if (more) load_tiles(it + 1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      bf16x8 af[QM], bfr[QN];
#pragma unroll
      for (int tm = 0; tm < QM; ++tm)
        af[tm] = wg_tr_frag(sA[cur] + a_base, (16 * st) * WG_RS + tm * 32, (16 * st + 4) * WG_RS + tm * 32);
#pragma unroll
      for (int tn = 0; tn < QN; ++tn)
        bfr[tn] = wg_tr_frag(sD[cur] + d_base, (16 * st) * WG_RS + tn * 32, (16 * st + 4) * WG_RS + tn * 32);
#pragma unroll
      for (int tm = 0; tm < QM; ++tm)
#pragma unroll
        for (int tn = 0; tn < QN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // D[k-column][cout] (32 x 32 tiles): register 4 q + r of lane l = k-column 8 q + 4 (l / 32) + r, cout l % 32 -> 16-byte stores
  if (MODE != GATHER_FWD || !a.tickets) {
#pragma unroll
    for (int tn = 0; tn < QN; ++tn) {
      const int cout = cb * BCW + wn * (BCW / 2) + tn * 32 + (lane & 31);
#pragma unroll
      for (int tm = 0; tm < QM; ++tm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kc = kb * BKC + wm * 64 + tm * 32 + 8 * q + 4 * h2;
          if (kc < g.Ktot)
            *reinterpret_cast<f32x4*>(a.partial + ((size_t)split * a.Cout + cout) * g.Ktot + kc) =
                (f32x4){acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]};
        }
    }
    return;
  }
  // in-launch split-K reduction (vfs_wgrad_tail.h): partials as register images, the last split of this (kb, cb) tile to arrive sums
  // all of them in split order
  constexpr int NI = QN * QM * 4;
  const int tile = cb * nkb + kb, ntiles = nkb * ncb;
  const __amdgpu_buffer_rsrc_t prs = wgt_partial_rsrc(a, ntiles, NI);
#pragma unroll
  for (int tn = 0; tn < QN; ++tn)
#pragma unroll
    for (int tm = 0; tm < QM; ++tm)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (kb * BKC + wm * 64 + tm * 32 < g.Ktot)      // (whole 32-column tiles: Ktot % 64 == 0)
          wgt_store_piece(prs, wgt_piece_off(split, tile, ntiles, NI, (tn * QM + tm) * 4 + q),
                          (f32x4){acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]});
  if (!wgt_last_arriver(a, tile)) return;
  const unsigned sstride = (unsigned)((size_t)ntiles * NI * 4096);
  // four 16-byte pieces (one 32 x 32 accumulator tile) at a time: 16 loads in flight per lane, the registers of the main loop suffice
#pragma unroll 1
  for (int tt = 0; tt < QN * QM; ++tt) {
    const int tn = tt / QM, tm = tt - tn * QM;
    const int cout = cb * BCW + wn * (BCW / 2) + tn * 32 + (lane & 31);
    const bool ok = kb * BKC + wm * 64 + tm * 32 < g.Ktot;
    unsigned off[4];
    f32x4 sum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = ok ? wgt_piece_off(0, tile, ntiles, NI, tt * 4 + q) : WGT_SKIP;
    wgt_sum_splits<4>(a, prs, sstride, off, sum);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ok) wgt_add_grad(a, cout, kb * BKC + wm * 64 + tm * 32 + 8 * q + 4 * h2, sum[q]);
  }
}

// ---------------------------------------------------------------------------------------------
// Round 5: the 1x1 / stride-1 weight gradient (LIN = 1 above) on an LDS-DMA ring.  The register-staged kernel keeps ONE 32 KB
// batch of loads in flight per workgroup and runs one workgroup per CU on the small maps (16 splits x 16 tiles): every 64-pixel
// step waits a memory latency (~31 us for 16 steps where the operands take 9 us at the HBM rate).  Here wave w copies tile w
// of a stage (x: k-column halves 0 / 1, dY: cout halves 0 / 1; 64 pixel rows x 128 bytes each) with eight 1 KB `buffer_load ...
// lds` pieces, WGR_RING - 1 stages ahead, no VGPR staging and one barrier per step.  The rows land UNPADDED (a wave-wide DMA
// piece writes 1 KB contiguously); the 16-byte chunks of a row are XOR-swizzled with the row index (the lane picks WHICH
// global chunk it fetches, csrc/labelprop2.hip) so that the transpose reads of a 16-lane group - four rows x 32 bytes - hit four
// distinct 8-bank sets.  Same MFMA order over the pixels as the staged kernel: bit-identical partials.
#ifndef WGR_RING
#define WGR_RING 4
#endif
__device__ __forceinline__ bf16x8 wg_tr_frag_b(const unsigned char* tile, int lo_off, int hi_off) {      // byte offsets
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + lo_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + hi_off));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
__global__ __launch_bounds__(256, 1) void conv_wgrad_ring_kernel(WgradArgs a) {
  constexpr int RING = WGR_RING, TB = 64 * 128, SB = 4 * TB;      // tile / stage bytes
  __shared__ __attribute__((aligned(1024))) unsigned char sRing[RING][SB];
  const ConvGeom g = a.g;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nkb = g.Ktot >> 7, ncb = a.Cout >> 7;
  int b = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int kb = b % nkb; b /= nkb;
  const int cb = b % ncb; b /= ncb;
  const int split = b;
  const int pix_begin = split * a.pix_per_split;
  const int iters = a.pix_per_split >> 6;

  // ---- the wave's copy job: tile `wave` of every stage
  const bool is_x = wave < 2;
  const unsigned cdim = is_x ? (unsigned)g.C : (unsigned)a.Cout;
  const vfs_rsrc_words rs = is_x ? vfs_make_rsrc_words(a.x, (unsigned)((size_t)g.M * g.C * 2))
                                 : vfs_make_rsrc_words(a.dy, (unsigned)((size_t)g.M * a.Cout * 2));
  const int prow = lane >> 3, cpos = lane & 7;
  const unsigned ch0 = (is_x ? (unsigned)kb : (unsigned)cb) * 128u + (unsigned)(wave & 1) * 64u + (unsigned)((cpos ^ prow) << 3);
  unsigned voff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) voff[j] = (unsigned)((((size_t)pix_begin + 8 * j + prow) * cdim + ch0) * 2);
  const unsigned sstep = 64u * cdim * 2u;
  const vfs_lds_t ring0 = vfs_lds_addr(&sRing[0][0]) + (vfs_lds_t)(wave * TB);
  auto issue = [&](int it, int stg) {
#pragma unroll
    for (int j = 0; j < 8; ++j) vfs_dma16_async_at(rs, ring0 + (vfs_lds_t)(stg * SB + j * 1024), voff[j], (unsigned)it * sstep);
  };

  // ---- fragment geometry (as in the staged kernel); byte offsets inside a swizzled tile for (tm | tn, lo | hi)
  const int l16 = lane & 15, h2 = lane >> 5;
  const int pl = 8 * h2 + (l16 >> 2), chq = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
  int foff[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = pl + 4 * hh, ch = chq + 32 * q;
      foff[q][hh] = row * 128 + ((((ch >> 3) ^ (row & 7))) << 4) + (ch & 7) * 2;
    }
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  f32x16 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

#pragma unroll
  for (int d = 0; d < RING - 1; ++d)
    if (d < iters) issue(d, d);
  int stg = 0;
  for (int it = 0; it < iters; ++it) {
    const int ahead = min(RING - 2, iters - 1 - it);
    if (ahead >= 3) vfs_dma_wait<24>(); else if (ahead == 2) vfs_dma_wait<16>(); else if (ahead == 1) vfs_dma_wait<8>(); else vfs_dma_wait<0>();
    __syncthreads();                       // everybody's pieces of step `it` have landed, everybody is done with step it - 1
    if (it + RING - 1 < iters) issue(it + RING - 1, stg == 0 ? RING - 1 : stg - 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* sa = &sRing[stg][wm * TB];
    const unsigned char* sd = &sRing[stg][(2 + wn) * TB];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) af[tm] = wg_tr_frag_b(sa, st * 2048 + foff[tm][0], st * 2048 + foff[tm][1]);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) bfr[tn] = wg_tr_frag_b(sd, st * 2048 + foff[tn][0], st * 2048 + foff[tn][1]);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    stg = stg == RING - 1 ? 0 : stg + 1;
  }
  if (!a.tickets) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int cout = cb * 128 + wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kc = kb * 128 + wm * 64 + tm * 32 + 8 * q + 4 * h2;
          *reinterpret_cast<f32x4*>(a.partial + ((size_t)split * a.Cout + cout) * g.Ktot + kc) =
              (f32x4){acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]};
        }
    }
    return;
  }
  // in-launch split-K reduction (vfs_wgrad_tail.h)
  const int tile = cb * nkb + kb, ntiles = nkb * ncb;
  const __amdgpu_buffer_rsrc_t prs = wgt_partial_rsrc(a, ntiles, 16);
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        wgt_store_piece(prs, wgt_piece_off(split, tile, ntiles, 16, (tn * 2 + tm) * 4 + q),
                        (f32x4){acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]});
  if (!wgt_last_arriver(a, tile)) return;
  const unsigned sstride = (unsigned)((size_t)ntiles * 16 * 4096);
#pragma unroll 1
  for (int tt = 0; tt < 4; ++tt) {
    const int tn = tt >> 1, tm = tt & 1;
    const int cout = cb * 128 + wn * 64 + tn * 32 + (lane & 31);
    unsigned off[4];
    f32x4 sum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = wgt_piece_off(0, tile, ntiles, 16, tt * 4 + q);
    wgt_sum_splits<4>(a, prs, sstride, off, sum);
#pragma unroll
    for (int q = 0; q < 4; ++q) wgt_add_grad(a, cout, kb * 128 + wm * 64 + tm * 32 + 8 * q + 4 * h2, sum[q]);
  }
}

int vfs_option_wgrad_ring = 1;  // the LDS-DMA ring for 1x1 / stride-1 weight gradients with 128 | C, 128 | Cout (0: the register-staged kernel, A/B knob)
int vfs_option_wgrad_xcd = 1;   // XCD-aware block order of the generic weight-gradient kernel (A/B knob)
int vfs_option_wgrad_lin2 = 1;  // ... and its generalisation to evenly tiled 3x3 / stride-2 problems (A/B knob)
int vfs_option_wgrad_lin = 1;   // the linear-address path for 1x1 / stride-1 problems (A/B knob)

template <int BCW, int MODE, int LIN = 0>
static int launch_wgrad(const WgradArgs& a0, hipStream_t stream) {
  WgradArgs a = a0;
  int nkb = (a.g.Ktot + 127) / 128;
  int ncb = a.Cout / BCW;
  a.xcd_swizzle = vfs_option_wgrad_xcd && nkb * ncb > 1 && nkb * ncb * a.nsplit >= 16;
  hipLaunchKernelGGL((conv_wgrad_kernel<BCW, MODE, LIN>), dim3(nkb * ncb * a.nsplit), dim3(256), 0, stream, a);
  return vfs_check_launch("conv_wgrad");
}

int vfs_conv_wgrad_dispatch(const WgradArgs& a, int mode, hipStream_t stream) {
  if (a.g.Ktot % 64 || a.Cout % 64 || a.pix_per_split % 64 || a.nsplit < 1)
    return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: K%64, Cout%64, pix_per_split%64");
  if ((long long)a.pix_per_split * a.nsplit < a.g.M) return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: splits do not cover M");
  if (a.tickets && (mode != GATHER_FWD || !a.grad || ((a.g.Ktot + 127) / 128) * (a.Cout / 64) > VFS_WGRAD_TICKETS || a.g.C % 4 ||
                    (size_t)a.nsplit * a.Cout * ((a.g.Ktot + 127) / 128 * 128) * 4 >= 0xFFFFFFF0ull))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: the in-launch reduction takes forward-layout problems with a gradient, Cin % 4 == 0, at most 4096 tiles and < 4 GiB of partials");
  const bool wide = (a.Cout % 128 == 0);
  // 1x1 / stride 1 / no padding with 32-bit row offsets (the last split may run past M by less than pix_per_split rows)
  const size_t rows = (size_t)a.g.M + a.pix_per_split, widest = (size_t)(a.g.C > a.Cout ? a.g.C : a.Cout);
  const bool lin = vfs_option_wgrad_lin && mode == GATHER_FWD && a.g.KH * a.g.KW == 1 && a.g.stride == 1 && a.g.pad == 0 &&
                   a.g.H == a.g.Ho && a.g.W == a.g.Wo && rows * widest * 2 < 0xFFF00000ull;
  if (a.in_bnp && !(lin && a.in_npg > 0 && ((long long)a.in_npg * a.g.H * a.g.W) % 128 == 0))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: the input BatchNorm folds into the 1x1 / stride-1 kernel only, groups of whole 128-pixel tiles");
  if (lin && vfs_option_wgrad_ring && a.g.Ktot % 128 == 0 && a.Cout % 128 == 0 && !a.in_bnp) {      // (the DMA ring cannot transform what it moves)
    WgradArgs b = a;
    const int tiles = (a.g.Ktot >> 7) * (a.Cout >> 7);
    b.xcd_swizzle = vfs_option_wgrad_xcd && tiles > 1 && tiles * a.nsplit >= 16;
    hipLaunchKernelGGL(conv_wgrad_ring_kernel, dim3(tiles * a.nsplit), dim3(256), 0, stream, b);
    return vfs_check_launch("conv_wgrad_ring");
  }
  if (lin) return wide ? launch_wgrad<128, GATHER_FWD, 1>(a, stream) : launch_wgrad<64, GATHER_FWD, 1>(a, stream);
  // evenly tiled 3x3 / 1x1 with stride 1 or 2 (LIN = 2)
  const size_t inrows = (size_t)a.g.N * a.g.H * a.g.W + (size_t)a.pix_per_split * a.g.stride * a.g.stride + 4 * (size_t)a.g.W;
  const bool lin2 = vfs_option_wgrad_lin >= 1 && vfs_option_wgrad_lin2 && mode == GATHER_FWD && a.g.KH == a.g.KW && (a.g.KH == 1 || a.g.KH == 3) && a.g.pad == a.g.KH / 2 &&
                    (a.g.stride == 1 || a.g.stride == 2) && a.g.dil == 1 && a.g.H == a.g.stride * a.g.Ho && a.g.W == a.g.stride * a.g.Wo &&
                    a.g.Wo % 4 == 0 && 64 % a.g.Wo == 0 && a.g.C % 64 == 0 && inrows * (size_t)a.g.C * 2 < 0xFFF00000ull && rows * (size_t)a.Cout * 2 < 0xFFF00000ull;
  if (lin2) return wide ? launch_wgrad<128, GATHER_FWD, 2>(a, stream) : launch_wgrad<64, GATHER_FWD, 2>(a, stream);
  if (mode == GATHER_FWD) return wide ? launch_wgrad<128, GATHER_FWD>(a, stream) : launch_wgrad<64, GATHER_FWD>(a, stream);
  if (mode == GATHER_STEM) return launch_wgrad<64, GATHER_STEM>(a, stream);
  return vfs_set_error(VFS_ERR_ARG, "conv_wgrad: bad mode");
}

// ---------------------------------------------------------------------------------------------
// partial[nsplit][Cout][Ktot] -> grad (+=) in the reference's parameter layout (OIHW fp32):
//   FWD  : k = (r*KW + s)*Cin + cin          -> grad[cout][cin][r][s]
//   STEM : k = (r*8 + (s+1))*4 + c           -> grad[cout][c][r][s]   (r<7, 0<=s<7, c<3)
// one pass of a workgroup over `total` elements starting at block `blk` of `nblk`: EL lanes x 4 consecutive elements (one
// 16-byte load each) x SL split-slices; a lane walks its slice's splits four at a time (four independent 16-byte loads in
// flight), the SL slice sums meet in LDS and are added in a fixed order (deterministic).  total % 4 == 0 (Ktot % 4 == 0).
template <int SL>
__device__ __forceinline__ void wgrad_reduce_body(float (*sh)[(256 / SL) * 4], const float* __restrict__ partial, float* __restrict__ grad,
                                                  int nsplit, int Cout, int Ktot, int Cin, int KH, int KW, int stem, int blk, int nblk) {
  constexpr int EL = 256 / SL;
  const size_t total = (size_t)Cout * Ktot;
  const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
  for (size_t base = (size_t)blk * (EL * 4); base < total; base += (size_t)nblk * (EL * 4)) {
    const size_t i = base + (size_t)el * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (i < total) {
      const float* p = partial + i;
      int sp = sl;
      for (; sp + 3 * SL < nsplit; sp += 4 * SL) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)sp * total);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + SL) * total);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 2 * SL) * total);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 3 * SL) * total);
        sum += (v0 + v1) + (v2 + v3);
      }
      for (; sp < nsplit; sp += SL) sum += *reinterpret_cast<const f32x4*>(p + (size_t)sp * total);
    }
    *reinterpret_cast<f32x4*>(&sh[sl][el * 4]) = sum;
    __syncthreads();
    // EL*4 elements, one per thread (EL*4 <= 256 only for SL >= 4; below that a thread takes several)
    for (int e = threadIdx.x; e < EL * 4; e += 256) {
      const size_t ii = base + e;
      if (ii >= total) continue;
      float tot = sh[0][e];
#pragma unroll
      for (int q = 1; q < SL; ++q) tot += sh[q][e];
      const int cout = (int)(ii / Ktot), k = (int)(ii - (size_t)cout * Ktot);
      int cin, r, s2;
      bool ok = true;
      if (stem == 1) {            // direct stem kernel / implicit-GEMM stem: k = (r*8 + (s+1))*4 + c
        cin = k & 3;
        const int si = (k >> 2) & 7;
        r = k >> 5; s2 = si - 1;
        ok = (cin < 3) && (r < 7) && (si >= 1);
      } else {
        const int tap = k / Cin;
        cin = k - tap * Cin;
        r = tap / KW; s2 = tap - r * KW;
      }
      if (ok) {
        const int cin_n = stem ? 3 : Cin;
        grad[(((size_t)cout * cin_n + cin) * KH + r) * KW + s2] += tot;
      }
    }
    __syncthreads();
  }
}

template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ grad,
                                                           int nsplit, int Cout, int Ktot, int Cin, int KH, int KW,
                                                           int stem) {
  __shared__ __attribute__((aligned(16))) float sh[SL][(256 / SL) * 4];
  wgrad_reduce_body<SL>(sh, partial, grad, nsplit, Cout, Ktot, Cin, KH, KW, stem, blockIdx.x, gridDim.x);
}

// Table-driven form: the split-K partials of MANY layers reduced by ONE launch (the per-layer reduction launches were
// 58 of ResNet-50's ~430 launches per step and 0.95 ms of the weight-gradient stream).  Workgroup b serves the
// descriptor d with d.block_start <= b < d.block_start + d.nblocks (binary search, uniform per workgroup).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_table_kernel(const WgradReduceDesc* __restrict__ tab, int n) {
  __shared__ __attribute__((aligned(16))) float sh[SL][(256 / SL) * 4];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WgradReduceDesc d = tab[lo];
  wgrad_reduce_body<SL>(sh, d.partial, d.grad, d.nsplit, d.Cout, d.Ktot, d.Cin, d.KH, d.KW, d.stem, (int)blockIdx.x - d.block_start,
                        d.nblocks);
}

int vfs_wgrad_reduce_table_launch(const WgradReduceDesc* tab, int n, int total_blocks, hipStream_t stream) {
  if (n < 1 || total_blocks < 1) return VFS_OK;
  hipLaunchKernelGGL((wgrad_reduce_table_kernel<8>), dim3(total_blocks), dim3(256), 0, stream, tab, n);
  return vfs_check_launch("wgrad_reduce_table");
}

template <int SL>
static void launch_reduce(const float* partial, float* grad, int nsplit, int Cout, int Ktot, int Cin, int KH, int KW, int stem,
                          hipStream_t stream) {
  const size_t total = (size_t)Cout * Ktot;
  const int per = (256 / SL) * 4;
  int blocks = (int)((total + per - 1) / per);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((wgrad_reduce_kernel<SL>), dim3(blocks), dim3(256), 0, stream, partial, grad, nsplit, Cout, Ktot, Cin,
                     KH, KW, stem);
}

int vfs_wgrad_reduce_launch(const float* partial, float* grad, int nsplit, int Cout, int Ktot, int Cin, int KH, int KW,
                            int stem, hipStream_t stream) {
  if (Ktot % 4) return vfs_set_error(VFS_ERR_SHAPE, "wgrad_reduce: Ktot % 4");
  // as many split-slices per workgroup as there are splits to share (<= 16), the rest of the 256 lanes
  // go to consecutive elements; small tensors with many splits (layer1: 37k elements x 256 splits)
  // still launch hundreds of workgroups
  if (nsplit >= 16) launch_reduce<16>(partial, grad, nsplit, Cout, Ktot, Cin, KH, KW, stem, stream);
  else if (nsplit >= 8) launch_reduce<8>(partial, grad, nsplit, Cout, Ktot, Cin, KH, KW, stem, stream);
  else if (nsplit >= 4) launch_reduce<4>(partial, grad, nsplit, Cout, Ktot, Cin, KH, KW, stem, stream);
  else launch_reduce<1>(partial, grad, nsplit, Cout, Ktot, Cin, KH, KW, stem, stream);
  return vfs_check_launch("wgrad_reduce");
}

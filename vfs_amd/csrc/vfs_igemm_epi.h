// Epilogue of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_pw.hip): the accumulators of one
// 128-pixel x BC-channel workgroup tile -> (+bias) (+residual, optionally gated by a bit-packed ReLU mask) -> bf16 ->
// whole NHWC pixel rows in HBM, plus the statistics rows the BatchNorm kernels consume (forward: sum / sum of squares
// of the stored values; dgrad: the fused BatchNorm-backward sums of vfs_conv.h BnBwdFuse).
//
// Four waves: wave = (wc, wp), wc = wave >> 1 owns WC = BC / 2 channels, wp = wave & 1 owns 64 pixels; acc[tm][tn] is the
// 16x16 MFMA accumulator of channels wc*WC + 16 tm .. and pixels wp*64 + 16 tn ...
//   stage : LDS, 4 * 64 * (WC + 8) bf16 (+ the statistics rows of the FBN variants, igemm_stage_elems); nobody else may
//           touch it between entry and exit (the callers' K loops end with a barrier or use a separate arena)
//   sRed  : LDS float [2][BC][2] (forward statistics only)
//   pb    : index of this tile's statistics row
#pragma once
#include "vfs_conv.h"

template <int BC, bool FBN>
constexpr int igemm_stage_elems() {   // bf16 elements
  return 4 * 64 * (BC / 2 + 8) + (FBN ? 4 * (64 / (BC / 16)) * 2 * (BC / 2) * 2 : 0);
}

// SETTLE (the persistent kernel, conv_pw.hip): every operand load of the epilogue is CONSUMED on every path before the output
// stores are issued.  The loads sit under lane conditions (rows past M, channels past Cout) that differ from the conditions of
// their use, so the compiler must assume one may still be in flight where the registers are next written - in a persistent
// kernel that is the first LDS read of the NEXT tile, and the s_waitcnt vmcnt(0) it puts there also waits for this tile's output
// stores (one in-order counter): the store drain would be exposed once per tile.  Zero-initialised operands + an empty asm that
// names them (a use the compiler cannot drop) end every load's lifetime here, before the stores exist.
template <typename T>
__device__ __forceinline__ void vfs_settle(const T& v) {
#ifndef VFS_EMU
  asm volatile("" ::"v"(v));
#else
  (void)v;
#endif
}

template <int BC, int MODE, bool FBN, bool SETTLE = false>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs& a, const int m0, const int c0, const int pb, const int Mc,
                                               const int ph, const int pw, const int Hc, const int Wc,
                                               f32x4 (&acc)[BC / 32][4], bf16_t* __restrict__ smem, float (*sRed)[BC][2]) {
  constexpr int WC = BC / 2, TM = WC / 16, TN = 4, SROW = WC + 8;
  constexpr bool CAN_BN = FBN;
  constexpr bool HAS_STATS = (MODE == GATHER_FWD || MODE == GATHER_STEM);
  constexpr int SMEM = igemm_stage_elems<BC, FBN>();
  const ConvGeom& g = a.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wc = wave >> 1, wp = wave & 1;
  // (as conv_halo.hip) the accumulator layout gives a lane 8 bytes of one pixel per MFMA tile; each wave
  // transposes its WC x 64 outputs through a private LDS slab and writes 16-byte pieces of whole pixel
  // rows.  The K loop ended with a barrier: nobody reads the operand tiles any more.
  const int lr = lane & 15, lq = lane >> 4;
  const bool do_stats = HAS_STATS && a.stats != nullptr, do_add = a.add != nullptr, do_bias = a.bias != nullptr;
  const bool mstats = do_stats && !do_add && !do_bias && a.mfma_stats;   // uniform
  bf16_t* slab = smem + wave * (64 * SROW);
  auto pixel_dst = [&](int m) -> size_t {         // class-local pixel -> row of the output matrix
    if (MODE != GATHER_DGRAD2) return (size_t)m;
    const int hw = Hc * Wc;
    const int n = m / hw;
    const int rem = m - n * hw;
    const int hc = rem / Wc, wcx = rem - hc * Wc;
    return ((size_t)n * g.Ho + (2 * hc + ph)) * g.Wo + (2 * wcx + pw);
  };
  float s1[TM][4], s2[TM][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[tm][r] = 0.f; s2[tm][r] = 0.f; }
  // fused BatchNorm-backward statistics of the output (vfs_conv.h): the lane's operand rows are requested
  // NOW (one HBM round trip overlapped with the staging below), consumed in the row-store loop
  constexpr int CPR = WC / 8, PPI = 64 / CPR;     // 16-byte chunks per staged row, pixels per store instruction
  const bool do_bn = CAN_BN && a.bn.partial != nullptr;
  BnFuseLane bl;
  if (SETTLE && CAN_BN) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { bl.sc[i] = 0.f; bl.sh[i] = 0.f; bl.mean[i] = 0.f; bl.inv[i] = 0.f; bl.s1[i] = 0.f; bl.s2[i] = 0.f; }
  }
  // Every operand of the epilogue is requested HERE, in straight-line batches, before any of it is used.  (Loads placed next
  // to their use inside the conditional tile loops - the first version - made the compiler wait for each one separately:
  // s_waitcnt vmcnt(0) after each of the 16 residual loads and after each of the 8 statistics-operand pairs of a lane, i.e.
  // ~24 dependent memory round trips per workgroup; SQ counters of the 64 -> 256 dgrad: waves waiting 75 % of their lifetime.)
  u32x4 bxv[CPR];
  unsigned bym[CPR];        // ReLU-mask byte of each piece (from the bit-packed mask, or compressed from the activation on arrival)
  if (SETTLE && CAN_BN) {
#pragma unroll
    for (int i = 0; i < CPR; ++i) { bxv[i] = zero16(); bym[i] = 0u; }
  }
  if (do_bn && c0 + wc * WC + (lane % CPR) * 8 < a.Cout) {
    const int cch = c0 + wc * WC + (lane % CPR) * 8;
    int mmv[CPR];
#pragma unroll
    for (int i = 0; i < CPR; ++i) {
      const int m = m0 + wp * 64 + i * PPI + lane / CPR;
      mmv[i] = m < Mc ? m : m0;
    }
#pragma unroll
    for (int i = 0; i < CPR; ++i) bxv[i] = ld16(a.bn.x + (size_t)mmv[i] * a.Cout + cch);
    if (a.bn.y && a.bn.relu == VFS_MASK_BITS) {
#pragma unroll
      for (int i = 0; i < CPR; ++i) bym[i] = mask8_load(a.bn.y, mmv[i], cch, Mc, a.Cout);
    } else if (a.bn.y) {          // the activation itself as mask operand: reduced to its mask byte (same test, y > 0) in two batches
#pragma unroll
      for (int h = 0; h < CPR; h += 4) {
        u32x4 yv[4];
#pragma unroll
        for (int i = 0; i < 4 && h + i < CPR; ++i) yv[i] = ld16(a.bn.y + (size_t)mmv[h + i] * a.Cout + cch);
#pragma unroll
        for (int i = 0; i < 4 && h + i < CPR; ++i) bym[h + i] = mask8_of(yv[i]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // the residual operand (`add`) in the accumulator layout - 8 bytes per MFMA tile and lane - is requested for TWO pixel
  // tiles at a time (8 loads in flight per lane, two round trips in all; all 16 at once costs 32 more VGPRs and spills);
  // rows / channels past the edge are clamped to a valid address, their values are never used
#pragma unroll
  for (int th = 0; th < TN; th += 2) {
  u32x2 adv[2][TM];
  unsigned long long amwv[2];
  if (SETTLE) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      amwv[t2] = 0ull;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) adv[t2][tm] = (u32x2){0u, 0u};
    }
  }
  if (do_add) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int m = m0 + wp * 64 + (th + t2) * 16 + lr;
      const size_t mdst = pixel_dst(m < Mc ? m : m0);
      const int cw = c0 + wc * WC < a.Cout ? c0 + wc * WC : c0;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int c = c0 + wc * WC + tm * 16 + lq * 4;
        adv[t2][tm] = ld8(a.add + mdst * a.Cout + (c < a.Cout ? c : c0));
      }
      amwv[t2] = a.add_mask ? addmask_word<WC>(a.add_mask, (long long)mdst, cw, a.add_rows, a.Cout) : ~0ull;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (SETTLE) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      vfs_settle(amwv[t2]);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) vfs_settle(adv[t2][tm]);
    }
  }
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int tn = th + t2;
    const int m = m0 + wp * 64 + tn * 16 + lr;
    const bool mok = m < Mc;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int c = c0 + wc * WC + tm * 16 + lq * 4;
      const bool ok = mok && c < a.Cout;
      float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
      if (do_bias && ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += a.bias[c + r];
      }
      if (do_add && ok) {
        const u32x2 ad = adv[t2][tm];
        if (a.add_mask) {
          const unsigned nib = (unsigned)(amwv[t2] >> (tm * 16 + lq * 4));
          v[0] += (nib & 1u) ? bflo(ad.x) : 0.f; v[1] += (nib & 2u) ? bfhi(ad.x) : 0.f;
          v[2] += (nib & 4u) ? bflo(ad.y) : 0.f; v[3] += (nib & 8u) ? bfhi(ad.y) : 0.f;
        } else {
          v[0] += bflo(ad.x); v[1] += bfhi(ad.x); v[2] += bflo(ad.y); v[3] += bfhi(ad.y);
        }
      }
      u32x2 pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      if (mstats && !mok) { pk.x = 0u; pk.y = 0u; }   // rows past M: the DMA-ring variant computes them from a re-fetched valid row
      st8(&slab[(tn * 16 + lr) * SROW + tm * 16 + lq * 4], pk);
      if (do_stats && ok && !mstats) {   // statistics of the STORED (bf16) values
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
  if (mstats) {
    // Statistics rows on the matrix cores (idle here: 16 MFMAs of a one-K-step problem): with Y the wave's staged
    // [64 pixels][WC channels] bf16 tile, sum_p y = Y^T * 1 and sum_p y^2 = diag(Y^T * Y), pixels as the MFMA k index
    // (transposing LDS reads).  The per-element VALU version (unpack, add, fma, then a 16-lane DPP tree per value) was
    // ~400 of the ~770 VALU instructions a wave of the 1x1 forward kernels executed, and those kernels are
    // issue-bound.  Rows past M and channels past Cout hold zeros (no bias / residual on this path).
    const int pl = 4 * lq + (lr >> 2), chq = (lr & 3) * 4;
    const bf16_t* tb = slab + pl * SROW + chq;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3f80;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 f = tile_tr_frag(tb, (32 * ks) * SROW + tm * 16, (32 * ks + 16) * SROW + tm * 16);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, ones, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, a2, 0, 0, 0);
      }
      // output element [row = 4 lq + r][col = lr]: every column of a1 is the sum; the diagonal of a2 is the sum of squares
      if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sRed[wp][wc * WC + tm * 16 + lq * 4 + r][0] = a1[r];
      }
      if ((lr >> 2) == lq) {
        const int r = lr & 3;
        sRed[wp][wc * WC + tm * 16 + lr][1] = r == 0 ? a2[0] : r == 1 ? a2[1] : r == 2 ? a2[2] : a2[3];
      }
    }
  }
  // ---- forward statistics rows.  Plain: written after the output rows (the end of the epilogue).  With COARSE rows (vfs_conv.h)
  // the row leaves BEFORE the output stores - behind them, the release that must precede the ticket would wait for the whole
  // store drain (one in-order counter) - and the ticket's atomic is in flight while the output rows are stored; its result is
  // looked at when they are out.
  const bool coarse = do_stats && a.coarse_log2 > 0 && a.stats_coarse != nullptr;      // uniform
  unsigned my_ticket = 0;
  auto stats_rows_out = [&]() {
    if (!mstats) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x1 = s1[tm][r], x2 = s2[tm][r];
          x1 = row16_sum(x1);   // VALU (DPP) reduction over the 16 pixel lanes
          x2 = row16_sum(x2);
          if (lr == 0) {
            int cl = wc * WC + tm * 16 + lq * 4 + r;
            sRed[wp][cl][0] = x1;
            sRed[wp][cl][1] = x2;
          }
        }
    }
    __syncthreads();
    if (t < BC && c0 + t < a.Cout) {
      float* dst = a.stats + (size_t)pb * 2 * a.Cout;
      const float v0 = sRed[0][t][0] + sRed[1][t][0], v1 = sRed[0][t][1] + sRed[1][t][1];
      if (coarse) { vfs_store_agent(dst + c0 + t, v0); vfs_store_agent(dst + a.Cout + c0 + t, v1); }
      else { dst[c0 + t] = v0; dst[a.Cout + c0 + t] = v1; }
    }
    if (coarse) {
      vfs_release_workgroup();      // this wave's row stores have been performed (nothing else of this wave is in flight yet)
      __syncthreads();
      if (t == 0) my_ticket = vfs_ticket_agent(&a.stats_tickets[(pb >> a.coarse_log2) * ((a.Cout + BC - 1) / BC) + c0 / BC]);
    }
  };
  if (coarse) stats_rows_out();
  {
    const int ch = lane % CPR, c = c0 + wc * WC + ch * 8;
    // statistics group of this wave's 64 rows (blocks never straddle groups); a wave whose rows all lie past M - a single ragged
    // block of a tiny map - must not index a group that does not exist
    if (do_bn && c < a.Cout) bnfuse_init(bl, a.bn, a.Cout, min(m0 + wp * 64, Mc - 1) / a.bn.mpg, c);
    if (SETTLE && CAN_BN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { vfs_settle(bl.sc[i]); vfs_settle(bl.sh[i]); vfs_settle(bl.mean[i]); vfs_settle(bl.inv[i]); }
#pragma unroll
      for (int i = 0; i < CPR; ++i) { vfs_settle(bxv[i]); vfs_settle(bym[i]); }
    }
#pragma unroll
    for (int i = 0; i < CPR; ++i) {
      const int p = i * PPI + lane / CPR;
      const int m = m0 + wp * 64 + p;
      if (m < Mc && c < a.Cout) {
        const size_t o = pixel_dst(m) * a.Cout + c;
        const u32x4 gv = ld16(&slab[p * SROW + ch * 8]);
        st16(a.out + o, gv);
        if (do_bn) bnfuse_accum(bl, a.bn, gv, bxv[i], bym[i]);
      }
    }
    if (do_bn) {   // uniform: one {S1, S2} row per 128-pixel workgroup, summed over pixel groups and the two pixel waves
      float* sB = reinterpret_cast<float*>(smem + 4 * 64 * SROW);
      static_assert(!CAN_BN || SMEM * 2 >= 4 * 64 * SROW * 2 + 4 * PPI * 2 * WC * 4, "statistics do not fit");
      if (c >= a.Cout) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { bl.s1[k] = 0.f; bl.s2[k] = 0.f; }
      }
      bnfuse_spill(bl, sB, wave, lane / CPR, PPI, WC, ch);
      __syncthreads();
      for (int e = t; e < 2 * BC; e += 256) {
        const int st = e / BC, cl = e - st * BC;
        if (c0 + cl >= a.Cout) continue;
        float sum = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const int wv = (cl / WC) * 2 + w2;
          for (int grp = 0; grp < PPI; ++grp) sum += sB[((size_t)(wv * PPI + grp) * 2 + st) * WC + (cl % WC)];
        }
        a.bn.partial[(size_t)pb * 2 * a.Cout + st * a.Cout + c0 + cl] = sum;
      }
    }
  }
  if (do_stats && !coarse) stats_rows_out();
  if (coarse) {
    const int L = a.coarse_log2, grp = pb >> L, row0 = grp << L, npb = (Mc + 127) >> 7;
    const int nrow = min(1 << L, npb - row0);
    vfs_stats_coarsen_finish(a, my_ticket, grp, row0, nrow, nrow, grp * ((a.Cout + BC - 1) / BC) + c0 / BC, c0, min(BC, a.Cout - c0));
  }
}

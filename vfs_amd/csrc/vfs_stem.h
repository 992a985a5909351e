// Stem backward helper shared by bn.hip (materialising apply pass) and stem.hip (fused wgrad):
// gradient wrt the raw stem conv output of one 8-channel vector, rebuilt from the pooled tensors.
#pragma once
#include "vfs_ops.h"

#define STEM_MAX_GROUPS 8
#define STEM_TAB (3 * 64)   // per group: A, B, D of dx = A*ga + B*x + D (as bn_bwd_apply_kernel) for 64 channels

// fill the per-group coefficient table (LDS) once per workgroup; caller syncs afterwards
__device__ __forceinline__ void stem_fill_table(const StemBwdArgs& a, float* tab, float rc) {
  const int ngroups = (a.N + a.npg - 1) / a.npg;
  for (int i = threadIdx.x; i < ngroups * 64; i += blockDim.x) {
    const int gi = i >> 6, c = i & 63;
    const float* bp = a.bnp + (size_t)gi * 4 * 64 + c;
    const float A = bp[0], mean = bp[2 * 64], inv = bp[3 * 64];
    const float m1 = (float)a.sums[((size_t)gi * 2) * 64 + c] * rc, m2 = (float)a.sums[((size_t)gi * 2 + 1) * 64 + c] * rc;
    float* tb = tab + gi * STEM_TAB + c;
    tb[0] = A;
    tb[64] = -A * inv * m2;
    tb[128] = A * (mean * inv * m2 - m1);
  }
}

// d[8] = A*bf16(ga) + (B*x + D) for eight channels: ga as a materialising path would have stored it (bf16), then the BN backward
// formula - two explicit fused multiply-adds, so that every kernel that rebuilds dY (gather: stem_dx_vec; scatter:
// stem_wgrad_scatter_kernel, csrc/stem.hip) produces the same bits whatever the compiler would have contracted
__device__ __forceinline__ void stem_dx_combine(const float* tb, const float* g, u32x4 xv, float* d) {
  float x[8];
  unpack8(xv, x);
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = __builtin_fmaf(tb[i], round_bf(g[i]), __builtin_fmaf(tb[64 + i], x[i], tb[128 + i]));
}

// d[8] = A*bf16(ga) + B*x + D  (= scale * (ga - m1 - xhat*m2)) for pixel (n,h,w), channels c..c+7 (C = 64).
// ga = sum over the <=4 pooling windows whose argmax is (h,w) of gp (windows whose pooled activation is not positive carry
// code 0xFF and match nothing); the window loads are issued up front.
// Round 6: a pixel in an EVEN row lies in ONE window row (h>>1 == (h+1)>>1), one in an even column in one window column - only
// (odd, odd) pixels have four windows, the average is 2.25.  The windows that do not exist are skipped by branches (they used to be
// loaded from clamped addresses and masked): callers that give a wave pixels of one row and one column parity (stem_wgrad_fused)
// save the loads and the vector work of 44 % of the windows; the sums are the same (the skipped terms were never added).
__device__ __forceinline__ void stem_dx_vec(const StemBwdArgs& a, const float* tab, int n, int h, int w, int c, float* d) {
  const int gi = n / a.npg;
  const int hp[2] = {h >> 1, (h + 1) >> 1}, wp[2] = {w >> 1, (w + 1) >> 1};
  u32x2 id[4];
  u32x4 gv[4];
  unsigned code[4];
  bool ok[4];
  const unsigned prow = (unsigned)a.Wp * 64u;
  const unsigned o00 = (((unsigned)n * (unsigned)a.Hp + (unsigned)hp[0]) * (unsigned)a.Wp + (unsigned)wp[0]) * 64u + (unsigned)c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int hh = hp[k >> 1], ww = wp[k & 1];
    ok[k] = hh < a.Hp && ww < a.Wp && !((k >> 1) && hp[1] == hp[0]) && !((k & 1) && wp[1] == wp[0]);
    code[k] = (unsigned)((h - (2 * hh - 1)) * 3 + (w - (2 * ww - 1)));
    if (ok[k]) {      // (id / gv of a window that does not exist stay unset: they are only read under the same predicate)
      // 32-bit element offsets (the launchers refuse tensors of 2^31 elements or more): the 64-bit index arithmetic of the first
      // version was a third of the vector instructions that were left after the window skip
      const unsigned o = o00 + ((k >> 1) ? prow : 0u) + ((k & 1) ? 64u : 0u);
      id[k] = ld8(a.idx + o);      // argmax code, 0xFF where the pooled activation is not positive (bn_relu_maxpool_kernel): the
      gv[k] = ld16(a.gp + o);      // ReLU mask travels in the code, yp is not read
    }
  }
  const u32x4 xv = ld16(a.x + ((((unsigned)n * (unsigned)a.H + (unsigned)h) * (unsigned)a.W + (unsigned)w) * 64u + (unsigned)c));
  float g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (ok[k]) {
      float gp[8];
      unpack8(gv[k], gp);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned b = ((i < 4 ? id[k].x : id[k].y) >> (8 * (i & 3))) & 0xffu;
        if (b == code[k]) g[i] += gp[i];
      }
    }
  }
  stem_dx_combine(tab + gi * STEM_TAB + c, g, xv, d);
}

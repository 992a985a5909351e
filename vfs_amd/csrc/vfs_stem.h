// Stem backward helper shared by bn.hip (materialising apply pass) and stem.hip (fused wgrad):
// gradient wrt the raw stem conv output of one 8-channel vector, rebuilt from the pooled tensors.
#pragma once
#include "vfs_ops.h"

__device__ __forceinline__ void stem_ld8f(const float* p, float* f) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
  f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

// d[8] = scale * (bf16(ga) - m1 - xhat*m2) for pixel (n,h,w), channels c..c+7
__device__ __forceinline__ void stem_dx_vec(const StemBwdArgs& a, int n, int h, int w, int c, float rc, float* d) {
  const int gi = n / a.npg;
  float g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = 0.f;
  for (int hp = h >> 1; hp <= (h + 1) >> 1; ++hp) {
    if (hp >= a.Hp) continue;
    const int dy = h - (2 * hp - 1);
    for (int wp = w >> 1; wp <= (w + 1) >> 1; ++wp) {
      if (wp >= a.Wp) continue;
      const int dx = w - (2 * wp - 1);
      const unsigned code = (unsigned)(dy * 3 + dx);
      const size_t o = ((((size_t)n * a.Hp + hp) * a.Wp) + wp) * a.C + c;
      const u32x2 id = ld8(a.idx + o);
      float gp[8], yp[8];
      unpack8(ld16(a.gp + o), gp);
      unpack8(ld16(a.yp + o), yp);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned b = ((i < 4 ? id.x : id.y) >> (8 * (i & 3))) & 0xffu;
        if (b == code && yp[i] > 0.f) g[i] += gp[i];
      }
    }
  }
  const size_t o = (((size_t)n * a.H + h) * a.W + w) * a.C + c;
  float x[8], sc[8], mean[8], inv[8];
  unpack8(ld16(a.x + o), x);
  const float* bp = a.bnp + (size_t)gi * 4 * a.C;
  stem_ld8f(bp + c, sc);
  stem_ld8f(bp + 2 * a.C + c, mean);
  stem_ld8f(bp + 3 * a.C + c, inv);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float m1 = (float)a.sums[((size_t)gi * 2) * a.C + c + i] * rc;
    const float m2 = (float)a.sums[((size_t)gi * 2 + 1) * a.C + c + i] * rc;
    // ga as a materialising path would have stored it (bf16), then the BN backward formula
    d[i] = sc[i] * (round_bf(g[i]) - m1 - ((x[i] - mean[i]) * inv[i]) * m2);
  }
}

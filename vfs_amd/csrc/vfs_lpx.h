// Shared pieces of the fp32 ("exact") label propagation kernels (exact_f32.hip: the dense kernel and the merge / softmax kernel;
// labelprop2.hip: the two-pass kernels): the softmax exponential, the total order of the top-k and its sorted insertion.
// Arithmetic contract: every operation here is ONE correctly rounded fp32 operation (files including this switch contraction off).
#pragma once
#include "vfs_ops.h"

__device__ __forceinline__ f32x4 lpx_ldf4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// exp(x), x <= 0 (oracle: xo_exp): n = rint(x log2 e), r = x - n ln2 (two fmaf), degree-6 Horner, 2^n scaling
__device__ __forceinline__ float vexp(float x) {
  if (x < -87.0f) return 0.0f;
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  p = __builtin_fmaf(p, r * r, r);
  p = p + 1.0f;
  return ldexpf(p, (int)n);
}

#define LPX_TOPK 10
#define LPX_QCAP 8
typedef __attribute__((ext_vector_type(2))) unsigned int vfs_u32x2q;
#define LPX_NONE 0x7fffffff
// total order of the top-k: larger score first, equal scores: smaller candidate id first
__device__ __forceinline__ bool lpx_better(float s, int id, float ts, int tid) { return s > ts || (s == ts && id < tid); }
__device__ __forceinline__ void lpx_insert(float (&tv)[LPX_TOPK], int (&ti)[LPX_TOPK], float s, int id) {
  if (lpx_better(s, id, tv[LPX_TOPK - 1], ti[LPX_TOPK - 1])) { tv[LPX_TOPK - 1] = s; ti[LPX_TOPK - 1] = id; }
#pragma unroll
  for (int j = LPX_TOPK - 1; j > 0; --j) {
    const bool sw = lpx_better(tv[j], ti[j], tv[j - 1], ti[j - 1]);
    const float a = tv[j - 1], b = tv[j];
    const int ia = ti[j - 1], ib = ti[j];
    tv[j - 1] = sw ? b : a; tv[j] = sw ? a : b;
    ti[j - 1] = sw ? ib : ia; ti[j] = sw ? ia : ib;
  }
}


// CosineSimLoss beyond the shipped configs' [N,C] case (mmaction/models/losses/sim_loss.py:42-63): spatial inputs
// [B][C][S] (the reference's NC* layout, flattened), optional L2 normalisation over C, the PAIRWISE affinity
//     prod[b][i][j] = sum_c a^[b][c][i] * l^[b][c][j]            (einsum 'bci,bcj->bij', sim_loss.py:51)
// optionally multiplied by a mask [B][Sa][Sl], mean over (i, j), loss = 2 - 2 * mean (or -mean), and its backward.
// The affinity is a dense contraction: it runs on the matrix cores in fp32 (v_mfma_f32_32x32x2_f32 - the reference is
// fp32; these tensors are small next to the backbone, so no bf16 rounding is spent here).  Non-pairwise spatial inputs
// (sum over C, mean over S) are the diagonal of the same product (`pairwise = 0`).
//
// fp32 throughout; the normalisation is applied as a row / column scale of the accumulated tile
// (prod = acc * inva[i] * invl[j]), reductions run in a fixed order (deterministic).
#include "vfs_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// inv[b][s] = 1 / max(||x[b][:][s]||_2, 1e-12)      (F.normalize(p=2, dim=1, eps=1e-12), sim_loss.py:44-45)
__global__ __launch_bounds__(256) void simloss_colnorm_kernel(const float* __restrict__ x, float* __restrict__ inv, int C, int S) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x * 64 + lane;
  const float* xb = x + (size_t)b * C * S;
  float acc = 0.f;
  if (s < S)
    for (int c = wv; c < C; c += 4) {
      const float v = xb[(size_t)c * S + s];
      acc = fmaf(v, v, acc);
    }
  red[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && s < S) {
    const float n = sqrtf(((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]);
    inv[(size_t)b * S + s] = 1.0f / fmaxf(n, 1e-12f);
  }
}

struct SimFwdArgs {
  const float *a, *l, *inva, *invl, *mask;
  float* partial;      // [B][tiles_i * tiles_j]
  int C, Sa, Sl, pairwise;
};

// one workgroup = one 32 x 32 tile of prod[b]; the four waves split C (interleaved channel pairs), scale / mask their partial
// tile (linear) and reduce it to one number each; partial[b][tile] = the four numbers added in wave order
__global__ __launch_bounds__(256) void simloss_pair_fwd_kernel(SimFwdArgs p) {
  __shared__ float red[4];
  // diagonal mode (pairwise = 0): the grid holds the DIAGONAL tiles only (gridDim.x == 1, j0 = i0)
  const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = p.pairwise ? blockIdx.x * 32 : i0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, kk = lane >> 5;
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  float tot = 0.f;
  {
    const float* ab = p.a + (size_t)b * p.C * p.Sa;
    const float* lb = p.l + (size_t)b * p.C * p.Sl;
    const bool va = i0 + li < p.Sa, vl = j0 + li < p.Sl;
    f32x16 acc = {0};
    for (int c = 2 * wv; c < p.C; c += 8) {
      const int cc = c + kk;
      const bool vc = cc < p.C;
      const float av = (va && vc) ? ab[(size_t)cc * p.Sa + i0 + li] : 0.f;
      const float lv = (vl && vc) ? lb[(size_t)cc * p.Sl + j0 + li] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, lv, acc, 0, 0, 0);
    }
    const int col = j0 + li;
    const float cl = (col < p.Sl) ? (p.invl ? p.invl[(size_t)b * p.Sl + col] : 1.f) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (row >= p.Sa || col >= p.Sl) continue;
      if (!p.pairwise && row != col) continue;
      float v = acc[r] * (p.inva ? p.inva[(size_t)b * p.Sa + row] : 1.f) * cl;
      if (p.mask) v *= p.mask[((size_t)b * p.Sa + row) * p.Sl + col];
      tot += v;
    }
    for (int off = 32; off; off >>= 1) tot += __shfl_xor(tot, off);
  }
  if (lane == 0) red[wv] = tot;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[(size_t)b * gridDim.x * gridDim.y + tile] = ((red[0] + red[1]) + red[2]) + red[3];
}

// loss[b] = weight * (negative ? -mean : 2 - 2 * mean), mean = sum(partial[b][:]) / count      (sim_loss.py:59-62, base.py:37)
__global__ __launch_bounds__(64) void simloss_finish_kernel(const float* __restrict__ partial, float* __restrict__ loss, int B, int tiles,
                                                            double count, int negative, float weight) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double acc = 0.0;
  for (int t = 0; t < tiles; ++t) acc += (double)partial[(size_t)b * tiles + t];
  const double m = acc / count;
  loss[b] = weight * (float)(negative ? -m : 2.0 - 2.0 * m);
}

struct SimBwdArgs {
  const float *other, *invo, *mask, *gloss;
  float* d;          // [B][C][Sself]: gradient wrt the NORMALISED self operand
  int C, Sself, Sother, pairwise, mask_transposed, negative;
  float weight;
  double count;
};

// d[b][c][i] = coef[b] * sum_j other^[b][c][j] * Mk(i, j),   Mk = mask[b][i][j] (or [b][j][i] when this side is the einsum's j
// operand), 1 without a mask; coef[b] = gloss[b] * weight * (negative ? -1 : -2) / count.  One workgroup = 32 channels x 32
// positions; the four waves split the contraction index j, partial tiles are added through LDS in wave order.
__global__ __launch_bounds__(256) void simloss_pair_bwd_kernel(SimBwdArgs p) {
  __shared__ float sacc[4][16][64];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, kk = lane >> 5;
  const float* ob = p.other + (size_t)b * p.C * p.Sother;
  const float coef = p.gloss[b] * p.weight * (p.negative ? -1.f : -2.f) / (float)p.count;
  if (!p.pairwise) {      // diagonal: d[c][i] = coef * other^[c][i]
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
      const int c = c0 + (e >> 5), i = i0 + (e & 31);
      if (c < p.C && i < p.Sself)
        p.d[((size_t)b * p.C + c) * p.Sself + i] = coef * ob[(size_t)c * p.Sother + i] * (p.invo ? p.invo[(size_t)b * p.Sother + i] : 1.f);
    }
    return;
  }
  f32x16 acc = {0};
  const bool vc = c0 + li < p.C, vi = i0 + li < p.Sself;
  for (int j = 2 * wv; j < p.Sother; j += 8) {
    const int jj = j + kk;
    const bool vj = jj < p.Sother;
    float ov = 0.f, mv = 0.f;
    if (vj && vc) ov = ob[(size_t)(c0 + li) * p.Sother + jj] * (p.invo ? p.invo[(size_t)b * p.Sother + jj] : 1.f);
    if (vj && vi) {
      if (!p.mask) mv = 1.f;
      else mv = p.mask_transposed ? p.mask[((size_t)b * p.Sother + jj) * p.Sself + i0 + li] : p.mask[((size_t)b * p.Sself + i0 + li) * p.Sother + jj];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ov, mv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) sacc[wv][r][lane] = acc[r];
  __syncthreads();
  if (wv == 0) {
    const int col = i0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = c0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (row < p.C && col < p.Sself)
        p.d[((size_t)b * p.C + row) * p.Sself + col] = coef * (((sacc[0][r][lane] + sacc[1][r][lane]) + sacc[2][r][lane]) + sacc[3][r][lane]);
    }
  }
}

// backward of x^ = x * inv (inv = 1 / max(||x||, eps)): dx = (d - x^ * <x^, d>) * inv per column; inv == nullptr: dx = d
__global__ __launch_bounds__(256) void simloss_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ inv,
                                                               const float* __restrict__ d, float* __restrict__ dx, int C, int S) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x * 64 + lane;
  const size_t base = (size_t)b * C * S;
  const float iv = (inv && s < S) ? inv[(size_t)b * S + s] : 1.f;
  float acc = 0.f;
  if (inv && s < S)
    for (int c = wv; c < C; c += 4) acc = fmaf(x[base + (size_t)c * S + s] * iv, d[base + (size_t)c * S + s], acc);
  red[wv][lane] = acc;
  __syncthreads();
  if (s >= S) return;
  const float dot = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  for (int c = wv; c < C; c += 4) {
    const size_t o = base + (size_t)c * S + s;
    dx[o] = inv ? (d[o] - x[o] * iv * dot) * iv : d[o];
  }
}

int vfs_simloss_colnorm_launch(const float* x, float* inv, int B, int C, int S, hipStream_t s) {
  hipLaunchKernelGGL(simloss_colnorm_kernel, dim3((S + 63) / 64, B), dim3(256), 0, s, x, inv, C, S);
  return vfs_check_launch("simloss_colnorm");
}

int vfs_simloss_fwd_launch(const float* a, const float* l, const float* inva, const float* invl, const float* mask, float* partial,
                           float* loss, int B, int C, int Sa, int Sl, int pairwise, int negative, float weight, hipStream_t s) {
  SimFwdArgs p;
  p.a = a; p.l = l; p.inva = inva; p.invl = invl; p.mask = mask; p.partial = partial; p.C = C; p.Sa = Sa; p.Sl = Sl; p.pairwise = pairwise;
  const int ti = (Sa + 31) / 32, tj = (Sl + 31) / 32;
  // diagonal mode: only the ti diagonal tiles exist (the full ti x tj grid was ~98x the work at S = 56 * 56, and the finish kernel
  // walked ti * tj partials per sample)
  const int gx = pairwise ? tj : 1;
  hipLaunchKernelGGL(simloss_pair_fwd_kernel, dim3(gx, ti, B), dim3(256), 0, s, p);
  int rc = vfs_check_launch("simloss_pair_fwd");
  if (rc) return rc;
  const double count = pairwise ? (double)Sa * Sl : (double)Sa;
  hipLaunchKernelGGL(simloss_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const float*)partial, loss, B, ti * gx, count, negative, weight);
  return vfs_check_launch("simloss_finish");
}

int vfs_simloss_bwd_launch(const float* other, const float* invo, const float* mask, int mask_transposed, const float* gloss, float* d,
                           int B, int C, int Sself, int Sother, int pairwise, int negative, float weight, hipStream_t s) {
  SimBwdArgs p;
  p.other = other; p.invo = invo; p.mask = mask; p.gloss = gloss; p.d = d; p.C = C; p.Sself = Sself; p.Sother = Sother;
  p.pairwise = pairwise; p.mask_transposed = mask_transposed; p.negative = negative; p.weight = weight;
  p.count = pairwise ? (double)Sself * Sother : (double)Sself;
  hipLaunchKernelGGL(simloss_pair_bwd_kernel, dim3((Sself + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, p);
  return vfs_check_launch("simloss_pair_bwd");
}

int vfs_simloss_norm_bwd_launch(const float* x, const float* inv, const float* d, float* dx, int B, int C, int S, hipStream_t s) {
  hipLaunchKernelGGL(simloss_norm_bwd_kernel, dim3((S + 63) / 64, B), dim3(256), 0, s, x, inv, d, dx, C, S);
  return vfs_check_launch("simloss_norm_bwd");
}

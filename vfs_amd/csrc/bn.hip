// BatchNorm (training mode, per-group batch statistics), activation, residual and the stem
// max-pool for NHWC bf16 tensors on gfx950 -- all HBM-bound, 16 bytes (8 channels) per lane.
//
// Replaces torch batch_norm / SyncBatchNorm + relu + residual add + max_pool2d as the reference
// composes them (mmcv ConvModule conv->norm->act; resnet.py:102-111,221-230,435; configs/*:9,15)
// and their autograd backward.  "Groups" are independent BN batches inside one tensor: the two
// augmented views of forward_train (sim_siam_base_tracker.py:69-70) are separate backbone calls
// in the reference, so their statistics must not be merged.
//
// bnp layout (per layer): float[G][4][C] = {scale = gamma*invstd, shift = beta - mean*scale,
//                                           mean, invstd}
#include "vfs_ops.h"
#include "vfs_p2p.h"
#include "vfs_stem.h"

// SyncBN (configs/r*_*.py:9,15): the sums of a BatchNorm layer are needed over ALL ranks before the apply pass.  When the P2P
// window exchange is up (vfs_amd/p2p.py), the reduction kernels below finish with it: the workgroup that completes LAST of
// `nblocks` (a ticket in state[2]) pushes sums[0..n) into every rank's window, waits for the peers' stamps and writes back the
// sum over the ranks (vfs_p2p.h) - "local sums" and "sums over the ranks" are one launch, and no collective-library call.
// Every writer of `sums` uses agent-scope stores and calls this with ALL its threads after the last store.
__device__ __forceinline__ void bn_xchg_tail(const P2PTail& x, double* sums, int n, unsigned nblocks) {
  __shared__ unsigned s_last;
  __shared__ int s_failed;
  vfs_release_workgroup();
  __syncthreads();
  if (threadIdx.x == 0) s_last = vfs_ticket_agent(reinterpret_cast<unsigned*>(x.state + 2));
  __syncthreads();
  if (s_last != nblocks - 1u) return;
  p2p_exchange_body(sums, n, x.peers, x.rank, x.world, x.state, 3, x.spin_limit, &s_failed);
  if (threadIdx.x == 0) vfs_store_agent(reinterpret_cast<unsigned*>(x.state + 2), 0u);
}
__device__ __forceinline__ void bn_store_sum(const P2PTail& x, double* p, double v) {
  if (x.peers) vfs_store_agent(p, v); else *p = v;
}

// ------------------------------------------------------------------------------------------
// partial[nblk][2][C] (fp32, one per producer block) -> sums[G][2][C] (fp64), fixed order.
// group gi owns rows [gi*bpg, (gi+1)*bpg).  Large row counts are reduced in two deterministic
// stages (row chunks in parallel -> fp64 chunk sums -> final) so that e.g. the 16384 partials
// per view of the stem do not serialise on four workgroups.
template <typename TIN>
__global__ __launch_bounds__(256) void bn_reduce_rows_kernel(const TIN* __restrict__ in, double* __restrict__ out, int bpg,
                                                             int C, int rpc, int nchunks, P2PTail x) {
  __shared__ double sh[8][2][32];
  const int t = threadIdx.x, cl = t & 31, sl = t >> 5;
  const int c = blockIdx.x * 32 + cl, gi = blockIdx.y, ch = blockIdx.z;
  const int r0 = ch * rpc, r1 = min(bpg, r0 + rpc);
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    const TIN* p = in + (size_t)gi * bpg * 2 * C;
    for (int b = r0 + sl; b < r1; b += 8) {
      a0 += (double)p[(size_t)b * 2 * C + c];
      a1 += (double)p[(size_t)b * 2 * C + C + c];
    }
  }
  sh[sl][0][cl] = a0;
  sh[sl][1][cl] = a1;
  __syncthreads();
  if (sl == 0 && c < C) {
    double r0s = 0.0, r1s = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) { r0s += sh[s][0][cl]; r1s += sh[s][1][cl]; }
    double* o = out + ((size_t)gi * nchunks + ch) * 2 * C;
    bn_store_sum(x, &o[c], r0s);
    bn_store_sum(x, &o[C + c], r1s);
  }
  if (x.peers) bn_xchg_tail(x, out, (int)(gridDim.y * 2 * C), gridDim.x * gridDim.y);      // (only launched with nchunks == 1: out = sums)
}

// running = (1 - momentum) * running + momentum * statistic (unbiased variance).  A NaN statistic - the poison a failed SyncBN
// window exchange writes into its sums (vfs_p2p.h) - must not reach the running statistics: they outlive the step (checkpoints).
// (written with explicit fmaf: the compiler contracted `a * b + c * d` differently in different kernels - the fused and the
// two-launch finalisation must produce the same bits, tests/test_emu_bn.py::test_bn_chunked_single_launch_reduction)
// (bn_running_update: vfs_ops.h - shared with the fused Linear + BatchNorm1d launch of csrc/conv_pw.hip)

// sums[G][2][C] (sum x, sum x^2; already all-reduced across ranks for SyncBN) + count ->
// bnp, and the running statistics updated group after group (each group is one BN call of the
// reference: running = (1-momentum)*running + momentum*stat, unbiased variance).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ bnp,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          int G, int C, double count, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  for (int gi = 0; gi < G; ++gi) {
    const double mean = sums[((size_t)gi * 2) * C + c] / count;
    double var = sums[((size_t)gi * 2 + 1) * C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * invstd;
    float* o = bnp + (size_t)gi * 4 * C;
    o[c] = scale;
    o[C + c] = beta[c] - (float)mean * scale;
    o[2 * C + c] = (float)mean;
    o[3 * C + c] = invstd;
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    bn_running_update(rm, rv, momentum, mean, unbiased);
  }
  if (running_mean) running_mean[c] = rm;
  if (running_var) running_var[c] = rv;
}

// Single-process fast path: the last reduction stage fused with bn_finalize (MODE 0) or with
// bn_param_grad (MODE 1) -- one launch less per BatchNorm layer and direction.  One workgroup owns
// 32 channels and walks the groups in order (the running statistics are updated group by group).
template <typename TIN, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_fused_kernel(const TIN* __restrict__ in, double* __restrict__ sums, int G, int rows,
                                                              int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ bnp, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, double count, float eps,
                                                              float momentum, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              P2PTail x) {
  __shared__ double sh[8][2][32];
  const int t = threadIdx.x, cl = t & 31, sl = t >> 5;
  const int c = blockIdx.x * 32 + cl;
  float rm = 0.f, rv = 0.f;
  double g1 = 0.0, g2 = 0.0;
  if (MODE == 0 && sl == 0 && c < C) { rm = running_mean ? running_mean[c] : 0.f; rv = running_var ? running_var[c] : 0.f; }
  for (int gi = 0; gi < G; ++gi) {
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
      const TIN* p = in + (size_t)gi * rows * 2 * C;
      for (int b = sl; b < rows; b += 8) {
        a0 += (double)p[(size_t)b * 2 * C + c];
        a1 += (double)p[(size_t)b * 2 * C + C + c];
      }
    }
    __syncthreads();
    sh[sl][0][cl] = a0;
    sh[sl][1][cl] = a1;
    __syncthreads();
    if (sl == 0 && c < C) {
      double r0 = 0.0, r1 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { r0 += sh[k][0][cl]; r1 += sh[k][1][cl]; }
      bn_store_sum(x, &sums[((size_t)gi * 2 + 0) * C + c], r0);
      bn_store_sum(x, &sums[((size_t)gi * 2 + 1) * C + c], r1);
      if (MODE == 0) {
        const double mean = r0 / count;
        double var = r1 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * invstd;
        float* o = bnp + (size_t)gi * 4 * C;
        o[c] = scale;
        o[C + c] = beta[c] - (float)mean * scale;
        o[2 * C + c] = (float)mean;
        o[3 * C + c] = invstd;
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        bn_running_update(rm, rv, momentum, mean, unbiased);
      } else {
        g1 += r0; g2 += r1;
      }
    }
  }
  if (sl == 0 && c < C) {
    if (MODE == 0) {
      if (running_mean) running_mean[c] = rm;
      if (running_var) running_var[c] = rv;
    } else {
      dbeta[c] += (float)g1;
      dgamma[c] += (float)g2;
    }
  }
  if (MODE != 0 && x.peers) bn_xchg_tail(x, sums, G * 2 * C, gridDim.x);
}

// Statistics straight from the stored bf16 conv output for SMALL groups (the head's Linear layers: 32 rows per view):
// lets one conv launch cover all groups when a group is not a multiple of the 128-pixel statistics rows.
__global__ __launch_bounds__(256) void bn_stats_raw_kernel(const bf16_t* __restrict__ x, double* __restrict__ sums, int G, int rows, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ bnp, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, double count, float eps, float momentum) {
  __shared__ double sh[8][2][32];
  const int t = threadIdx.x, cl = t & 31, sl = t >> 5;
  const int c = blockIdx.x * 32 + cl;
  float rm = 0.f, rv = 0.f;
  if (sl == 0 && c < C) { rm = running_mean ? running_mean[c] : 0.f; rv = running_var ? running_var[c] : 0.f; }
  for (int gi = 0; gi < G; ++gi) {
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
      const bf16_t* p = x + (size_t)gi * rows * C;
      for (int b = sl; b < rows; b += 8) {
        const double v = (double)bf2f(p[(size_t)b * C + c]);
        a0 += v;
        a1 += v * v;
      }
    }
    __syncthreads();
    sh[sl][0][cl] = a0;
    sh[sl][1][cl] = a1;
    __syncthreads();
    if (sl == 0 && c < C) {
      double r0 = 0.0, r1 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { r0 += sh[k][0][cl]; r1 += sh[k][1][cl]; }
      sums[((size_t)gi * 2 + 0) * C + c] = r0;
      sums[((size_t)gi * 2 + 1) * C + c] = r1;
      const double mean = r0 / count;
      double var = r1 / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float scale = gamma[c] * invstd;
      float* o = bnp + (size_t)gi * 4 * C;
      o[c] = scale;
      o[C + c] = beta[c] - (float)mean * scale;
      o[2 * C + c] = (float)mean;
      o[3 * C + c] = invstd;
      const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
      bn_running_update(rm, rv, momentum, mean, unbiased);
    }
  }
  if (sl == 0 && c < C) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}
int vfs_bn_stats_raw_launch(const bf16_t* x, double* sums, int G, int rows, int C, const float* gamma, const float* beta, float* bnp,
                            float* rm, float* rv, double count, float eps, float momentum, hipStream_t s) {
  if (G <= 0 || rows <= 0 || C <= 0) return vfs_set_error(VFS_ERR_SHAPE, "bn_stats_raw: empty");
  hipLaunchKernelGGL(bn_stats_raw_kernel, dim3((C + 31) / 32), dim3(256), 0, s, x, sums, G, rows, C, gamma, beta, bnp, rm, rv, count,
                     eps, momentum);
  return vfs_check_launch("bn_stats_raw");
}

// The same two stages in ONE launch for large row counts: grid (channel blocks, groups, row chunks);
// every workgroup writes the fp64 sums of its chunk to `chunks`, publishes them (agent-scope fence)
// and draws a ticket for its channel block; the workgroup that draws the LAST ticket of a channel
// block finishes that block: fixed-order sum over the chunks, then the fused finalize (MODE 0),
// parameter-gradient (MODE 1) or plain sums (MODE 2) step.  Deterministic (the summation order does
// not depend on which workgroup is last); the ticket counters return to zero.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_ticket_kernel(const float* __restrict__ in, double* __restrict__ sums,
                                                               double* __restrict__ chunks, unsigned* __restrict__ tickets,
                                                               int G, int bpg, int C, int rpc, int nchunks,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ bnp, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, double count, float eps,
                                                               float momentum, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               P2PTail x) {
  __shared__ double sh[8][2][32];
  __shared__ unsigned s_ticket;
  const int t = threadIdx.x, cl = t & 31, sl = t >> 5;
  const int c = blockIdx.x * 32 + cl;
  {
    const int gi = blockIdx.y, ch = blockIdx.z;
    const int r0 = ch * rpc, r1 = min(bpg, r0 + rpc);
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
      const float* p = in + (size_t)gi * bpg * 2 * C;
      int b = r0 + sl;
      for (; b + 24 < r1; b += 32) {       // four rows (eight independent loads) in flight per lane
        const float x0 = p[(size_t)b * 2 * C + c], y0 = p[(size_t)b * 2 * C + C + c];
        const float x1 = p[(size_t)(b + 8) * 2 * C + c], y1 = p[(size_t)(b + 8) * 2 * C + C + c];
        const float x2 = p[(size_t)(b + 16) * 2 * C + c], y2 = p[(size_t)(b + 16) * 2 * C + C + c];
        const float x3 = p[(size_t)(b + 24) * 2 * C + c], y3 = p[(size_t)(b + 24) * 2 * C + C + c];
        a0 += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
        a1 += ((double)y0 + (double)y1) + ((double)y2 + (double)y3);
      }
      for (; b < r1; b += 8) {
        a0 += (double)p[(size_t)b * 2 * C + c];
        a1 += (double)p[(size_t)b * 2 * C + C + c];
      }
    }
    sh[sl][0][cl] = a0;
    sh[sl][1][cl] = a1;
    __syncthreads();
    if (sl == 0 && c < C) {
      double r0s = 0.0, r1s = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { r0s += sh[k][0][cl]; r1s += sh[k][1][cl]; }
      double* o = chunks + ((size_t)gi * nchunks + ch) * 2 * C;
      vfs_store_agent(&o[c], r0s);       // device-coherent stores (no L2 write-back fence, see vfs_common.h)
      vfs_store_agent(&o[C + c], r1s);
    }
  }
  vfs_release_workgroup();               // this wave's chunk sums have been performed
  __syncthreads();
  if (t == 0) s_ticket = vfs_ticket_agent(&tickets[blockIdx.x]);
  __syncthreads();
  if (s_ticket != (unsigned)(gridDim.y * gridDim.z) - 1u) return;
  float rm = 0.f, rv = 0.f;
  double g1 = 0.0, g2 = 0.0;
  if (MODE == 0 && sl == 0 && c < C) { rm = running_mean ? running_mean[c] : 0.f; rv = running_var ? running_var[c] : 0.f; }
  for (int gi = 0; gi < G; ++gi) {
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
      const double* p = chunks + (size_t)gi * nchunks * 2 * C;
      int b = sl;                          // device-coherent loads of the other workgroups' sums, four chunks
      for (; b + 24 < nchunks; b += 32) {  // (eight loads) in flight per lane: they bypass the L2, ~1 us each
        const double x0 = vfs_load_agent(&p[(size_t)b * 2 * C + c]), y0 = vfs_load_agent(&p[(size_t)b * 2 * C + C + c]);
        const double x1 = vfs_load_agent(&p[(size_t)(b + 8) * 2 * C + c]), y1 = vfs_load_agent(&p[(size_t)(b + 8) * 2 * C + C + c]);
        const double x2 = vfs_load_agent(&p[(size_t)(b + 16) * 2 * C + c]), y2 = vfs_load_agent(&p[(size_t)(b + 16) * 2 * C + C + c]);
        const double x3 = vfs_load_agent(&p[(size_t)(b + 24) * 2 * C + c]), y3 = vfs_load_agent(&p[(size_t)(b + 24) * 2 * C + C + c]);
        a0 += (x0 + x1) + (x2 + x3);
        a1 += (y0 + y1) + (y2 + y3);
      }
      for (; b < nchunks; b += 8) {
        a0 += vfs_load_agent(&p[(size_t)b * 2 * C + c]);
        a1 += vfs_load_agent(&p[(size_t)b * 2 * C + C + c]);
      }
    }
    __syncthreads();
    sh[sl][0][cl] = a0;
    sh[sl][1][cl] = a1;
    __syncthreads();
    if (sl == 0 && c < C) {
      double r0 = 0.0, r1 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { r0 += sh[k][0][cl]; r1 += sh[k][1][cl]; }
      bn_store_sum(x, &sums[((size_t)gi * 2 + 0) * C + c], r0);
      bn_store_sum(x, &sums[((size_t)gi * 2 + 1) * C + c], r1);
      if (MODE == 0) {
        const double mean = r0 / count;
        double var = r1 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * invstd;
        float* o = bnp + (size_t)gi * 4 * C;
        o[c] = scale;
        o[C + c] = beta[c] - (float)mean * scale;
        o[2 * C + c] = (float)mean;
        o[3 * C + c] = invstd;
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        bn_running_update(rm, rv, momentum, mean, unbiased);
      } else if (MODE == 1) {
        g1 += r0; g2 += r1;
      }
    }
  }
  if (sl == 0 && c < C) {
    if (MODE == 0) {
      if (running_mean) running_mean[c] = rm;
      if (running_var) running_var[c] = rv;
    } else if (MODE == 1) {
      dbeta[c] += (float)g1;
      dgamma[c] += (float)g2;
    }
  }
  if (t == 0) tickets[blockIdx.x] = 0u;  // ready for the next launch on this stream
  if (MODE != 0 && x.peers) bn_xchg_tail(x, sums, G * 2 * C, gridDim.x);      // one finisher per channel block arrives here
}

// eval-mode BN: bnp from running statistics (G = 1)
__global__ __launch_bounds__(256) void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ running_mean,
                                                             const float* __restrict__ running_var, float* __restrict__ bnp,
                                                             int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float scale = gamma[c] * invstd;
  bnp[c] = scale;
  bnp[C + c] = beta[c] - running_mean[c] * scale;
  bnp[2 * C + c] = running_mean[c];
  bnp[3 * C + c] = invstd;
}

__device__ __forceinline__ void ld8f(const float* p, float* f) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
  f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}


// Streaming geometry shared by the elementwise BatchNorm kernels: a workgroup owns `ppb` pixels of ONE
// statistics group x one slab of <= 64 channels (blockIdx.y); thread = (pixel lane rt, 8-channel chunk
// ct).  The per-channel coefficients stay in registers for the whole block and four pixel rows
// (8-12 independent 16-byte loads per lane) are in flight per trip; no per-element index division.
struct SlabGeom {
  int cv, rows, rt, c, gi;
  long long m0, mend;
};
// `wide` (the plain, non-FIN launches on tensors of >= 128 channels): the slab is up to 512 channels (1 KB per pixel row, a whole
// wave per row) instead of 64 - a workgroup then streams whole pixel rows linearly instead of 128-byte pieces at a stride of
// 2 C bytes.  Measured ceiling of this box for the same traffic pattern (2 reads + 1 write, torch's elementwise add): 5.7-6.0 TB/s;
// the 64-channel slabs reached 4.4-4.8 (tools/probe_stream_bw.py, MEASUREMENTS.md).  The bit-packed mask keeps its slab-major
// layout (64-channel sub-slabs, vfs_common.h): a lane still writes / reads the byte of its own 8 channels.
__device__ __forceinline__ int slab_width(int C, int wide) { return C < 64 ? C : (wide ? (C < 512 ? C : 512) : 64); }
__device__ __forceinline__ SlabGeom slab_geom(long long M, int C, int mpg, int ppb, int wide = 0) {
  SlabGeom s;
  const int cslab = slab_width(C, wide);
  s.cv = cslab >> 3;
  s.rows = 256 / s.cv;
  const int t = threadIdx.x;
  s.rt = t / s.cv;
  s.c = blockIdx.y * cslab + (t - s.rt * s.cv) * 8;
  const int bpg = (mpg + ppb - 1) / ppb;
  s.gi = blockIdx.x / bpg;
  s.m0 = (long long)s.gi * mpg + (long long)(blockIdx.x - s.gi * bpg) * ppb;
  long long e = s.m0 + ppb;
  const long long ge = (long long)(s.gi + 1) * mpg;
  if (e > ge) e = ge;
  if (e > M) e = M;
  s.mend = e;
  return s;
}
// host side: pixels per block (a multiple of the 4-row trip) for ~4096 workgroups, and the grid
static inline int slab_wide_ok(int C) { return C >= 128 && (C <= 512 ? (512 % C == 0 || C % 64 == 0) && 256 % (C >> 3) == 0 : C % 512 == 0); }
static inline dim3 slab_grid(long long M, int C, int mpg, int* ppb_out, int wide = 0) {
  const int cslab = C < 64 ? C : (wide ? (C < 512 ? C : 512) : 64), rows = 256 / (cslab >> 3), unit = 4 * rows;
  const int slabs = C < 64 ? 1 : C / cslab;
  long long per = (M * slabs + 4095) / 4096;
  long long ppb = ((per + unit - 1) / unit) * unit;
  if (ppb < unit) ppb = unit;
  const int bpg = (int)((mpg + ppb - 1) / ppb);
  const int G = (int)((M + mpg - 1) / mpg);
  *ppb_out = (int)ppb;
  return dim3(G * bpg, slabs);
}
static inline bool slab_ok(int C) { return C % 8 == 0 && (C < 64 ? 256 % (C >> 3) == 0 : C % 64 == 0); }

// prologue helper of the FIN kernels: double sums (row 0, row 1) of the partial rows of group gi for the <= 64
// channels of this workgroup's slab -> sums_s[2][64] (valid after the call; contains barriers, call uniformly)
__device__ __forceinline__ void slab_rows_reduce(const BnFin& f, int gi, int C, int cslab, double (*red)[2][64], double (*sums_s)[64]) {
  const int t = threadIdx.x, ch = t & 63, rl = t >> 6;
  const int c = blockIdx.y * cslab + ch;
  double a0 = 0.0, a1 = 0.0;
  if (ch < cslab) {
    const float* p = f.partial + (size_t)gi * f.bpg * 2 * C + c;
    int b = rl;
    for (; b + 12 < f.bpg; b += 16) {      // four rows in flight (eight independent loads), fixed summation order
      const float x0 = p[(size_t)b * 2 * C], y0 = p[(size_t)b * 2 * C + C];
      const float x1 = p[(size_t)(b + 4) * 2 * C], y1 = p[(size_t)(b + 4) * 2 * C + C];
      const float x2 = p[(size_t)(b + 8) * 2 * C], y2 = p[(size_t)(b + 8) * 2 * C + C];
      const float x3 = p[(size_t)(b + 12) * 2 * C], y3 = p[(size_t)(b + 12) * 2 * C + C];
      a0 += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
      a1 += ((double)y0 + (double)y1) + ((double)y2 + (double)y3);
    }
    for (; b < f.bpg; b += 4) {
      a0 += (double)p[(size_t)b * 2 * C];
      a1 += (double)p[(size_t)b * 2 * C + C];
    }
  }
  __syncthreads();               // previous users of red / sums_s are done
  red[rl][0][ch] = a0;
  red[rl][1][ch] = a1;
  __syncthreads();
  if (rl == 0) {
    sums_s[0][ch] = (red[0][0][ch] + red[1][0][ch]) + (red[2][0][ch] + red[3][0][ch]);
    sums_s[1][ch] = (red[0][1][ch] + red[1][1][ch]) + (red[2][1][ch] + red[3][1][ch]);
  }
  __syncthreads();
}

template <bool FIN>
__global__ __launch_bounds__(256) void bn_act_kernel(BnActArgs a, BnFin f, int ppb) {
  const SlabGeom s = slab_geom(a.M, a.C, a.mpg, ppb, FIN ? 0 : a.wide);
  float sc[8], sh[8], rsc[8], rsh[8];
  if (FIN) {
    __shared__ double red[4][2][64], sums_s[2][64];
    __shared__ float coef[2][64];
    __shared__ double xloc[256];                 // folded exchange: the lead's sums [g][row][channel of the slab]
    __shared__ int s_failed;
    const int t = threadIdx.x, cslab = a.C < 64 ? a.C : 64;
    const bool lead = blockIdx.x == 0;            // first block of group 0: writes bnp / sums / running statistics
    const bool xch = f.x.peers != nullptr;        // SyncBN: the slab leads exchange their sums with the peers' leads (vfs_p2p.h)
    const int c = blockIdx.y * cslab + t;
    P2PSlab sl;
    sl.slab = blockIdx.y; sl.cslab = cslab; sl.C = a.C;
    sl.epoch = xch ? p2p_fold_epoch(f.x) : 0ull;
    float rm = 0.f, rv = 0.f;
    if (lead && t < cslab) { rm = f.running_mean ? f.running_mean[c] : 0.f; rv = f.running_var ? f.running_var[c] : 0.f; }
    if (xch && lead) {                            // local sums of every group -> sums over the ranks
      for (int g = 0; g < f.G; ++g) {
        slab_rows_reduce(f, g, a.C, cslab, red, sums_s);
        if (t < cslab) { xloc[(2 * g) * cslab + t] = sums_s[0][t]; xloc[(2 * g + 1) * cslab + t] = sums_s[1][t]; }
      }
      __syncthreads();
      p2p_exchange_slab(xloc, f.G * 2 * cslab, sl, f.x, &s_failed);
    } else if (xch) {
      p2p_slab_wait(f.x, sl);                     // the lead of this slab has published the sums over the ranks
    }
    for (int g = lead ? 0 : s.gi; g < (lead ? f.G : s.gi + 1); ++g) {
      if (xch) {
        __syncthreads();
        if (t < cslab) {
          sums_s[0][t] = lead ? xloc[(2 * g) * cslab + t] : vfs_load_agent(f.sums + ((size_t)g * 2 + 0) * a.C + c);
          sums_s[1][t] = lead ? xloc[(2 * g + 1) * cslab + t] : vfs_load_agent(f.sums + ((size_t)g * 2 + 1) * a.C + c);
        }
        __syncthreads();
      } else if (f.partial) {
        slab_rows_reduce(f, g, a.C, cslab, red, sums_s);
      } else {                                    // SyncBN: f.sums already holds the all-reduced totals
        __syncthreads();
        if (t < cslab) {
          sums_s[0][t] = f.sums[((size_t)g * 2 + 0) * a.C + c];
          sums_s[1][t] = f.sums[((size_t)g * 2 + 1) * a.C + c];
        }
        __syncthreads();
      }
      if (t < cslab) {
        const double mean = sums_s[0][t] / f.count;
        double var = sums_s[1][t] / f.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float scale = f.gamma[c] * invstd, shift = f.beta[c] - (float)mean * scale;
        if (g == s.gi) { coef[0][t] = scale; coef[1][t] = shift; }
        if (lead) {
          float* o = f.bnp + (size_t)g * 4 * a.C;
          o[c] = scale; o[a.C + c] = shift; o[2 * a.C + c] = (float)mean; o[3 * a.C + c] = invstd;
          if (xch) {
            vfs_store_agent(f.sums + ((size_t)g * 2 + 0) * a.C + c, sums_s[0][t]);
            vfs_store_agent(f.sums + ((size_t)g * 2 + 1) * a.C + c, sums_s[1][t]);
          } else if (f.partial) {
            f.sums[((size_t)g * 2 + 0) * a.C + c] = sums_s[0][t];
            f.sums[((size_t)g * 2 + 1) * a.C + c] = sums_s[1][t];
          }
          const double unbiased = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
          bn_running_update(rm, rv, f.momentum, mean, unbiased);
        }
      }
    }
    if (xch && lead) p2p_slab_release(f.x, sl);  // the slab's other workgroups may go
    if (lead && t < cslab) {
      if (f.running_mean) f.running_mean[c] = rm;
      if (f.running_var) f.running_var[c] = rv;
    }
    __syncthreads();
    if (s.rt >= s.rows) return;
    const int cl = s.c - blockIdx.y * cslab;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = coef[0][cl + i]; sh[i] = coef[1][cl + i]; }
  } else {
    if (s.rt >= s.rows) return;
    const float* bp = a.bnp + (size_t)s.gi * 4 * a.C + s.c;
    ld8f(bp, sc);
    ld8f(bp + a.C, sh);
  }
  if (a.rres) {
    const float* rp = a.rbnp + (size_t)s.gi * 4 * a.C + s.c;
    ld8f(rp, rsc);
    ld8f(rp + a.C, rsh);
#pragma unroll
    for (int i = 0; i < 8; ++i) sh[i] += rsh[i];
  }
  for (long long r = s.m0 + s.rt; r < s.mend; r += 4 * s.rows) {
    u32x4 xv[4], rv[4], qv[4];
    size_t oo[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long m = r + (long long)u * s.rows;
      ok[u] = m < s.mend;
      oo[u] = (size_t)(ok[u] ? m : s.m0) * a.C + s.c;
      xv[u] = ld16(a.x + oo[u]);
    }
    if (a.res) {              // optional operands: one straight-line batch each
#pragma unroll
      for (int u = 0; u < 4; ++u) rv[u] = ld16(a.res + oo[u]);
    }
    if (a.rres) {
#pragma unroll
      for (int u = 0; u < 4; ++u) qv[u] = ld16(a.rres + oo[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      float x[8];
      unpack8(xv[u], x);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = x[i] * sc[i] + sh[i];
      if (a.res) {
        float q[8];
        unpack8(rv[u], q);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] += q[i];
      }
      if (a.rres) {
        float q[8];
        unpack8(qv[u], q);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] += q[i] * rsc[i];
      }
      if (a.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fmaxf(x[i], 0.f);
      }
      const size_t o = (size_t)(r + (long long)u * s.rows) * a.C + s.c;
      const u32x4 pk = pack8(x);
      st16(a.y + o, pk);
      if (a.mbits) a.mbits[mask8_index(r + (long long)u * s.rows, s.c, a.M, a.C)] = (unsigned char)mask8_of(pk);
    }
  }
}


__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(BnPoolArgs a) {
  const unsigned cv = a.C >> 3;
  const unsigned total = (unsigned)a.N * a.Hp * a.Wp * cv;   // < 2^31 (checked by the launcher): 32-bit index math
  for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < total; v += gridDim.x * 256u) {
    unsigned p = v / cv;
    const int c = (int)(v - p * cv) * 8;
    const int wp = (int)(p % (unsigned)a.Wp); p /= (unsigned)a.Wp;
    const int hp = (int)(p % (unsigned)a.Hp);
    const int n = (int)(p / (unsigned)a.Hp);
    const int gi = n / a.npg;
    float sc[8], sh[8], best[8], xb[8];
    int bi[8], key[8];
    ld8f(a.bnp + (size_t)gi * 4 * a.C + c, sc);
    ld8f(a.bnp + (size_t)gi * 4 * a.C + a.C + c, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) { key[i] = (int)0x80000000u; xb[i] = 0.f; }
    // the nine window vectors are requested together (taps outside the map: clamped address, ignored below); with the loads
    // inside the bounds conditionals every tap was one dependent memory round trip
    u32x4 win[9];
    unsigned valid = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int h = 2 * hp - 1 + dy;
      const bool hok = (unsigned)h < (unsigned)a.H;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int w = 2 * wp - 1 + dx;
        const bool ok = hok && (unsigned)w < (unsigned)a.W;
        valid |= (ok ? 1u : 0u) << (dy * 3 + dx);
        win[dy * 3 + dx] = ld16(a.x + (((size_t)n * a.H + (hok ? h : 2 * hp)) * a.W + (ok ? w : 2 * wp)) * a.C + c);
      }
    }
    // Round 6: value and tap travel in ONE integer key - the stored (bf16) activation in the upper half (non-negative floats order
    // like integers; -0 sorts below +0), 8 - k below it: the larger key is the larger activation, among equals the EARLIER tap (the
    // first maximum wins, as the scan `val > best` in ascending k did).  One compare + two selects per tap and element instead of
    // a rounding sequence, a compare and three selects (the kernel is bound by this vector work: 9 taps x 8 channels per lane).
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (!((valid >> k) & 1u)) continue;
      float x[8];
      unpack8(win[k], x);
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const unsigned pv = pack2bf(fmaxf(x[i] * sc[i] + sh[i], 0.f), fmaxf(x[i + 1] * sc[i + 1] + sh[i + 1], 0.f));  // pool the STORED (bf16) activation
        const int k0 = (int)((pv << 16) | (unsigned)(8 - k)), k1 = (int)((pv & 0xffff0000u) | (unsigned)(8 - k));
        if (k0 > key[i]) { key[i] = k0; xb[i] = x[i]; }
        if (k1 > key[i + 1]) { key[i + 1] = k1; xb[i + 1] = x[i + 1]; }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      best[i] = __builtin_bit_cast(float, (unsigned)key[i] & 0xffff0000u);
      bi[i] = 8 - (key[i] & 0xf);
    }
    const size_t o = ((((size_t)n * a.Hp + hp) * a.Wp) + wp) * a.C + c;
    st16(a.y + o, pack8(best));
    if (a.xpool) st16(a.xpool + o, pack8(xb));   // exact: x is bf16 already
    if (a.idx) {
      // round 6: a pooled activation that is not positive (every tap of the window <= 0 after BatchNorm: no gradient flows) gets code
      // 0xFF, which no position matches - the backward kernels that route the gradient by the code no longer read y for the ReLU mask
#pragma unroll
      for (int i = 0; i < 8; ++i) bi[i] = best[i] > 0.f ? bi[i] : 0xFF;
      u32x2 pk;
      pk.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      pk.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
      st8(a.idx + o, pk);
    }
  }
}


__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(PoolBwdArgs a) {
  const int cv = a.C >> 3;
  const long long total = (long long)a.N * a.H * a.W * cv;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
    long long p = v / cv;
    const int c = (int)(v - p * cv) * 8;
    const int w = (int)(p % a.W); p /= a.W;
    const int h = (int)(p % a.H);
    const int n = (int)(p / a.H);
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = 0.f;
    // windows hp with 2hp-1 <= h <= 2hp+1
    for (int hp = (h) >> 1; hp <= (h + 1) >> 1; ++hp) {
      if (hp >= a.Hp) continue;
      const int dy = h - (2 * hp - 1);
      if (dy < 0 || dy > 2) continue;
      for (int wp = (w) >> 1; wp <= (w + 1) >> 1; ++wp) {
        if (wp >= a.Wp) continue;
        const int dx = w - (2 * wp - 1);
        if (dx < 0 || dx > 2) continue;
        const unsigned code = (unsigned)(dy * 3 + dx);
        const size_t o = ((((size_t)n * a.Hp + hp) * a.Wp) + wp) * a.C + c;
        const u32x2 id = ld8(a.idx + o);
        float gp[8];
        unpack8(ld16(a.gp + o), gp);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned b = ((i < 4 ? id.x : id.y) >> (8 * (i & 3))) & 0xffu;      // 0xFF: ReLU-dead pooled element (bn_relu_maxpool_kernel)
          if (b == code) g[i] += gp[i];
        }
      }
    }
    st16(a.ga + (((size_t)n * a.H + h) * a.W + w) * a.C + c, pack8(g));
  }
}


// One statistics row of the BatchNorm backward: S1 = sum g*mask, S2 = sum g*mask*xhat over the ppb pixels from m0 on, for the <= 64
// channels of slab blockIdx.y; every thread of the workgroup calls it; the row lands in red and thread (ec, ei) of the
// first cv * 16 returns ITS element (channel ec * 8 + (ei & 7), statistic ei >> 3) - summed over the row threads in fixed order.
// Shared by bn_bwd_reduce_kernel (the row goes to HBM) and bn_bwd_apply_kernel<true, true> (round 6: the row never leaves the launch).
__device__ __forceinline__ float bn_bwd_row(const BnBwdArgs& a, long long m0, int ppb, float (*red)[17], int e) {
  const int cslab = a.C < 64 ? a.C : 64;
  const int cv = cslab >> 3;        // chunk-threads per pixel (<= 8)
  const int rows = 256 / cv;        // pixels processed per step
  const int t = threadIdx.x;
  const int ct = t % cv, rt = t / cv;
  const int gi = (int)(m0 / a.mpg);
  const int c = blockIdx.y * cslab + ct * 8;
  const bool bits = a.relu == VFS_MASK_BITS;
  float s1[8], s2[8], mean[8], inv[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (rt < rows) {
    ld8f(a.bnp + (size_t)gi * 4 * a.C + c, sc);
    ld8f(a.bnp + (size_t)gi * 4 * a.C + a.C + c, sh);
    ld8f(a.bnp + (size_t)gi * 4 * a.C + 2 * a.C + c, mean);
    ld8f(a.bnp + (size_t)gi * 4 * a.C + 3 * a.C + c, inv);
    // four pixel rows per trip: 8-12 independent 16-byte loads in flight per lane
    for (int r = rt; r < ppb; r += 4 * rows) {
      u32x4 gv[4], xv[4];
      unsigned ym[4];
      long long moff[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long m = m0 + r + u * rows;
        ok[u] = (r + u * rows < ppb) && (m < a.M);
        const size_t o = (size_t)(ok[u] ? m : m0) * a.C + c;
        gv[u] = ld16(a.g + o);
        xv[u] = ld16(a.x + o);
        moff[u] = ok[u] ? m : m0;
      }
      // the mask operand in its own straight-line batch per mode (a load inside the per-row conditional is waited for alone)
      if (a.y && bits) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ym[u] = mask8_load(a.y, moff[u], c, a.M, a.C);
      } else if (a.y) {
        u32x4 yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) yv[u] = ld16(a.y + (size_t)moff[u] * a.C + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) ym[u] = mask8_of(yv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        float g[8], x[8];
        unpack8(gv[u], g);
        unpack8(xv[u], x);
        if (a.y) {
#pragma unroll
          for (int i = 0; i < 8; ++i) g[i] = ((ym[u] >> i) & 1u) ? g[i] : 0.f;
        } else if (a.relu) {   // plain conv->BN->ReLU unit: the mask is recomputed from x (saves a read)
#pragma unroll
          for (int i = 0; i < 8; ++i) g[i] = (x[i] * sc[i] + sh[i] > 0.f) ? g[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          s1[i] += g[i];
          s2[i] += g[i] * ((x[i] - mean[i]) * inv[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[t][i] = s1[i]; red[t][8 + i] = s2[i]; }
  __syncthreads();
  // thread (ct, i16) sums over the row-threads in fixed order
  float s = 0.f;
  if (e < cv * 16) {
    const int ec = e / 16, ei = e % 16;
    for (int r = 0; r < rows; ++r) s += red[r * cv + ec][ei];
  }
  return s;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a) {
  // workgroup = ppb pixels x one slab of <= 64 channels (blockIdx.y): wide layers (C up to 2048 with
  // few pixels) still fill the chip, and every lane keeps 8-12 independent 16-byte loads in flight
  __shared__ float red[256][17];
  const int cslab = a.C < 64 ? a.C : 64, cv = cslab >> 3;
  const int e = threadIdx.x;        // cv * 16 <= 128 result elements: one per thread
  const float s = bn_bwd_row(a, (long long)blockIdx.x * a.ppb, a.ppb, red, e);
  if (e < cv * 16) {
    const int ec = e / 16, ei = e % 16;
    const int ch = blockIdx.y * cslab + ec * 8 + (ei & 7);
    a.partial[(size_t)blockIdx.x * 2 * a.C + (ei >> 3) * a.C + ch] = s;
  }
}

// pass 2: dx = scale * (gm - S1/count - xhat * S2/count) = A*gm + B*x + D with per-channel
// A = scale, B = -scale*invstd*m2, D = scale*(mean*invstd*m2 - m1) held in registers
// RAW (round 6, with FIN): a group has ONE statistics row (mpg | 512 or mpg < 16: the SimSiam head's BatchNorm1d layers, tiny maps) -
// the prologue computes it from (g, x, mask) itself, in bn_bwd_reduce_kernel's order (bn_bwd_row: the same bits), so that the
// reduction launch in front of the apply pass disappears.
template <bool FIN, bool RAW = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a, BnFin f, int ppb) {
  const SlabGeom s = slab_geom(a.M, a.C, a.mpg, ppb, FIN ? 0 : a.wide);
  double s1d[8], s2d[8];
  if (FIN) {
    __shared__ double red[4][2][64], sums_s[2][64], mine[2][64];
    __shared__ double xloc[256];                 // folded exchange: the lead's sums [g][row][channel of the slab]
    __shared__ int s_failed;
    const int t = threadIdx.x, cslab = a.C < 64 ? a.C : 64;
    const bool lead = blockIdx.x == 0;            // writes sums, accumulates dgamma / dbeta over the groups
    const bool xch = f.x.peers != nullptr;        // SyncBN: S1 / S2 over all ranks (vfs_p2p.h); dgamma / dbeta stay LOCAL sums
    const int c = blockIdx.y * cslab + t;
    P2PSlab sl;
    sl.slab = blockIdx.y; sl.cslab = cslab; sl.C = a.C;
    sl.epoch = xch ? p2p_fold_epoch(f.x) : 0ull;
    double g1 = 0.0, g2 = 0.0;
    if (xch && !lead) {
      p2p_slab_wait(f.x, sl);
      if (t < cslab) {
        mine[0][t] = vfs_load_agent(f.sums + ((size_t)s.gi * 2 + 0) * a.C + c);
        mine[1][t] = vfs_load_agent(f.sums + ((size_t)s.gi * 2 + 1) * a.C + c);
      }
    } else {
      for (int g = lead ? 0 : s.gi; g < (lead ? f.G : s.gi + 1); ++g) {
        if constexpr (RAW) {
          __shared__ float rawred[256][17];
          __syncthreads();               // previous users of sums_s / rawred are done
          const float v = bn_bwd_row(a, (long long)g * a.mpg, a.mpg, rawred, t);
          if (t < (cslab >> 3) * 16) sums_s[(t % 16) >> 3][(t / 16) * 8 + (t & 7)] = (double)v;      // the one row of the group, as the rows path would read it back
          __syncthreads();
        } else {
          slab_rows_reduce(f, g, a.C, cslab, red, sums_s);
        }
        if (t < cslab) {
          if (g == s.gi) { mine[0][t] = sums_s[0][t]; mine[1][t] = sums_s[1][t]; }
          if (lead) {
            if (xch) { xloc[(2 * g) * cslab + t] = sums_s[0][t]; xloc[(2 * g + 1) * cslab + t] = sums_s[1][t]; }
            else {
              f.sums[((size_t)g * 2 + 0) * a.C + c] = sums_s[0][t];
              f.sums[((size_t)g * 2 + 1) * a.C + c] = sums_s[1][t];
            }
            g1 += sums_s[0][t];
            g2 += sums_s[1][t];
          }
        }
      }
      if (xch) {                                  // (lead) local sums -> sums over the ranks, published to the slab
        __syncthreads();
        p2p_exchange_slab(xloc, f.G * 2 * cslab, sl, f.x, &s_failed);
        if (t < cslab) {
          for (int g = 0; g < f.G; ++g) {
            vfs_store_agent(f.sums + ((size_t)g * 2 + 0) * a.C + c, xloc[(2 * g) * cslab + t]);
            vfs_store_agent(f.sums + ((size_t)g * 2 + 1) * a.C + c, xloc[(2 * g + 1) * cslab + t]);
          }
          mine[0][t] = xloc[(2 * s.gi) * cslab + t];
          mine[1][t] = xloc[(2 * s.gi + 1) * cslab + t];
        }
        p2p_slab_release(f.x, sl);
      }
    }
    if (lead && t < cslab) {
      f.dbeta[c] += (float)g1;
      f.dgamma[c] += (float)g2;
    }
    __syncthreads();
    if (s.rt >= s.rows) return;
    const int cl = s.c - blockIdx.y * cslab;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1d[i] = mine[0][cl + i]; s2d[i] = mine[1][cl + i]; }
  } else {
    if (s.rt >= s.rows) return;
    const double* sp = a.sums + (size_t)s.gi * 2 * a.C + s.c;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1d[i] = sp[i]; s2d[i] = sp[a.C + i]; }
  }
  const bool bits = a.relu == VFS_MASK_BITS;
  float A[8], B[8], D[8], sh[8];
  {
    float mean[8], inv[8];
    const float* bp = a.bnp + (size_t)s.gi * 4 * a.C + s.c;
    ld8f(bp, A);
    ld8f(bp + a.C, sh);
    ld8f(bp + 2 * a.C, mean);
    ld8f(bp + 3 * a.C, inv);
    const float rc = (float)(1.0 / a.count);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float m1 = (float)s1d[i] * rc, m2 = (float)s2d[i] * rc;
      B[i] = -A[i] * inv[i] * m2;
      D[i] = A[i] * (mean[i] * inv[i] * m2 - m1);
    }
  }
  for (long long r = s.m0 + s.rt; r < s.mend; r += 4 * s.rows) {
    u32x4 gv[4], xv[4];
    unsigned ym[4];
    long long moff[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long m = r + (long long)u * s.rows;
      ok[u] = m < s.mend;
      moff[u] = ok[u] ? m : s.m0;
      const size_t o = (size_t)moff[u] * a.C + s.c;
      gv[u] = ld16(a.g + o);
      xv[u] = ld16(a.x + o);
    }
    if (a.y && bits) {          // the mask operand in its own straight-line batch per mode (see bn_bwd_reduce_kernel)
#pragma unroll
      for (int u = 0; u < 4; ++u) ym[u] = mask8_load(a.y, moff[u], s.c, a.M, a.C);
    } else if (a.y) {
      u32x4 yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) yv[u] = ld16(a.y + (size_t)moff[u] * a.C + s.c);
#pragma unroll
      for (int u = 0; u < 4; ++u) ym[u] = mask8_of(yv[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      float g[8], x[8], d[8];
      unpack8(gv[u], g);
      unpack8(xv[u], x);
      if (a.y) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = ((ym[u] >> i) & 1u) ? g[i] : 0.f;
      } else if (a.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = (x[i] * A[i] + sh[i] > 0.f) ? g[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = A[i] * g[i] + (B[i] * x[i] + D[i]);
      const size_t o = (size_t)(r + (long long)u * s.rows) * a.C + s.c;
      st16(a.dx + o, pack8(d));
      if (a.gm) st16(a.gm + o, pack8(g));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Stem: BN backward THROUGH the 3x3/s2 max-pool + ReLU without materialising the full-resolution
// gradient.  ga[h][w] = sum over the <=4 windows whose argmax is (h,w) of gp*(yp>0); it is nonzero
// only at argmax positions, so the statistics are gathered per pooled window (pass 1) and dx is
// rebuilt per full-resolution pixel from the (4x smaller, cache-resident) pooled tensors (pass 2).
__global__ __launch_bounds__(256) void stem_pool_bn_bwd_reduce_kernel(StemBwdArgs a) {
  __shared__ float red[256][17];
  const int cv = a.C >> 3, rows = 256 / cv;
  const int t = threadIdx.x, ct = t % cv, rt = t / cv, c = ct * 8;
  const long long P = (long long)a.N * a.Hp * a.Wp;
  const long long p0 = (long long)blockIdx.x * a.ppb;
  const int gi = (int)(p0 / ((long long)a.npg * a.Hp * a.Wp));
  float s1[8], s2[8], mean[8], inv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (rt < rows) {
    ld8f(a.bnp + (size_t)gi * 4 * a.C + 2 * a.C + c, mean);
    ld8f(a.bnp + (size_t)gi * 4 * a.C + 3 * a.C + c, inv);
    if (a.xp) {
      // raw x at the argmax was saved by the forward pooling kernel: a pure stream over three pooled
      // tensors (pooled pixels are linear in memory), four rows in flight per lane
      for (int r = rt; r < a.ppb; r += 4 * rows) {
        u32x4 gv[4], xv[4];
        u32x2 iv[4];      // the ReLU mask comes with the argmax codes (0xFF: pooled activation not positive - bn_relu_maxpool_kernel):
        bool ok[4];       // 8 bytes per vector instead of the 16 of y
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long p = p0 + r + u * rows;
          ok[u] = (r + u * rows < a.ppb) && p < P;
          const size_t o = (size_t)(ok[u] ? p : p0) * a.C + c;
          gv[u] = ld16(a.gp + o);
          iv[u] = ld8(a.idx + o);
          xv[u] = ld16(a.xp + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          float g[8], x[8];
          unpack8(gv[u], g);
          unpack8(xv[u], x);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            g[i] = (((i < 4 ? iv[u].x : iv[u].y) >> (8 * (i & 3))) & 0xffu) != 0xffu ? g[i] : 0.f;
            s1[i] += g[i];
            s2[i] += g[i] * ((x[i] - mean[i]) * inv[i]);
          }
        }
      }
    } else
    for (int r = rt; r < a.ppb; r += rows) {
      long long p = p0 + r;
      if (p >= P) break;
      const int wp = (int)(p % a.Wp); p /= a.Wp;
      const int hp = (int)(p % a.Hp);
      const int n = (int)(p / a.Hp);
      const size_t o = ((((size_t)n * a.Hp + hp) * a.Wp) + wp) * a.C + c;
      float g[8], yp[8];
      unpack8(ld16(a.gp + o), g);
      unpack8(ld16(a.yp + o), yp);
      const u32x2 id = ld8(a.idx + o);
      unsigned code[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        code[i] = ((i < 4 ? id.x : id.y) >> (8 * (i & 3))) & 0xffu;
        g[i] = yp[i] > 0.f ? g[i] : 0.f;
        s1[i] += g[i];
      }
      for (int dy = 0; dy < 3; ++dy) {
        const int h = 2 * hp - 1 + dy;
        if ((unsigned)h >= (unsigned)a.H) continue;
        for (int dx = 0; dx < 3; ++dx) {
          const int w = 2 * wp - 1 + dx;
          if ((unsigned)w >= (unsigned)a.W) continue;
          float x[8];
          unpack8(ld16(a.x + (((size_t)n * a.H + h) * a.W + w) * a.C + c), x);
          const unsigned pos = (unsigned)(dy * 3 + dx);
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (code[i] == pos) s2[i] += g[i] * ((x[i] - mean[i]) * inv[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[t][i] = s1[i]; red[t][8 + i] = s2[i]; }
  __syncthreads();
  for (int e = t; e < cv * 16; e += 256) {
    const int ec = e / 16, ei = e % 16;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[r * cv + ec][ei];
    a.partial[(size_t)blockIdx.x * 2 * a.C + (ei >> 3) * a.C + ec * 8 + (ei & 7)] = s;
  }
}

__global__ __launch_bounds__(256) void stem_pool_bn_bwd_apply_kernel(StemBwdArgs a) {
  __shared__ float tab[STEM_MAX_GROUPS * STEM_TAB];
  stem_fill_table(a, tab, (float)(1.0 / a.count));
  __syncthreads();
  const long long total = (long long)a.N * a.H * a.W * 8;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
    long long p = v >> 3;
    const int c = (int)(v & 7) * 8;
    const int w = (int)(p % a.W); p /= a.W;
    const int h = (int)(p % a.H);
    const int n = (int)(p / a.H);
    float d[8];
    stem_dx_vec(a, tab, n, h, w, c, d);
    st16(a.dx + (((size_t)n * a.H + h) * a.W + w) * 64 + c, pack8(d));
  }
}

int vfs_stem_pool_bn_bwd_reduce_launch(const StemBwdArgs& a, int nblk, hipStream_t s) {
  if (a.C % 8 || 256 % (a.C >> 3)) return vfs_set_error(VFS_ERR_SHAPE, "stem_pool_bn_bwd: C must be 8*2^k");
  hipLaunchKernelGGL(stem_pool_bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, s, a);
  return vfs_check_launch("stem_pool_bn_bwd_reduce");
}
int vfs_stem_pool_bn_bwd_apply_launch(const StemBwdArgs& a, hipStream_t s) {
  if (a.C != 64 || (a.N + a.npg - 1) / a.npg > STEM_MAX_GROUPS) return vfs_set_error(VFS_ERR_SHAPE, "stem_pool_bn_bwd_apply: C == 64, <= 8 groups");
  if ((long long)a.N * a.H * a.W * 64 >= (1ll << 31)) return vfs_set_error(VFS_ERR_SHAPE, "stem_pool_bn_bwd_apply: 2^31 elements or more (32-bit offsets)");
  long long b = ((long long)a.N * a.H * a.W * (a.C >> 3) + 255) / 256;
  hipLaunchKernelGGL(stem_pool_bn_bwd_apply_kernel, dim3((int)(b > 8192 ? 8192 : b)), dim3(256), 0, s, a);
  return vfs_check_launch("stem_pool_bn_bwd_apply");
}

// dgamma[c] += sum_g S2_local[g][c],  dbeta[c] += sum_g S1_local[g][c]  (local sums: DDP averages)
__global__ __launch_bounds__(256) void bn_param_grad_kernel(const double* __restrict__ sums, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int G, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int gi = 0; gi < G; ++gi) {
    s1 += sums[((size_t)gi * 2) * C + c];
    s2 += sums[((size_t)gi * 2 + 1) * C + c];
  }
  dbeta[c] += (float)s1;
  dgamma[c] += (float)s2;
}

// ------------------------------------------------------------------------------------------ launchers
static inline int grid_for(long long total_vec) {
  long long b = (total_vec + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

int vfs_option_bn_wide = 1;          // plain bn_act / bn_bwd_apply on >= 128-channel tensors: whole pixel rows per workgroup (A/B knob)
int vfs_option_bn_wide_min_mb = 8;   // ... from this tensor size on
int vfs_option_bn_ticket = 1;   // capi: vfs_set_option("bn_ticket", 0) = single-workgroup-per-channel-block reduction
// scratch = [VFS_BN_TICKETS ticket counters (zero before the first use, left zero by every launch)]
//           [double[G][<=VFS_BN_MAX_CHUNKS][2][C] chunk sums]
int vfs_option_bn_chunk_rows = 64;   // rows per chunk of the ticket reduction (A/B knob; was 32)
static inline void bn_chunk_plan(int bpg, int* nchunks, int* rpc) {
  int n = (bpg + vfs_option_bn_chunk_rows - 1) / vfs_option_bn_chunk_rows;
  if (n > VFS_BN_MAX_CHUNKS) n = VFS_BN_MAX_CHUNKS;
  *rpc = (bpg + n - 1) / n;
  *nchunks = (bpg + *rpc - 1) / *rpc;
}
int vfs_bn_reduce_fused_launch(int mode, const float* partial, double* sums, double* scratch, int G, int bpg, int C,
                               const float* gamma, const float* beta, float* bnp, float* rm, float* rv, double count, float eps,
                               float momentum, float* dgamma, float* dbeta, hipStream_t s, const P2PTail* tail) {
  const int cb = (C + 31) / 32;
  P2PTail x;
  if (tail) {
    if (mode == 0) return vfs_set_error(VFS_ERR_ARG, "bn_reduce: the fused finalize has no exchange (SyncBN finalizes after the exchange)");
    if (G * 2 * C > P2P_MAXN || tail->world < 1 || tail->world > P2P_MAXW || tail->rank < 0 || tail->rank >= tail->world || !tail->peers || !tail->state)
      return vfs_set_error(VFS_ERR_ARG, "bn_reduce + exchange: G*2*C <= 8192 doubles, world <= 8, peers / state set");
    x = *tail;
  }
  if (vfs_option_bn_ticket && bpg > 64 && scratch != nullptr && cb <= VFS_BN_TICKETS) {   // chunked, single launch (last workgroup finishes)
    int nchunks, rpc;
    bn_chunk_plan(bpg, &nchunks, &rpc);
    unsigned* tickets = reinterpret_cast<unsigned*>(scratch);
    double* chunks = scratch + VFS_BN_TICKETS / 2;
    const dim3 grid(cb, G, nchunks);
    if (mode == 0) hipLaunchKernelGGL((bn_reduce_ticket_kernel<0>), grid, dim3(256), 0, s, partial, sums, chunks, tickets, G, bpg, C, rpc, nchunks, gamma, beta, bnp, rm, rv, count, eps, momentum, dgamma, dbeta, x);
    else if (mode == 1) hipLaunchKernelGGL((bn_reduce_ticket_kernel<1>), grid, dim3(256), 0, s, partial, sums, chunks, tickets, G, bpg, C, rpc, nchunks, gamma, beta, bnp, rm, rv, count, eps, momentum, dgamma, dbeta, x);
    else hipLaunchKernelGGL((bn_reduce_ticket_kernel<2>), grid, dim3(256), 0, s, partial, sums, chunks, tickets, G, bpg, C, rpc, nchunks, gamma, beta, bnp, rm, rv, count, eps, momentum, dgamma, dbeta, x);
    return vfs_check_launch("bn_reduce_ticket");
  }
  if (mode == 0) hipLaunchKernelGGL((bn_reduce_fused_kernel<float, 0>), dim3(cb), dim3(256), 0, s, partial, sums, G, bpg, C, gamma, beta, bnp, rm, rv, count, eps, momentum, dgamma, dbeta, x);
  else if (mode == 1) hipLaunchKernelGGL((bn_reduce_fused_kernel<float, 1>), dim3(cb), dim3(256), 0, s, partial, sums, G, bpg, C, gamma, beta, bnp, rm, rv, count, eps, momentum, dgamma, dbeta, x);
  else hipLaunchKernelGGL((bn_reduce_rows_kernel<float>), dim3(cb, G, 1), dim3(256), 0, s, partial, sums, bpg, C, bpg, 1, x);
  return vfs_check_launch("bn_reduce_fused");
}
// mode 0: statistics -> bnp + running stats ; mode 1: backward sums -> bsums + dgamma/dbeta ; 2: sums only
int vfs_bn_reduce_partials_launch(const float* partial, double* sums, double* scratch, int G, int bpg, int C, hipStream_t s, const P2PTail* tail) {
  return vfs_bn_reduce_fused_launch(2, partial, sums, scratch, G, bpg, C, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, 0.f, 0.f,
                                    nullptr, nullptr, s, tail);
}
int vfs_bn_finalize_launch(const double* sums, const float* gamma, const float* beta, float* bnp, float* rm, float* rv,
                           int G, int C, double count, float eps, float momentum, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, gamma, beta, bnp, rm, rv, G, C,
                     count, eps, momentum);
  return vfs_check_launch("bn_finalize");
}
int vfs_bn_eval_params_launch(const float* gamma, const float* beta, const float* rm, const float* rv, float* bnp, int C,
                              float eps, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, rm, rv, bnp, C, eps);
  return vfs_check_launch("bn_eval_params");
}
int vfs_bn_act_launch(const BnActArgs& a, hipStream_t s) {
  if (!slab_ok(a.C)) return vfs_set_error(VFS_ERR_SHAPE, "bn_act: C must be 8*2^k below 64, a multiple of 64 above");
  if (a.M <= 0) return VFS_OK;
  int ppb;
  BnActArgs aw = a;
  aw.wide = vfs_option_bn_wide && slab_wide_ok(a.C) && (long long)a.M * a.C * 2 >= (long long)vfs_option_bn_wide_min_mb << 20;
  const dim3 grid = slab_grid(a.M, a.C, a.mpg, &ppb, aw.wide);
  hipLaunchKernelGGL((bn_act_kernel<false>), grid, dim3(256), 0, s, aw, BnFin{}, ppb);
  return vfs_check_launch("bn_act");
}
static int fin_check(const BnFin& f, long long M, int mpg, const char* who, bool sums_ok = false) {
  const bool rows = f.partial && f.bpg > 0;
  if (!(rows || (sums_ok && !f.partial)) || !f.sums || f.G <= 0 || (long long)f.G * mpg != M) return vfs_set_error(VFS_ERR_ARG, who);
  return VFS_OK;
}
// the folded SyncBN exchange (BnFin::x): what the windows and the slab layout can hold
static int fin_xchg_check(const BnFin& f, int C, const char* who) {
  if (!f.x.peers) return VFS_OK;
  const int cslab = C < 64 ? C : 64, slabs = C < 64 ? 1 : C / 64;
  if (!f.partial || f.G * 2 * cslab > 256 || slabs > P2P_MAXSLAB || f.G * 2 * C > P2P_MAXN || f.x.world < 1 || f.x.world > P2P_MAXW ||
      f.x.rank < 0 || f.x.rank >= f.x.world || !f.x.state)
    return vfs_set_error(VFS_ERR_ARG, who);
  return VFS_OK;
}
int vfs_bn_act_fin_launch(const BnActArgs& a, const BnFin& f, hipStream_t s) {
  if (fin_xchg_check(f, a.C, "bn_act_fin_xchg: statistics rows, G*2*min(C,64) <= 256, C <= 4096, G*2*C <= 8192, world <= 8, state set")) return VFS_ERR_ARG;
  if (!slab_ok(a.C)) return vfs_set_error(VFS_ERR_SHAPE, "bn_act_fin: C must be 8*2^k below 64, a multiple of 64 above");
  if (a.M <= 0) return vfs_set_error(VFS_ERR_SHAPE, "bn_act_fin: empty");
  if (fin_check(f, a.M, a.mpg, "bn_act_fin: statistics rows / groups", true) || !f.gamma || !f.beta || !f.bnp)
    return vfs_set_error(VFS_ERR_ARG, "bn_act_fin: null operand or groups do not tile M");
  int ppb;
  const dim3 grid = slab_grid(a.M, a.C, a.mpg, &ppb);
  hipLaunchKernelGGL((bn_act_kernel<true>), grid, dim3(256), 0, s, a, f, ppb);
  return vfs_check_launch("bn_act_fin");
}
int vfs_bn_relu_maxpool_launch(const BnPoolArgs& a, hipStream_t s) {
  if (a.C % 8) return vfs_set_error(VFS_ERR_SHAPE, "bn_relu_maxpool: C%8");
  if ((long long)a.N * a.Hp * a.Wp * (a.C >> 3) >= (1ll << 31)) return vfs_set_error(VFS_ERR_SHAPE, "bn_relu_maxpool: too many elements");
  hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(grid_for((long long)a.N * a.Hp * a.Wp * (a.C >> 3))), dim3(256), 0, s, a);
  return vfs_check_launch("bn_relu_maxpool");
}
int vfs_maxpool_relu_bwd_launch(const PoolBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(maxpool_relu_bwd_kernel, dim3(grid_for((long long)a.N * a.H * a.W * (a.C >> 3))), dim3(256), 0, s, a);
  return vfs_check_launch("maxpool_relu_bwd");
}
int vfs_bn_bwd_reduce_launch(const BnBwdArgs& a, int nblk, hipStream_t s) {
  if (a.C % 8 || (a.C < 64 && 256 % (a.C >> 3)) || (a.C >= 64 && a.C % 64))
    return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_reduce: C must be 8*2^k below 64, a multiple of 64 above");
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, a.C < 64 ? 1 : a.C / 64), dim3(256), 0, s, a);
  return vfs_check_launch("bn_bwd_reduce");
}
int vfs_bn_bwd_apply_launch(const BnBwdArgs& a, hipStream_t s) {
  if (!slab_ok(a.C)) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply: C must be 8*2^k below 64, a multiple of 64 above");
  if (a.M <= 0) return VFS_OK;
  int ppb;
  BnBwdArgs aw = a;
  aw.wide = vfs_option_bn_wide && slab_wide_ok(a.C) && (long long)a.M * a.C * 2 >= (long long)vfs_option_bn_wide_min_mb << 20;
  const dim3 grid = slab_grid(a.M, a.C, a.mpg, &ppb, aw.wide);
  hipLaunchKernelGGL((bn_bwd_apply_kernel<false>), grid, dim3(256), 0, s, aw, BnFin{}, ppb);
  return vfs_check_launch("bn_bwd_apply");
}
// the apply pass that computes its single statistics row per group itself (bn_bwd_apply_kernel<true, true>)
int vfs_bn_bwd_apply_raw_launch(const BnBwdArgs& a, const BnFin& f, hipStream_t s) {
  if (!slab_ok(a.C)) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_raw: C must be 8*2^k below 64, a multiple of 64 above");
  if (a.M <= 0 || a.mpg <= 0 || a.M % a.mpg || a.mpg > 512 || f.G != (int)(a.M / a.mpg) || !f.sums || !f.dgamma || !f.dbeta || f.x.peers)
    return vfs_set_error(VFS_ERR_ARG, "bn_bwd_apply_raw: groups of at most 512 rows that tile M, sums / dgamma / dbeta given, no exchange");
  int ppb;
  const dim3 grid = slab_grid(a.M, a.C, a.mpg, &ppb);
  hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), grid, dim3(256), 0, s, a, f, ppb);
  return vfs_check_launch("bn_bwd_apply_raw");
}
int vfs_bn_bwd_apply_fin_launch(const BnBwdArgs& a, const BnFin& f, hipStream_t s) {
  if (fin_xchg_check(f, a.C, "bn_bwd_apply_fin_xchg: statistics rows, G*2*min(C,64) <= 256, C <= 4096, G*2*C <= 8192, world <= 8, state set")) return VFS_ERR_ARG;
  if (!slab_ok(a.C)) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_fin: C must be 8*2^k below 64, a multiple of 64 above");
  if (a.M <= 0) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_fin: empty");
  if (fin_check(f, a.M, a.mpg, "bn_bwd_apply_fin: statistics rows / groups") || !f.dgamma || !f.dbeta)
    return vfs_set_error(VFS_ERR_ARG, "bn_bwd_apply_fin: null operand or groups do not tile M");
  int ppb;
  const dim3 grid = slab_grid(a.M, a.C, a.mpg, &ppb);
  hipLaunchKernelGGL((bn_bwd_apply_kernel<true>), grid, dim3(256), 0, s, a, f, ppb);
  return vfs_check_launch("bn_bwd_apply_fin");
}
int vfs_bn_param_grad_launch(const double* sums, float* dgamma, float* dbeta, int G, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, dgamma, dbeta, G, C);
  return vfs_check_launch("bn_param_grad");
}

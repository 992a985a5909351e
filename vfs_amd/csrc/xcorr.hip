// SiamFC cross-correlation head (§8f rank 4): out[m][i][j] = scale * sum_{u,v,c} z[m % nz][u][v][c] * x[m][i+u][j+v][c]
// - what `_fast_xcorr` computes with a grouped conv2d (projects/siamfc-pytorch/siamfc/heads.py:16-23,51-58:
// x.view(-1, nz*c, h, w) convolved with z, groups = nz, i.e. search feature m is correlated with exemplar m % nz).
// NHWC bf16 operands, fp32 accumulation and output.  The op is tiny (3 x 18 x 18 responses of a 15 x 15 x 512 filter:
// 0.2 GFLOP) and reads the 3 MB search feature from L2: one wave per response element, 16-byte loads along the
// channels, a wave reduction at the end - no MFMA reshaping of what is a bandwidth / latency problem.
#include "vfs_common.h"
#include "vfs_ops.h"

__global__ __launch_bounds__(256) void xcorr_fwd_kernel(XcorrArgs a) {
  const int lane = threadIdx.x & 63;
  const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per output element
  const int ho = a.H - a.Hz + 1, wo = a.W - a.Wz + 1;
  const long long total = (long long)a.nx * ho * wo;
  if (e >= total) return;
  const int j = (int)(e % wo), i = (int)((e / wo) % ho), m = (int)(e / ((long long)wo * ho));
  const bf16_t* z = a.z + (size_t)(m % a.nz) * a.Hz * a.Wz * a.C;
  const bf16_t* x = a.x + ((size_t)m * a.H * a.W + (size_t)i * a.W + j) * a.C;
  const int cv = a.C >> 3;                       // 16-byte chunks per pixel
  const int rowv = a.Wz * cv;                    // chunks of one filter row: contiguous in z, and in x (window row)
  float acc = 0.f;
  for (int u = 0; u < a.Hz; ++u) {
    const bf16_t* zr = z + (size_t)u * a.Wz * a.C;
    const bf16_t* xr = x + (size_t)u * a.W * a.C;
    for (int k = lane; k < rowv; k += 64) {
      float fz[8], fx[8];
      unpack8(ld16(zr + (size_t)k * 8), fz);
      unpack8(ld16(xr + (size_t)k * 8), fx);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += fz[q] * fx[q];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) a.out[e] = acc * a.scale;
}

int vfs_xcorr_fwd_launch(const XcorrArgs& a, hipStream_t s) {
  if (a.nz <= 0 || a.nx <= 0 || a.nx % a.nz) return vfs_set_error(VFS_ERR_SHAPE, "xcorr: nx must be a multiple of nz");
  if (a.C % 8 || a.Hz > a.H || a.Wz > a.W || a.Hz <= 0 || a.Wz <= 0) return vfs_set_error(VFS_ERR_SHAPE, "xcorr: C % 8, filter <= search size");
  const long long total = (long long)a.nx * (a.H - a.Hz + 1) * (a.W - a.Wz + 1);
  if (total >= (1ll << 31)) return vfs_set_error(VFS_ERR_SHAPE, "xcorr: too many outputs");
  hipLaunchKernelGGL(xcorr_fwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a);
  return vfs_check_launch("xcorr_fwd");
}

// ---------------------------------------------------------------------------------------------
// Training the probe (projects/siamfc-pytorch/siamfc/siamfc_tracker_base.py:364-387 `train_step`): backward of the
// cross-correlation.  g[m][i][j] = dL/d(out[m][i][j]) (fp32); gradients leave as bf16 NHWC - they feed the 1x1 convs'
// weight-gradient kernel (vfs_conv_wgrad) exactly as any other dY.
//   dz[k][u][v][c] = scale * sum_{m : m % nz == k} sum_{i,j} g[m][i][j] * x[m][i+u][j+v][c]
//   dx[m][p][q][c] = scale * sum_{u,v} g[m][p-u][q-v] * z[m % nz][u][v][c]
// one workgroup per output pixel, lanes along the channels (8 per lane, 16-byte loads); the response map of one pair
// (<= 18 x 18 floats) is read from LDS.
#define XC_MAX_RESP 4096
__global__ __launch_bounds__(256) void xcorr_bwd_z_kernel(XcorrBwdArgs a) {
  __shared__ float sg[XC_MAX_RESP];     // [ho*wo] response gradient of the current pair
  const int ho = a.H - a.Hz + 1, wo = a.W - a.Wz + 1;
  const int v = blockIdx.x % a.Wz, u = (blockIdx.x / a.Wz) % a.Hz, k = blockIdx.x / (a.Wz * a.Hz);
  const int cv = a.C >> 3;
  for (int c8 = threadIdx.x; c8 < cv || c8 - threadIdx.x < cv; c8 += 256) {      // uniform trip count (barriers inside)
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int m = k; m < a.nx; m += a.nz) {
      __syncthreads();
      for (int e = threadIdx.x; e < ho * wo; e += 256) sg[e] = a.g[(size_t)m * ho * wo + e];
      __syncthreads();
      if (c8 < cv) {
        const bf16_t* xb = a.x + ((size_t)m * a.H * a.W) * a.C + (size_t)c8 * 8;
        for (int i = 0; i < ho; ++i)
          for (int j = 0; j < wo; ++j) {
            float fx[8];
            unpack8(ld16(xb + ((size_t)(i + u) * a.W + (j + v)) * a.C), fx);
            const float gv = sg[i * wo + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += gv * fx[q];
          }
      }
    }
    if (c8 < cv) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] *= a.scale;
      st16(a.dz + (((size_t)k * a.Hz + u) * a.Wz + v) * a.C + (size_t)c8 * 8, pack8(acc));
    }
  }
}

__global__ __launch_bounds__(256) void xcorr_bwd_x_kernel(XcorrBwdArgs a) {
  __shared__ float sg[XC_MAX_RESP];
  const int ho = a.H - a.Hz + 1, wo = a.W - a.Wz + 1;
  const int q0 = blockIdx.x % a.W, p = (blockIdx.x / a.W) % a.H, m = blockIdx.x / (a.W * a.H);
  for (int e = threadIdx.x; e < ho * wo; e += 256) sg[e] = a.g[(size_t)m * ho * wo + e];
  __syncthreads();
  const bf16_t* zb = a.z + (size_t)(m % a.nz) * a.Hz * a.Wz * a.C;
  const int cv = a.C >> 3;
  for (int c8 = threadIdx.x; c8 < cv; c8 += 256) {
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int u = 0; u < a.Hz; ++u) {
      const int i = p - u;
      if (i < 0 || i >= ho) continue;
      for (int v = 0; v < a.Wz; ++v) {
        const int j = q0 - v;
        if (j < 0 || j >= wo) continue;
        float fz[8];
        unpack8(ld16(zb + ((size_t)u * a.Wz + v) * a.C + (size_t)c8 * 8), fz);
        const float gv = sg[i * wo + j];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += gv * fz[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] *= a.scale;
    st16(a.dx + (((size_t)m * a.H + p) * a.W + q0) * a.C + (size_t)c8 * 8, pack8(acc));
  }
}

int vfs_xcorr_bwd_launch(const XcorrBwdArgs& a, hipStream_t s) {
  if (a.nz <= 0 || a.nx <= 0 || a.nx % a.nz) return vfs_set_error(VFS_ERR_SHAPE, "xcorr_bwd: nx must be a multiple of nz");
  if (a.C % 8 || a.Hz > a.H || a.Wz > a.W || a.Hz <= 0 || a.Wz <= 0) return vfs_set_error(VFS_ERR_SHAPE, "xcorr_bwd: C % 8, filter <= search size");
  const int ho = a.H - a.Hz + 1, wo = a.W - a.Wz + 1;
  if (ho * wo > XC_MAX_RESP) return vfs_set_error(VFS_ERR_SHAPE, "xcorr_bwd: response map larger than 4096 elements");
  if (a.dz) {
    hipLaunchKernelGGL(xcorr_bwd_z_kernel, dim3(a.nz * a.Hz * a.Wz), dim3(256), 0, s, a);
    int rc = vfs_check_launch("xcorr_bwd_z");
    if (rc) return rc;
  }
  if (a.dx) {
    hipLaunchKernelGGL(xcorr_bwd_x_kernel, dim3(a.nx * a.H * a.W), dim3(256), 0, s, a);
    return vfs_check_launch("xcorr_bwd_x");
  }
  return VFS_OK;
}

// The probe's losses on the response maps (projects/siamfc-pytorch/siamfc/losses.py), value and gradient in one launch
// of ONE workgroup (a batch of response maps is a few thousand elements):
//   mode 0  BalancedLoss (:24-41): weights 1/#pos on labels == 1, neg_weight/#neg on labels == 0, normalised to sum 1;
//           loss = sum w * BCEWithLogits(x, t);  dL/dx = w * (sigmoid(x) - t)
//   mode 1  FocalLoss (:44-65, gamma = param): l = -(t (1-p)^g log p + (1-t) p^g log(1-p)), a = t (1-p)^g + (1-t) p^g,
//           loss = mean(l / mean(a)) - the normaliser mean(a) is part of the graph:  dL/dx_j = (l'_j - loss * a'_j) / (n mean(a))
// scale = dL/d(loss) (1 for loss.backward()).  loss_out[0] = loss.
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float log_sigmoid_f(float x) {      // losses.py:8-13
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void siamfc_loss_kernel(const float* __restrict__ x, const float* __restrict__ tgt, float* __restrict__ loss_out,
                                                          float* __restrict__ grad, int n, int mode, float param, float scale) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  if (mode == 0) {
    float np_ = 0.f, nn_ = 0.f;
    for (int i = t; i < n; i += 256) { np_ += tgt[i] == 1.f ? 1.f : 0.f; nn_ += tgt[i] == 0.f ? 1.f : 0.f; }
    const float pos = block_sum(np_, red), neg = block_sum(nn_, red);
    const float wp = pos > 0.f ? 1.f / pos : 0.f, wn = neg > 0.f ? param / neg : 0.f;
    const float wsum = wp * pos + wn * neg;
    float l = 0.f;
    for (int i = t; i < n; i += 256) {
      const float w = (tgt[i] == 1.f ? wp : (tgt[i] == 0.f ? wn : 0.f)) / wsum;
      const float xi = x[i];
      l += w * (fmaxf(xi, 0.f) - xi * tgt[i] + log1pf(expf(-fabsf(xi))));
      if (grad) grad[i] = scale * w * (1.f / (1.f + expf(-xi)) - tgt[i]);
    }
    l = block_sum(l, red);
    if (t == 0) loss_out[0] = l;
    return;
  }
  float ls = 0.f, as = 0.f;
  for (int i = t; i < n; i += 256) {
    const float xi = x[i], ti = tgt[i];
    const float p = 1.f / (1.f + expf(-xi));
    const float pw = powf(1.f - p, param), nw = powf(p, param);
    ls += -(ti * pw * log_sigmoid_f(xi) + (1.f - ti) * nw * log_sigmoid_f(-xi));
    as += ti * pw + (1.f - ti) * nw;
  }
  const float S = block_sum(ls, red), A = block_sum(as, red) / (float)n;
  const float L = S / ((float)n * A);
  if (t == 0) loss_out[0] = L;
  if (!grad) return;
  for (int i = t; i < n; i += 256) {
    const float xi = x[i], ti = tgt[i];
    const float p = 1.f / (1.f + expf(-xi)), q = 1.f - p;
    const float pw = powf(q, param), nw = powf(p, param);
    const float lp = log_sigmoid_f(xi), ln = log_sigmoid_f(-xi);
    // d/dx[(1-p)^g log p] = (1-p)^g ((1-p) - g p log p);  d/dx[p^g log(1-p)] = p^g (g (1-p) log(1-p) - p)
    const float dl = -(ti * pw * (q - param * p * lp) + (1.f - ti) * nw * (param * q * ln - p));
    const float da = param * p * q * (-(ti) * powf(q, param - 1.f) + (1.f - ti) * powf(p, param - 1.f));
    grad[i] = scale * (dl - L * da) / ((float)n * A);
  }
}
int vfs_siamfc_loss_launch(const float* x, const float* tgt, float* loss_out, float* grad, int n, int mode, float param, float scale,
                           hipStream_t s) {
  if (n < 1 || (mode != 0 && mode != 1)) return vfs_set_error(VFS_ERR_ARG, "siamfc_loss: n >= 1, mode 0 (balanced) or 1 (focal)");
  hipLaunchKernelGGL(siamfc_loss_kernel, dim3(1), dim3(256), 0, s, x, tgt, loss_out, grad, n, mode, param, scale);
  return vfs_check_launch("siamfc_loss");
}

// torch.optim.Adam (amsgrad off), the probe's default optimizer (default_config_base.py:33; siamfc_tracker_base.py:139-145):
// g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  if (wd != 0.f) gi += wd * p[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
}
int vfs_adam_launch(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, int step,
                    hipStream_t s) {
  if (step < 1) return vfs_set_error(VFS_ERR_ARG, "adam_step: step >= 1");
  const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s);
  return vfs_check_launch("adam_step");
}

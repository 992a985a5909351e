// DAVIS label propagation for gfx950: local-window affinity on MFMA + streaming top-k + softmax
// + value gather in ONE kernel, and the bilinear-upsample / min-max / argmax post-processing.
//
// Replaces masked_attention_efficient (mmaction/models/common/local_attention.py:237-348) with
// the circular spatial_neighbor mask (common/affinity_utils.py:144-156), as called per frame by
// VanillaTracker.forward_test (trackers/vanilla_tracker.py:132-181).  The reference materialises
// the dense [T*HW, HW] affinity in chunks of 32 queries and a [HW,HW] boolean mask; here
//   * one workgroup owns an 8x8 tile of queries and walks only the key window that can be inside
//     the circle ((8+2(r-1))^2 keys per key frame), 128 keys per step, channels in K-steps of 64
//     through the same swizzled-LDS / v_mfma_f32_16x16x32_bf16 pipeline as the convolutions
//     (rows = keys, cols = queries: a lane owns 4 keys of ONE query per 16x16 tile)
//   * the mask is the integer test dy^2+dx^2 < r^2, scores are scaled by 1/temperature and fed
//     to a per-lane sorted top-10 list (registers); the 4 lanes that share a query merge through
//     LDS at the end, then softmax over the 10 and the weighted sum of the value logits
//   * features are L2-normalised ONCE per frame when they enter the bank (l2norm kernel), not per
//     propagation step.
#include "vfs_conv.h"
#include "vfs_ops.h"

#define LP_TOPK 10

__global__ __launch_bounds__(256) void l2norm_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long P,
                                                          int C) {
  // one wave per pixel row of C channels (F.normalize(p=2, dim=channel, eps=1e-12))
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  const bf16_t* src = x + (size_t)row * C;
  float ss = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    unpack8(ld16(src + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    unpack8(ld16(src + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= inv;
    st16(y + (size_t)row * C + c, pack8(f));
  }
}

int vfs_l2norm_rows_launch(const bf16_t* x, bf16_t* y, long long P, int C, hipStream_t s) {
  if (C % 8) return vfs_set_error(VFS_ERR_SHAPE, "l2norm: C%8");
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, s, x, y, P, C);
  return vfs_check_launch("l2norm_rows");
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void topk_insert(float (&tv)[LP_TOPK], int (&ti)[LP_TOPK], float s, int id) {
  if (s > tv[LP_TOPK - 1]) { tv[LP_TOPK - 1] = s; ti[LP_TOPK - 1] = id; }
#pragma unroll
  for (int j = LP_TOPK - 1; j > 0; --j) {
    const bool sw = tv[j] > tv[j - 1];
    const float a = tv[j - 1], b = tv[j];
    const int ia = ti[j - 1], ib = ti[j];
    tv[j - 1] = sw ? b : a; tv[j] = sw ? a : b;
    ti[j - 1] = sw ? ib : ia; ti[j] = sw ? ia : ib;
  }
}

__global__ __launch_bounds__(256) void labelprop_kernel(LabelPropArgs a) {
  constexpr int BQ = 64, BK = 128, TM = 8;
  __shared__ __attribute__((aligned(16))) bf16_t sK[2][BK * 64];
  __shared__ __attribute__((aligned(16))) bf16_t sQ[2][BQ * 64];
  __shared__ int sKC[BK];   // packed (ky << 16 | kx) of the key rows, -1 = outside the window
  // the merge area aliases sK after the main loop
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nkt = C >> 6;
  const int j = t & 7, row0 = t >> 3;

  // this lane's query (column lane&15 of the wave's 16-query tile: rows 2*wave, 2*wave+1 of the 8x8 tile)
  const int ql = wave * 16 + lr;
  const int qy = qy0 + (ql >> 3), qx = qx0 + (ql & 7);

  // loader: query rows (fixed for the whole kernel)
  size_t qoff[2];
  bool qok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = row0 + 32 * i;
    const int y = qy0 + (rr >> 3), x = qx0 + (rr & 7);
    qok[i] = (y < H) && (x < W);
    qoff[i] = ((size_t)a.qframe * HW + (size_t)(qok[i] ? y * W + x : 0)) * C;
  }

  float tv[LP_TOPK];
  int ti[LP_TOPK];
#pragma unroll
  for (int i = 0; i < LP_TOPK; ++i) { tv[i] = -INFINITY; ti[i] = -1; }

  // key frames are split over blockIdx.y (a DAVIS frame has only 8x14 query tiles: one workgroup
  // per tile would leave most of the 256 CUs idle); every split emits a partial top-k per query
  const int fpb = (a.nkeys + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * fpb, f_end = min(a.nkeys, f_begin + fpb);
  for (int f = f_begin; f < f_end; ++f) {
    const int slot = a.kslot[f];
    // the first non_mask_len key frames are not masked (local_attention.py:303-309: with_first_neighbor=False)
    const int r = f < a.non_mask_len ? 0 : a.radius;
    int wy0 = 0, wy1 = H - 1, wx0 = 0, wx1 = W - 1;
    if (r > 0) {
      wy0 = max(0, qy0 - (r - 1)); wy1 = min(H - 1, qy0 + 7 + (r - 1));
      wx0 = max(0, qx0 - (r - 1)); wx1 = min(W - 1, qx0 + 7 + (r - 1));
    }
    const int ww = wx1 - wx0 + 1, nwin = (wy1 - wy0 + 1) * ww;
    const int nkb = (nwin + BK - 1) / BK;
    for (int kb = 0; kb < nkb; ++kb) {
      // key rows of this block
      size_t koff[4];
      bool kok[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = kb * BK + row0 + 32 * i;
        kok[i] = kk < nwin;
        const int ky = wy0 + (kok[i] ? kk / ww : 0), kx = wx0 + (kok[i] ? kk % ww : 0);
        koff[i] = ((size_t)slot * HW + (size_t)(ky * W + kx)) * C;
        if (j == 0) sKC[row0 + 32 * i] = kok[i] ? ((ky << 16) | kx) : -1;
      }
      u32x4 kr[4], qr[2];
      auto load_tiles = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kr[i] = kok[i] ? ld16(a.fbank + koff[i] + kt * 64 + j * 8) : zero16();
#pragma unroll
        for (int i = 0; i < 2; ++i) qr[i] = qok[i] ? ld16(a.fbank + qoff[i] + kt * 64 + j * 8) : zero16();
      };
      auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st16(&sK[buf][lds_off(row0 + 32 * i, j)], kr[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) st16(&sQ[buf][lds_off(row0 + 32 * i, j)], qr[i]);
      };
      f32x4 acc[TM][1];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc[tm][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      load_tiles(0);
      store_tiles(0);
      __syncthreads();
      for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tiles(kt + 1);
        mma_kstep<TM, 1, false>(sK[cur], sQ[cur], 0, wave * 16, lane, acc);
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
      }
      // scores -> mask -> streaming top-k (lane: keys tm*16 + 4*lq + reg, query lane&15)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const u32x4 kc = *reinterpret_cast<const u32x4*>(&sKC[tm * 16 + lq * 4]);
        float sc[4];
        int id[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int pk = (int)kc[rg];
          const int ky = pk >> 16, kx = pk & 0xffff;
          bool ok = pk >= 0;
          if (r > 0) {
            const int dy = ky - qy, dx = kx - qx;
            ok = ok && (dy * dy + dx * dx < r * r);
          }
          sc[rg] = ok ? acc[tm][0][rg] * a.inv_temp : -INFINITY;
          id[rg] = f * HW + ky * W + kx;
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          if (__any(sc[rg] > tv[LP_TOPK - 1])) topk_insert(tv, ti, sc[rg], id[rg]);
        }
      }
      __syncthreads();   // sKC / tiles are rewritten by the next key block
    }
  }

  // merge the 4 partial lists of every query through LDS (aliases sK), then softmax + value gather
  float* mv = reinterpret_cast<float*>(&sK[0][0]);            // [64 queries][4*TOPK]
  int* mi = reinterpret_cast<int*>(mv + BQ * 4 * LP_TOPK);
#pragma unroll
  for (int i = 0; i < LP_TOPK; ++i) {
    mv[(ql * 4 + lq) * LP_TOPK + i] = tv[i];
    mi[(ql * 4 + lq) * LP_TOPK + i] = ti[i];
  }
  __syncthreads();
  if (t < BQ) {
    const int y = qy0 + (t >> 3), x = qx0 + (t & 7);
    if (y < H && x < W) {
      float* cv = mv + t * 4 * LP_TOPK;
      int* ci = mi + t * 4 * LP_TOPK;
      // top-k of this split's 4 x 10 candidates -> partial list [split][query][LP_TOPK]
      float* pv = a.pval + ((size_t)blockIdx.y * HW + (y * W + x)) * LP_TOPK;
      int* pi = a.pidx + ((size_t)blockIdx.y * HW + (y * W + x)) * LP_TOPK;
      for (int k = 0; k < LP_TOPK; ++k) {
        int best = 0;
        float bvv = cv[0];
        for (int c = 1; c < 4 * LP_TOPK; ++c)
          if (cv[c] > bvv) { bvv = cv[c]; best = c; }
        pv[k] = bvv; pi[k] = ci[best];
        cv[best] = -INFINITY;
      }
    }
  }
}

// merge the per-split partial lists, softmax over the top-k, weighted sum of the value logits
__global__ __launch_bounds__(256) void labelprop_merge_kernel(LabelPropArgs a, int nsplit) {
  const int HW = a.H * a.W;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= HW) return;
  const int K = a.topk < LP_TOPK ? a.topk : LP_TOPK;
  float bv[LP_TOPK];
  int bi[LP_TOPK];
#pragma unroll
  for (int k = 0; k < LP_TOPK; ++k) { bv[k] = -INFINITY; bi[k] = -1; }
  for (int sp = 0; sp < nsplit; ++sp) {
    const float* pv = a.pval + ((size_t)sp * HW + q) * LP_TOPK;
    const int* pi = a.pidx + ((size_t)sp * HW + q) * LP_TOPK;
    for (int c = 0; c < LP_TOPK; ++c) {
      const float v = pv[c];
      if (v > bv[LP_TOPK - 1]) topk_insert(bv, bi, v, pi[c]);
    }
  }
  const float m = bv[0];
  float wgt[LP_TOPK], z = 0.f;
#pragma unroll
  for (int k = 0; k < LP_TOPK; ++k) {
    wgt[k] = (k < K && bi[k] >= 0 && bv[k] > -INFINITY) ? expf(bv[k] - m) : 0.f;
    z += wgt[k];
  }
  const float iz = 1.0f / z;
  float* o = a.out + (size_t)q * a.CO;
  for (int c = 0; c < a.CO; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LP_TOPK; ++k) {
      if (wgt[k] > 0.f) {
        const int fr = bi[k] / HW, px = bi[k] - fr * HW;
        s += (wgt[k] * iz) * a.sbank[((size_t)a.kslot[fr] * HW + px) * a.CO + c];
      }
    }
    o[c] = s;
  }
}

int vfs_labelprop_launch(const LabelPropArgs& a, hipStream_t s) {
  if (a.C % 64) return vfs_set_error(VFS_ERR_SHAPE, "labelprop: C%64");
  if (a.nkeys < 1 || a.nkeys > LP_MAX_KEYS) return vfs_set_error(VFS_ERR_SHAPE, "labelprop: 1 <= nkeys <= 64");
  if (a.topk < 1 || a.topk > LP_TOPK) return vfs_set_error(VFS_ERR_SHAPE, "labelprop: 1 <= topk <= 10");
  if (a.H >= 65536 || a.W >= 65536) return vfs_set_error(VFS_ERR_SHAPE, "labelprop: H,W < 65536");
  const int tiles = ((a.H + 7) / 8) * ((a.W + 7) / 8);
  if (a.pval == nullptr || a.pidx == nullptr) return vfs_set_error(VFS_ERR_ARG, "labelprop: partial workspace missing");
  int nsplit = (768 + tiles - 1) / tiles;                 // ~3 workgroups per CU
  if (nsplit > a.nkeys) nsplit = a.nkeys;
  if (nsplit > LP_MAX_FSPLIT) nsplit = LP_MAX_FSPLIT;
  const int fpb = (a.nkeys + nsplit - 1) / nsplit;
  nsplit = (a.nkeys + fpb - 1) / fpb;
  hipLaunchKernelGGL(labelprop_kernel, dim3(tiles, nsplit), dim3(256), 0, s, a);
  int rc = vfs_check_launch("labelprop");
  if (rc) return rc;
  hipLaunchKernelGGL(labelprop_merge_kernel, dim3((a.H * a.W + 255) / 256), dim3(256), 0, s, a, nsplit);
  return vfs_check_launch("labelprop_merge");
}

// ---------------------------------------------------------------------------------------------
// post-processing of one propagated frame (vanilla_tracker.py:162-181):
//   up = F.interpolate(seg[HW][CO] -> [Ho][Wo], bilinear, align_corners=False)
//   per channel: (up - min)/(max - min + 1e-12) where max > 0
//   label = argmax over channels (first maximum) -> uint8
// pass 1: per-channel min / max of the upsampled map (one partial per workgroup, fixed order)
__device__ __forceinline__ float bilerp(const float* __restrict__ seg, int H, int W, int CO, int c, int oy, int ox,
                                        float sy, float sx) {
  float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v00 = seg[((size_t)y0 * W + x0) * CO + c], v01 = seg[((size_t)y0 * W + x1) * CO + c];
  const float v10 = seg[((size_t)y1 * W + x0) * CO + c], v11 = seg[((size_t)y1 * W + x1) * CO + c];
  return hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// grid (LP_POST_BLOCKS, CO): one class per workgroup (as seg_minmax_exact_kernel, csrc/exact_f32.hip)
__global__ __launch_bounds__(256) void seg_minmax_kernel(const float* __restrict__ seg, float* __restrict__ partial, int H, int W,
                                                         int CO, int Ho, int Wo) {
  __shared__ float smn[256], smx[256];
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  const int total = Ho * Wo, c = blockIdx.y;
  const int stride = gridDim.x * 256, dy = stride / Wo, dx = stride - dy * Wo;
  float mn = INFINITY, mx = -INFINITY;
  int p = blockIdx.x * 256 + threadIdx.x, oy = p / Wo, ox = p - oy * Wo;
  for (; p < total; p += stride) {
    const float v = bilerp(seg, H, W, CO, c, oy, ox, sy, sx);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
    oy += dy; ox += dx;
    if (ox >= Wo) { ox -= Wo; ++oy; }
  }
  smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]);
      smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((size_t)blockIdx.x * CO + c) * 2] = smn[0];
    partial[((size_t)blockIdx.x * CO + c) * 2 + 1] = smx[0];
  }
}

__global__ __launch_bounds__(256) void seg_argmax_kernel(const float* __restrict__ seg, const float* __restrict__ partial,
                                                         int nblk, uint8_t* __restrict__ label, int H, int W, int CO, int Ho,
                                                         int Wo) {
  __shared__ float smn[LP_MAX_CLASSES], smx[LP_MAX_CLASSES];
  if ((int)threadIdx.x < CO) {
    float mn = INFINITY, mx = -INFINITY;
    for (int b = 0; b < nblk; ++b) {
      mn = fminf(mn, partial[((size_t)b * CO + threadIdx.x) * 2]);
      mx = fmaxf(mx, partial[((size_t)b * CO + threadIdx.x) * 2 + 1]);
    }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
  }
  __syncthreads();
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  const int total = Ho * Wo;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
    const int oy = p / Wo, ox = p % Wo;
    float best = -INFINITY;
    int bc = 0;
    for (int c = 0; c < CO; ++c) {
      float v = bilerp(seg, H, W, CO, c, oy, ox, sy, sx);
      if (smx[c] > 0.f) v = (v - smn[c]) / (smx[c] - smn[c] + 1e-12f);
      if (v > best) { best = v; bc = c; }
    }
    label[p] = (uint8_t)bc;
  }
}

int vfs_seg_postprocess_launch(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                               hipStream_t s) {
  if (CO < 1 || CO > LP_MAX_CLASSES) return vfs_set_error(VFS_ERR_SHAPE, "seg_postprocess: 1 <= classes <= 256");
  const int nblk = LP_POST_BLOCKS;
  hipLaunchKernelGGL(seg_minmax_kernel, dim3(nblk, CO), dim3(256), 0, s, seg, partial, H, W, CO, Ho, Wo);
  int rc = vfs_check_launch("seg_minmax");
  if (rc) return rc;
  int blocks = (Ho * Wo + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(seg_argmax_kernel, dim3(blocks), dim3(256), 0, s, seg, partial, nblk, label, H, W, CO, Ho, Wo);
  return vfs_check_launch("seg_argmax");
}

// one-hot of a uint8 label map into the fp32 seg bank (vanilla_tracker.py:96-100)
__global__ __launch_bounds__(256) void onehot_kernel(const uint8_t* __restrict__ lab, float* __restrict__ out, int P, int CO) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P * CO) return;
  out[i] = (lab[i / CO] == (i % CO)) ? 1.f : 0.f;
}
int vfs_onehot_launch(const uint8_t* lab, float* out, int P, int CO, hipStream_t s) {
  hipLaunchKernelGGL(onehot_kernel, dim3((P * CO + 255) / 256), dim3(256), 0, s, lab, out, P, CO);
  return vfs_check_launch("onehot");
}

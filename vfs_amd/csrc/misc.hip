// Layout conversion, weight packing, pooling, cosine loss and SGD kernels (HBM / latency bound).
#include "vfs_ops.h"

// ------------------------------------------------------------------ input frames
// reference layout: imgs[b][v][c][t][h][w] fp32 (pipelines/formating.py:248-258), view v is
// video2images(imgs[:, v]) = frames ordered (b, t)  (common/utils.py:45-53)
__global__ __launch_bounds__(256) void imgs_to_nhwc4_kernel(const float* __restrict__ imgs, bf16_t* __restrict__ out,
                                                            int B, int V, int T, int H, int W, int Wp) {
  const long long total = (long long)V * B * T * H * Wp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long p = i;
    const int w = (int)(p % Wp); p /= Wp;
    const int h = (int)(p % H); p /= H;
    const int t = (int)(p % T); p /= T;
    const int b = (int)(p % B);
    const int v = (int)(p / B);
    u32x2 pk = {0u, 0u};
    if (w < W) {
      const size_t plane = (size_t)T * H * W;
      const size_t base = (((size_t)b * V + v) * 3) * plane + ((size_t)t * H + h) * W + w;
      pk.x = pack2bf(imgs[base], imgs[base + plane]);
      pk.y = pack2bf(imgs[base + 2 * plane], 0.f);
    }
    st8(out + (size_t)i * 4, pk);
  }
}
int vfs_imgs_to_nhwc4_launch(const float* imgs, bf16_t* out, int B, int V, int T, int H, int W, int Wp, hipStream_t s) {
  long long total = (long long)V * B * T * H * Wp;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(imgs_to_nhwc4_kernel, dim3((int)blocks), dim3(256), 0, s, imgs, out, B, V, T, H, W, Wp);
  return vfs_check_launch("imgs_to_nhwc4");
}

// ------------------------------------------------------------------ weight packing
// fp32 master weights [Cout][Cin][T] (T = KH*KW) -> bf16 wf [Cout][T][Cin] (forward operand) and
// wd [Cin][T][Cout] (dgrad operand), every step after the optimizer.  A workgroup transposes one
// TC x TC (cout x cin) tile through LDS: whole contiguous row segments in (coalesced), 2 packed
// elements (4 bytes) per lane out along cin for wf and along cout for wd.  T is 1 or 9 for every
// layer but the stem (kind 1: [64][3][7][7] -> [64][8][8][4], a few thousand elements).
#define PACK_TC 32
template <int T>
__device__ __forceinline__ void pack_tile(const PackDesc& d, int tile, bf16_t* sT, int KT) {
  constexpr int TC = PACK_TC;
  const int Tn = T > 0 ? T : KT;                  // taps (compile-time for the common cases)
  const int pitch = TC * Tn + 2;                  // bf16 elements per staged cout row (odd dword count)
  const int t = threadIdx.x;
  const int tiles_ci = (d.Cin + TC - 1) / TC;
  const int co0 = (tile / tiles_ci) * TC, ci0 = (tile - (tile / tiles_ci) * tiles_ci) * TC;
  const int ncout = min(TC, d.Cout - co0), ncin = min(TC, d.Cin - ci0);
  const int rowlen = ncin * Tn;
  // in: thread (row r = t / 32 + 8 i, k = t % 32 + 32 j) over the contiguous segment of each cout row
  for (int r = t >> 5; r < ncout; r += 8) {
    const float* src = d.w + ((size_t)(co0 + r) * d.Cin + ci0) * Tn;
    for (int k = t & 31; k < rowlen; k += 32) sT[r * pitch + k] = f2bf(src[k]);
  }
  __syncthreads();
  // wf[cout][tap][cin]: lane pair index = cin pair
  const int hp = (ncin + 1) >> 1;
  for (int q = t / 16; q < ncout * Tn; q += 16) {
    const int r = q / Tn, tap = q - r * Tn;
    const int cp = t & 15;
    if (cp < hp) {
      const int c = cp * 2;
      const bf16_t lo = sT[r * pitch + c * Tn + tap];
      bf16_t* dst = d.wf + ((size_t)(co0 + r) * Tn + tap) * d.Cin + ci0 + c;
      if (c + 1 < ncin && ((d.Cin | ci0) & 1) == 0) {
        const bf16_t hi = sT[r * pitch + (c + 1) * Tn + tap];
        *reinterpret_cast<uint32_t*>(dst) = (uint32_t)lo | ((uint32_t)hi << 16);
      } else {
        dst[0] = lo;
        if (c + 1 < ncin) dst[1] = sT[r * pitch + (c + 1) * Tn + tap];
      }
    }
  }
  if (d.wd) {   // wd[cin][tap][cout]: lane pair index = cout pair
    const int hq = (ncout + 1) >> 1;
    for (int q = t / 16; q < ncin * Tn; q += 16) {
      const int ci = q / Tn, tap = q - ci * Tn;
      const int rp = t & 15;
      if (rp < hq) {
        const int r = rp * 2;
        const bf16_t lo = sT[r * pitch + ci * Tn + tap];
        bf16_t* dst = d.wd + ((size_t)(ci0 + ci) * Tn + tap) * d.Cout + co0 + r;
        if (r + 1 < ncout && ((d.Cout | co0) & 1) == 0) {
          const bf16_t hi = sT[(r + 1) * pitch + ci * Tn + tap];
          *reinterpret_cast<uint32_t*>(dst) = (uint32_t)lo | ((uint32_t)hi << 16);
        } else {
          dst[0] = lo;
          if (r + 1 < ncout) dst[1] = sT[(r + 1) * pitch + ci * Tn + tap];
        }
      }
    }
  }
}


// Round 5: the two shapes that hold all but a few thousand of a ResNet's weights, with 16-byte global accesses and whole tiles in
// flight (the generic tile below moves 4 bytes per lane in two dependent phases: 1.9 TB/s over 305 MB, 164 us at the start of
// every ResNet-50 step).  T = 1, 64 | Cout, 64 | Cin: a 64 x 64 tile - wf is the converted row itself (8 bytes per lane straight
// from the registers), wd the transpose through LDS (a lane gathers 8 couts of one cin: 16-byte stores, 128-byte row segments).
// T = 9, 32 | Cout, 64 | Cin: a 32 x 64 x 9 tile - rows of 576 contiguous floats in, wf [cout][tap][64 cin] and wd [cin][tap][32 cout]
// gathered from the staged tile with 16-byte stores.
typedef __attribute__((ext_vector_type(2))) uint32_t pk_u32x2;
__device__ __forceinline__ void pack_tile_1x1_64(const PackDesc& d, int tile, bf16_t* sT) {
  constexpr int P = 66;                             // bf16 per staged row: 33 dwords (odd)
  const int t = threadIdx.x;
  const int tiles_ci = d.Cin >> 6;
  const int co0 = (tile / tiles_ci) << 6, ci0 = (tile % tiles_ci) << 6;
  const int r0 = t >> 4, c4 = (t & 15) << 2;
  f32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(d.w + (size_t)(co0 + r0 + 16 * i) * d.Cin + ci0 + c4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 16 * i;
    const uint32_t lo = pack2bf(v[i][0], v[i][1]), hi = pack2bf(v[i][2], v[i][3]);
    *reinterpret_cast<pk_u32x2*>(d.wf + (size_t)(co0 + r) * d.Cin + ci0 + c4) = (pk_u32x2){lo, hi};
    *reinterpret_cast<uint32_t*>(&sT[r * P + c4]) = lo;
    *reinterpret_cast<uint32_t*>(&sT[r * P + c4 + 2]) = hi;
  }
  if (!d.wd) return;
  __syncthreads();
  const int co8 = (t & 7) << 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = (t >> 3) + 32 * i;
    uint32_t w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = (uint32_t)sT[(co8 + 2 * j) * P + ci] | ((uint32_t)sT[(co8 + 2 * j + 1) * P + ci] << 16);
    *reinterpret_cast<u32x4*>(d.wd + (size_t)(ci0 + ci) * d.Cout + co0 + co8) = (u32x4){w4[0], w4[1], w4[2], w4[3]};
  }
}
__device__ __forceinline__ void pack_tile_3x3_32x64(const PackDesc& d, int tile, bf16_t* sT) {
  constexpr int RL = 64 * 9, P = RL + 2;            // staged row: 576 elements + pad (289 dwords, odd)
  const int t = threadIdx.x;
  const int tiles_ci = d.Cin >> 6;
  const int co0 = (tile / tiles_ci) << 5, ci0 = (tile % tiles_ci) << 6;
  // in: 32 rows x 144 float4; lane task q = t + 256 i -> row q / 144, float4 q % 144
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int q = t + 256 * (9 * half + i), r = q / 144, k4 = q - r * 144;
      v[i] = *reinterpret_cast<const f32x4*>(d.w + ((size_t)(co0 + r) * d.Cin + ci0) * 9 + 4 * k4);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int q = t + 256 * (9 * half + i), r = q / 144, k4 = q - r * 144;
      *reinterpret_cast<uint32_t*>(&sT[r * P + 4 * k4]) = pack2bf(v[i][0], v[i][1]);
      *reinterpret_cast<uint32_t*>(&sT[r * P + 4 * k4 + 2]) = pack2bf(v[i][2], v[i][3]);
    }
  }
  __syncthreads();
  // wf[cout][tap][cin]: 288 (cout, tap) rows of 64 cin = 8 lanes x 8 cin
  for (int q = t; q < 288 * 8; q += 256) {
    const int row = q >> 3, c8 = (q & 7) << 3, r = row / 9, tap = row - r * 9;
    const bf16_t* src = sT + r * P + c8 * 9 + tap;
    uint32_t w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = (uint32_t)src[(2 * j) * 9] | ((uint32_t)src[(2 * j + 1) * 9] << 16);
    *reinterpret_cast<u32x4*>(d.wf + ((size_t)(co0 + r) * 9 + tap) * d.Cin + ci0 + c8) = (u32x4){w4[0], w4[1], w4[2], w4[3]};
  }
  if (!d.wd) return;
  // wd[cin][tap][cout]: 576 (cin, tap) rows of 32 cout = 4 lanes x 8 cout
  for (int q = t; q < 576 * 4; q += 256) {
    const int row = q >> 2, co8 = (q & 3) << 3;       // row = ci * 9 + tap = the staged column
    const bf16_t* src = sT + co8 * P + row;
    uint32_t w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = (uint32_t)src[(2 * j) * P] | ((uint32_t)src[(2 * j + 1) * P] << 16);
    const int ci = row / 9, tap = row - ci * 9;
    *reinterpret_cast<u32x4*>(d.wd + ((size_t)(ci0 + ci) * 9 + tap) * d.Cout + co0 + co8) = (u32x4){w4[0], w4[1], w4[2], w4[3]};
  }
}

#define PACK_MAX_T 25
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackDesc* __restrict__ table, int n) {
  __shared__ __attribute__((aligned(16))) bf16_t sT[PACK_TC * (PACK_TC * PACK_MAX_T + 2)];
  // uniform: which tensor does this workgroup's tile belong to
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = table[lo];
  const int tile = (int)blockIdx.x - d.tile_start;
  if (d.kind == 1) {   // stem, 256 elements per workgroup
    const long long total = (long long)d.Cout * d.Cin * d.KH * d.KW;
    long long e = (long long)tile * 256 + threadIdx.x;
    if (e < total) {
      const bf16_t v = f2bf(d.w[e]);
      const int s = (int)(e % d.KW); e /= d.KW;
      const int r = (int)(e % d.KH); e /= d.KH;
      const int cin = (int)(e % d.Cin);
      const int cout = (int)(e / d.Cin);
      d.wf[(((size_t)cout * 8 + r) * 8 + (s + 1)) * 4 + cin] = v;
    }
    return;
  }
  const int KT = d.KH * d.KW;
  if (d.kind == 2) { pack_tile_1x1_64(d, tile, sT); return; }
  if (d.kind == 3) { pack_tile_3x3_32x64(d, tile, sT); return; }
  if (KT == 1) pack_tile<1>(d, tile, sT, 1);
  else if (KT == 9) pack_tile<9>(d, tile, sT, 9);
  else pack_tile<0>(d, tile, sT, KT);
}
int vfs_pack_weights_launch(const PackDesc* table, int ntensors, long long total_tiles, hipStream_t s) {
  if (total_tiles <= 0 || ntensors <= 0) return VFS_OK;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, table, ntensors);
  return vfs_check_launch("pack_weights");
}

// ------------------------------------------------------------------ global average pool
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int HW,
                                                          int C) {
  const int cv = C >> 3;
  const int total = N * cv;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < total; v += gridDim.x * 256) {
    const int n = v / cv, c = (v - n * cv) * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int p = 0; p < HW; ++p) {       // eight independent loads in flight (the rolled loop waited for each pixel in turn)
      float f[8];
      unpack8(ld16(x + ((size_t)n * HW + p) * C + c), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += f[i];
    }
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] *= inv;
    st16(y + (size_t)n * C + c, pack8(acc));
  }
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16_t* __restrict__ g, bf16_t* __restrict__ gx, int N, int HW,
                                                          int C) {
  const int cv = C >> 3;
  const long long total = (long long)N * HW * cv;
  const float inv = 1.0f / (float)HW;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
    const long long p = v / cv;
    const int c = (int)(v - p * cv) * 8;
    const int n = (int)(p / HW);
    float f[8];
    unpack8(ld16(g + (size_t)n * C + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= inv;
    st16(gx + (size_t)p * C + c, pack8(f));
  }
}
int vfs_avgpool_fwd_launch(const bf16_t* x, bf16_t* y, int N, int HW, int C, hipStream_t s) {
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((N * (C >> 3) + 255) / 256), dim3(256), 0, s, x, y, N, HW, C);
  return vfs_check_launch("avgpool_fwd");
}
int vfs_avgpool_bwd_launch(const bf16_t* g, bf16_t* gx, int N, int HW, int C, hipStream_t s) {
  long long b = ((long long)N * HW * (C >> 3) + 255) / 256;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((int)(b > 4096 ? 4096 : b)), dim3(256), 0, s, g, gx, N, HW, C);
  return vfs_check_launch("avgpool_bwd");
}

// db[c] += sum_m dy[m][c]   (64 channels x 4 row-slices per workgroup, fixed-order combine)
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ dy, float* __restrict__ db, int M, int C) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < C)
    for (int m = sl; m < M; m += 4) s += bf2f(dy[(size_t)m * C + c]);
  sh[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) db[c] += (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
int vfs_bias_grad_launch(const bf16_t* dy, float* db, int M, int C, hipStream_t s) {
  hipLaunchKernelGGL(bias_grad_kernel, dim3((C + 63) / 64), dim3(256), 0, s, dy, db, M, C);
  return vfs_check_launch("bias_grad");
}

// ------------------------------------------------------------------ cosine loss
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ int roll_src(int i, int T, int k) {  // index of roll(k)[i] within i's video
  const int b = i / T, t = i - b * T;
  int u = t - k;
  if (u < 0) u += T;
  return b * T + u;
}
// one wave per (k, i):  loss[k][i] = w * (0.5*L(p1[i], z2[j]) + 0.5*L(p2[j], z1[i])),  j = roll_k(i)
__global__ __launch_bounds__(64) void cosine_loss_fwd_kernel(LossArgs a) {
  const int i = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
  const int j = roll_src(i, a.T, k);
  float d1 = 0.f, d2 = 0.f, np1 = 0.f, nz2 = 0.f, np2 = 0.f, nz1 = 0.f;
#pragma unroll 8
  for (int c = lane; c < a.C; c += 64) {
    const float p1 = bf2f(a.p1[(size_t)i * a.C + c]), z1 = bf2f(a.z1[(size_t)i * a.C + c]);
    const float p2 = bf2f(a.p2[(size_t)j * a.C + c]), z2 = bf2f(a.z2[(size_t)j * a.C + c]);
    d1 += p1 * z2; d2 += p2 * z1;
    np1 += p1 * p1; nz2 += z2 * z2; np2 += p2 * p2; nz1 += z1 * z1;
  }
  d1 = wave_sum(d1); d2 = wave_sum(d2);
  np1 = wave_sum(np1); nz2 = wave_sum(nz2); np2 = wave_sum(np2); nz1 = wave_sum(nz1);
  if (lane == 0) {
    const float eps = 1e-12f;
    const float c1 = d1 / (fmaxf(sqrtf(np1), eps) * fmaxf(sqrtf(nz2), eps));
    const float c2 = d2 / (fmaxf(sqrtf(np2), eps) * fmaxf(sqrtf(nz1), eps));
    const float l1 = a.negative ? -c1 : 2.f - 2.f * c1;
    const float l2 = a.negative ? -c2 : 2.f - 2.f * c2;
    a.loss[(size_t)k * a.N + i] = (0.5f * l1 + 0.5f * l2) * a.weight;
  }
}
// one wave per (view, i): gradient wrt p (z is detached in the reference)
//   view 0: dp1[i] = sum_k gloss[k][i]      * w/2 * dL/da (a = p1[i], b = z2[roll_k(i)])
//   view 1: dp2[j] = sum_k gloss[k][inv_k(j)] * w/2 * dL/da (a = p2[j], b = z1[inv_k(j)])
//   dL/da = coef * (bhat - cos * ahat) / max(|a|, eps),  coef = -2 (or -1 when negative)
__global__ __launch_bounds__(64) void cosine_loss_bwd_kernel(LossArgs a) {
  const int i = blockIdx.x, view = blockIdx.y, lane = threadIdx.x;
  const bf16_t* A = view == 0 ? a.p1 : a.p2;
  const bf16_t* Bz = view == 0 ? a.z2 : a.z1;
  bf16_t* out = view == 0 ? a.dp1 : a.dp2;
  const float eps = 1e-12f, coef = a.negative ? -1.f : -2.f;
  float na = 0.f;
#pragma unroll 8
  for (int c = lane; c < a.C; c += 64) { const float v = bf2f(A[(size_t)i * a.C + c]); na += v * v; }
  na = fmaxf(sqrtf(wave_sum(na)), eps);
  const int bvid = i / a.T, t = i - bvid * a.T;
  constexpr int MAXC = 32;  // supports C <= 2048
  float acc[MAXC];
#pragma unroll
  for (int q = 0; q < MAXC; ++q) acc[q] = 0.f;
  for (int k = 0; k < a.K; ++k) {
    int u, gi;
    if (view == 0) { u = t - k; if (u < 0) u += a.T; gi = i; }        // partner = roll_k(i), loss index i
    else { u = t + k; if (u >= a.T) u -= a.T; gi = bvid * a.T + u; }  // p2[i] is paired with loss index inv_k(i)
    const int j = bvid * a.T + u;
    float nb = 0.f, dot = 0.f;
#pragma unroll 8
    for (int c = lane; c < a.C; c += 64) {
      const float av = bf2f(A[(size_t)i * a.C + c]), bv = bf2f(Bz[(size_t)j * a.C + c]);
      nb += bv * bv; dot += av * bv;
    }
    nb = fmaxf(sqrtf(wave_sum(nb)), eps);
    const float cs = wave_sum(dot) / (na * nb);
    const float gs = a.gloss[(size_t)k * a.N + gi] * a.weight * 0.5f * coef / na;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {       // loads without a per-lane condition (index clamped, the update is masked): they batch
      const int c = lane + q * 64, cc = c < a.C ? c : lane;
      const float av = bf2f(A[(size_t)i * a.C + cc]), bv = bf2f(Bz[(size_t)j * a.C + cc]);
      const float upd = gs * (bv / nb - cs * av / na);
      acc[q] += c < a.C ? upd : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int c = lane + q * 64;
    if (c < a.C) out[(size_t)i * a.C + c] = f2bf(acc[q]);
  }
}
// _parse_losses (trackers/base.py:76-110) for the K unreduced loss rows of one step: means[k] = mean_i loss[k][i],
// means[K] = their sum ('loss').  One wave, fixed summation order, double accumulation.
__global__ __launch_bounds__(64) void loss_means_kernel(const float* __restrict__ loss, float* __restrict__ means, int K, int N) {
  const int lane = threadIdx.x;
  double total = 0.0;
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    for (int i = lane; i < N; i += 64) acc += (double)loss[(size_t)k * N + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    const float m = (float)(acc / (double)N);
    if (lane == 0) means[k] = m;
    total += (double)m;
  }
  if (lane == 0) means[K] = (float)total;
}
int vfs_loss_means_launch(const float* loss, float* means, int K, int N, hipStream_t s) {
  if (K <= 0 || N <= 0) return vfs_set_error(VFS_ERR_SHAPE, "loss_means: empty");
  hipLaunchKernelGGL(loss_means_kernel, dim3(1), dim3(64), 0, s, loss, means, K, N);
  return vfs_check_launch("loss_means");
}
int vfs_cosine_loss_fwd_launch(const LossArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(cosine_loss_fwd_kernel, dim3(a.N, a.K), dim3(64), 0, s, a);
  return vfs_check_launch("cosine_loss_fwd");
}
int vfs_cosine_loss_bwd_launch(const LossArgs& a, hipStream_t s) {
  if (a.C > 2048) return vfs_set_error(VFS_ERR_SHAPE, "cosine_loss_bwd: C > 2048");
  hipLaunchKernelGGL(cosine_loss_bwd_kernel, dim3(a.N, 2), dim3(64), 0, s, a);
  return vfs_check_launch("cosine_loss_bwd");
}

// ------------------------------------------------------------------ SGD
// torch.optim.SGD (configs/*:134): g += wd*p ; buf = momentum*buf + g ; p -= lr*buf
// (a zero-initialised buf reproduces torch's "buf = g" on the first step)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long long n, float lr, float momentum, float wd,
                                                  const unsigned long long* __restrict__ skip) {
  if (skip && *skip) return;      // a poisoned step (failed SyncBN exchange, vfs_p2p.h) leaves weights and momentum alone
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 bv = reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float gg = gv[q] + wd * pv[q];
      bv[q] = momentum * bv[q] + gg;
      pv[q] -= lr * bv[q];
    }
    reinterpret_cast<f32x4*>(buf)[i] = bv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] + wd * p[i];
    buf[i] = momentum * buf[i] + gg;
    p[i] -= lr * buf[i];
  }
}
int vfs_sgd_launch(float* p, const float* g, float* buf, long long n, float lr, float momentum, float wd, const unsigned long long* skip,
                   hipStream_t s) {
  long long b = ((n >> 2) + 255) / 256;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0, s, p, g, buf, n, lr, momentum, wd, skip);
  return vfs_check_launch("sgd");
}
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ p, long long n, float scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] *= scale;
}
int vfs_scale_launch(float* p, long long n, float scale, hipStream_t s) {
  long long b = (n + 255) / 256;
  hipLaunchKernelGGL(scale_kernel, dim3((int)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0, s, p, n, scale);
  return vfs_check_launch("scale");
}

// bf16 gradient buckets for the data-parallel all-reduce (opt-in: halves the xGMI traffic of the 152.8 MB ResNet-50 gradient;
// the reference's DDP reduces fp32): dst = bf16(src * scale) before the collective, dst = float(src) after it
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n, float scale) {
  const long long n8 = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    const float f[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale};
    st16(dst + i * 8, pack8(f));
  }
  if (blockIdx.x == 0)
    for (long long i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = f2bf(src[i] * scale);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long n8 = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float f[8];
    unpack8(ld16(src + i * 8), f);
    *reinterpret_cast<f32x4*>(dst + i * 8) = (f32x4){f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = (f32x4){f[4], f[5], f[6], f[7]};
  }
  if (blockIdx.x == 0)
    for (long long i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = bf2f(src[i]);
}
int vfs_f32_to_bf16_launch(const float* src, bf16_t* dst, long long n, float scale, hipStream_t s) {
  if (((size_t)src & 15) || ((size_t)dst & 15)) return vfs_set_error(VFS_ERR_ARG, "f32_to_bf16: 16-byte aligned buffers");
  long long b = ((n >> 3) + 255) / 256;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((int)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0, s, src, dst, n, scale);
  return vfs_check_launch("f32_to_bf16");
}
int vfs_bf16_to_f32_launch(const bf16_t* src, float* dst, long long n, hipStream_t s) {
  if (((size_t)src & 15) || ((size_t)dst & 15)) return vfs_set_error(VFS_ERR_ARG, "bf16_to_f32: 16-byte aligned buffers");
  long long b = ((n >> 3) + 255) / 256;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((int)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0, s, src, dst, n);
  return vfs_check_launch("bf16_to_f32");
}

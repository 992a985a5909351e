// The fp32 EVALUATION path for gfx950 ("exact" precision): ResNet in eval mode, feature bank, label propagation and
// post-processing with fp32 storage and fp32 arithmetic whose results are DEFINED to the last bit, so that integer
// outputs (label maps) can be held to the reference's fp32 path and to the C oracle (oracle/exact_oracle.c) exactly.
//
// Replaces, for VanillaTracker.forward_test (mmaction/models/trackers/vanilla_tracker.py:80-206):
//   mmcv ConvModule in eval mode (conv -> BN(running stats) -> ReLU), resnet.py:15-232,555-575  -> conv_f32_kernel
//   nn.MaxPool2d(3,2,1), resnet.py:435                                                           -> maxpool_f32_kernel
//   F.normalize(dim=1), local_attention.py:277-279                                               -> l2norm_rows_f32_kernel
//   masked_attention_efficient + spatial_neighbor('circle'), local_attention.py:237-348          -> labelprop_f32_kernel (+merge)
//   F.interpolate(bilinear) / min-max / argmax, vanilla_tracker.py:162-181                       -> seg_*_exact_kernel
//
// Arithmetic contract (identical in the oracle):
//   * every dot product is ONE ascending fp32 chain acc = fma(a_k, b_k, acc) from +0: that is what
//     v_mfma_f32_32x32x2_f32 computes (MI355X guide: bitwise a k-ordered fmaf chain; 64 FLOP/clk/SIMD = the fp32
//     vector peak, 157 TFLOP/s) -- zero-filled taps / padding add fma(0, w, acc) steps, exactly as the oracle does;
//   * everything else is a single correctly rounded fp32 operation: contraction is switched OFF for this file, the
//     fused steps are written as explicit fmaf();
//   * exp() of the softmax is the explicit polynomial vexp(); top-k ties go to the lowest candidate index.
#include "vfs_ops.h"

#pragma clang fp contract(off)

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x4 ldf4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stf4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 zerof4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution, fp32 NHWC, any kernel size / stride / padding / dilation, Cin % 4 == 0
// workgroup = 128 output pixels x 64 output channels, K in chunks of CONV_F32_BK (kh, kw, cin ascending);
// wave (w&1, w>>1) owns 64 pixels x 32 channels = two 32x32x2 MFMA tiles (A = pixels, B = channels)
// LDS image [k][row]: lane (i = l&31, kk = l>>5) of MFMA step s reads element [2s + kk][row0 + i]
#ifndef CONV_F32_BK
#define CONV_F32_BK 32     // channels per chunk: 32 MFMAs per wave between two barriers (16: DAVIS R50 5.21-5.22 vs 5.15-5.18 ms per frame, R18 1.45 vs 1.41)
#endif
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF32Args a) {
  constexpr int BM = 128, BN = 64, BK = CONV_F32_BK, G = BK / 4, RPP = 256 / G;   // float4 groups per row and chunk, rows per loader pass
  // LDS planes [k][row], PAD dwords of padding per plane: MFMA step s reads plane 2s + (lane >> 5), 32 consecutive dwords per
  // half-wave - conflict-free; a half-wave of the loaders' ds_write_b32 covers 32 / G rows x G float4 groups at plane 4 * group + e:
  // bank (4 * group * (rows + PAD) + row) mod 32 = 8 * group + row (BK 16, PAD 2) or 4 * group + row (BK 32, PAD 1) - 32 different
  constexpr int PAD = CONV_F32_BK == 16 ? 2 : 1;
  constexpr int PA = BM + PAD, PB = BN + PAD;
  __shared__ float sA[BK * PA];
  __shared__ float sB[BK * PB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long M = (long long)a.N * a.Ho * a.Wo;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int C4 = a.Cin >> 2;
  const int K4 = a.KH * a.KW * C4;           // float4 groups along K
  // loaders: G CONSECUTIVE LANES read the 4 * BK contiguous bytes one row has in a chunk (thread = row t / G, float4 group t % G):
  // a wave instruction touches 64 / G rows.  With one row per lane (64 different cache lines per instruction) the vector memory
  // path needed ~156 clk per wave instruction (tools/probe_vmem_rate.hip) and bounded the kernel at half the MFMA rate.
  const int lq = t % G;
  // A: pixels t / G + RPP * i
  constexpr int NA = BM / RPP, NB = BN / RPP;
  const int ap = t / G;
  bool a_ok[NA];
  int an[NA], iy0[NA], ix0[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const long long am = m0 + ap + RPP * i;
    a_ok[i] = am < M;
    an[i] = 0; iy0[i] = 0; ix0[i] = 0;
    if (a_ok[i]) {
      const int hw = a.Ho * a.Wo;
      an[i] = (int)(am / hw);
      const int rem = (int)(am - (long long)an[i] * hw);
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      iy0[i] = oy * a.stride - a.pad; ix0[i] = ox * a.stride - a.pad;
    }
  }
  // B: output channels t / G + RPP * i
  const int bc = t / G;
  bool b_ok[NB];
  const float* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    b_ok[i] = n0 + bc + RPP * i < a.Cout;
    wrow[i] = a.w + (size_t)(b_ok[i] ? n0 + bc + RPP * i : 0) * K4 * 4;
  }

  f32x4 ra[NA], rb[NB];
  auto load = [&](int chunk) {
    const int k4 = chunk * G + lq;
    const bool k_ok = k4 < K4;
    const int tap = k_ok ? k4 / C4 : 0, c = (k4 - tap * C4) * 4;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x4 v = zerof4();
      if (a_ok[i] && k_ok) {
        const int iy = iy0[i] + kh * a.dil, ix = ix0[i] + kw * a.dil;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = ldf4(a.x + (((size_t)an[i] * a.H + iy) * a.W + ix) * a.Cin + c);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = (b_ok[i] && k_ok) ? ldf4(wrow[i] + (size_t)k4 * 4) : zerof4();
  };
  auto store = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < NA; ++i) sA[(4 * lq + e) * PA + ap + RPP * i] = ra[i][e];
#pragma unroll
      for (int i = 0; i < NB; ++i) sB[(4 * lq + e) * PB + bc + RPP * i] = rb[i][e];
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int wp0 = (wave & 1) * 64, wc0 = (wave >> 1) * 32;
  const int li = lane & 31, lk = lane >> 5;
  const int nchunks = (K4 + G - 1) / G;
  load(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    store();
    __syncthreads();
    if (ch + 1 < nchunks) load(ch + 1);
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float b = sB[(2 * s + lk) * PB + wc0 + li];
      const float a0 = sA[(2 * s + lk) * PA + wp0 + li], a1 = sA[(2 * s + lk) * PA + wp0 + 32 + li];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
  const int co = n0 + wc0 + li;
  if (co < a.Cout) {
    const float sc = a.scale ? a.scale[co] : 1.f, sh = a.scale ? a.shift[co] : 0.f;
    // the 32 identity values of this lane are requested together (rows past M clamped): inside the per-row conditional each
    // one was a dependent memory round trip
    float rv[2][16];
    if (a.res) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long m = m0 + wp0 + pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          rv[pt][r] = a.res[(size_t)(m < M ? m : M - 1) * a.Cout + co];
        }
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m0 + wp0 + pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < M) {
          float v = acc[pt][r];
          if (a.scale) v = __builtin_fmaf(v, sc, sh);
          const size_t o = (size_t)m * a.Cout + co;
          if (a.res) v = v + rv[pt][r];
          if (a.relu) v = v > 0.f ? v : 0.f;
          a.y[o] = v;
        }
      }
  }
}

int vfs_conv_f32_launch(const ConvF32Args& a, hipStream_t s) {
  if (a.Cin % 4) return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: Cin % 4 (pad the 3-channel input to NHWC4)");
  if (a.N < 1 || a.Ho < 1 || a.Wo < 1 || a.Cout < 1 || a.stride < 1 || a.dil < 1) return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: geometry");
  if (a.Ho != (a.H + 2 * a.pad - a.dil * (a.KH - 1) - 1) / a.stride + 1 || a.Wo != (a.W + 2 * a.pad - a.dil * (a.KW - 1) - 1) / a.stride + 1)
    return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: output size does not match (H + 2 pad - dil (K - 1) - 1) / stride + 1");
  if (a.scale && !a.shift) return vfs_set_error(VFS_ERR_ARG, "conv_f32: scale without shift");
  const long long M = (long long)a.N * a.Ho * a.Wo;
  hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)((M + 127) / 128), (unsigned)((a.Cout + 63) / 64)), dim3(256), 0, s, a);
  return vfs_check_launch("conv_f32");
}

// ---------------------------------------------------------------------------------------------
// imgs fp32 [B][V][3][T][H][W] -> fp32 NHWC4 [(v*B+b)*T+t][h][w][4] (channel 3 = 0)
__global__ __launch_bounds__(256) void imgs_to_nhwc4_f32_kernel(const float* __restrict__ imgs, float* __restrict__ out, int B, int V,
                                                                int T, int H, int W) {
  const long long total = (long long)B * V * T * H * W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int w = (int)(i % W);
  long long r = i / W;
  const int h = (int)(r % H); r /= H;
  const int tt = (int)(r % T); r /= T;
  const int b = (int)(r % B);
  const int v = (int)(r / B);
  const size_t plane = (size_t)T * H * W;
  const float* src = imgs + (((size_t)b * V + v) * 3) * plane + ((size_t)tt * H + h) * W + w;
  stf4(out + (size_t)i * 4, (f32x4){src[0], src[plane], src[2 * plane], 0.f});
}
int vfs_imgs_to_nhwc4_f32_launch(const float* imgs, float* out, int B, int V, int T, int H, int W, hipStream_t s) {
  const long long total = (long long)B * V * T * H * W;
  hipLaunchKernelGGL(imgs_to_nhwc4_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, imgs, out, B, V, T, H, W);
  return vfs_check_launch("imgs_to_nhwc4_f32");
}

// nn.MaxPool2d(3, 2, 1), NHWC fp32, C % 4 == 0
__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                                          int Ho, int Wo) {
  const int C4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * C4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4) * 4;
  long long r = i / C4;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oy = (int)(r % Ho);
  const int n = (int)(r / Ho);
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const f32x4 v = ldf4(x + (((size_t)n * H + iy) * W + ix) * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
    }
  stf4(y + (size_t)i * 4, m);
}
int vfs_maxpool_f32_launch(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
  if (C % 4) return vfs_set_error(VFS_ERR_SHAPE, "maxpool_f32: C % 4");
  if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return vfs_set_error(VFS_ERR_SHAPE, "maxpool_f32: output size");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, N, H, W, C, Ho, Wo);
  return vfs_check_launch("maxpool_f32");
}

// F.normalize(p=2, dim=channel, eps=1e-12) of rows [P][C]; one wave per row: lane l owns the float4 groups
// l, l+64, ... (one fmaf chain), butterfly over the lanes, y = x / max(sqrt(ss), eps)
__global__ __launch_bounds__(256) void l2norm_rows_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long P, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  const float* src = x + (size_t)row * C;
  float ss = 0.f;
  for (int g = lane; g * 4 < C; g += 64) {
    const f32x4 v = ldf4(src + g * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) ss = __builtin_fmaf(v[e], v[e], ss);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) ss = ss + __shfl_xor(ss, d);
  float nrm = sqrtf(ss);
  nrm = nrm > 1e-12f ? nrm : 1e-12f;
  for (int g = lane; g * 4 < C; g += 64) {
    f32x4 v = ldf4(src + g * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] / nrm;
    stf4(y + (size_t)row * C + g * 4, v);
  }
}
int vfs_l2norm_rows_f32_launch(const float* x, float* y, long long P, int C, hipStream_t s) {
  if (C % 4) return vfs_set_error(VFS_ERR_SHAPE, "l2norm_f32: C % 4");
  hipLaunchKernelGGL(l2norm_rows_f32_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, s, x, y, P, C);
  return vfs_check_launch("l2norm_rows_f32");
}

#include "vfs_lpx.h"

// one workgroup = an 8x8 tile of queries x the key frames [f_begin, f_end) of its split; per key frame only the
// (8 + 2(r-1))^2 window that can lie inside the circle, 64 keys per block, channels in stages of 32.
// wave (w&1, w>>1) = 32 keys x 32 queries: ONE 32x32x2 MFMA tile (A = keys, B = queries), so a lane owns
// 16 keys of one query; the four partial top-10 lists of a query (2 key halves x 2 lane halves) merge through LDS.
// RAGGED = C % 32 != 0: the last channel stage is zero-filled lane by lane.  Otherwise every load is UNCONDITIONAL - rows past the
// window / the map read a clamped (valid) row and their scores are masked below: the exec-mask branches and zero fills around
// predicated loads cost 12-16 % of the stage loop (tools/probe_lp_stage.hip, V0 vs V2 at equal occupancy)
template <bool RAGGED>
__global__ __launch_bounds__(256) void labelprop_f32_kernel(LabelPropF32Args a, int nsub) {
  constexpr int BQ = 64, BKEY = 64, BC = 32;
  if (a.run_flag && *a.run_flag == 0) return;      // fallback launch of the two-pass path (labelprop2.hip): nothing overflowed
  __shared__ __attribute__((aligned(16))) float sK[BC / 2][BKEY][2];
  __shared__ __attribute__((aligned(16))) float sQ[BC / 2][BQ][2];
  __shared__ int sKC[BKEY];
  __shared__ float sMV[BQ * 4 * LPX_TOPK];
  __shared__ int sMI[BQ * 4 * LPX_TOPK];
  __shared__ __attribute__((aligned(16))) vfs_u32x2q sQue[LPX_QCAP * 256];   // per-lane candidate queues, slot-major
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, C = a.C, HW = H * W;
  const int tiles_x = (W + 7) >> 3;
  const int qy0 = (blockIdx.x / tiles_x) * 8, qx0 = (blockIdx.x % tiles_x) * 8;
  const int nst = (C + BC - 1) / BC;
  const int kh = wave & 1, qh = wave >> 1;
  const int li = lane & 31, lk = lane >> 5;
  // this lane's query: column li of the wave's 32-query half
  const int ql = qh * 32 + li;
  const int qy = qy0 + (ql >> 3), qx = qx0 + (ql & 7);
  const bool q_in = qy < H && qx < W;
  // loaders: row t&63, float4 groups (t>>6) and (t>>6)+4 of a 32-channel stage
  const int lrow = t & 63, lq = t >> 6;
  const int lqy = qy0 + (lrow >> 3), lqx = qx0 + (lrow & 7);
  const bool lq_ok = lqy < H && lqx < W;
  const float* qsrc = a.fbank + ((size_t)a.qframe * HW + (size_t)(lq_ok ? lqy * W + lqx : 0)) * C;

  float tv[LPX_TOPK];
  int ti[LPX_TOPK];
#pragma unroll
  for (int i = 0; i < LPX_TOPK; ++i) { tv[i] = -INFINITY; ti[i] = LPX_NONE; }
  // two-stage streaming top-k (as labelprop.hip): a candidate STRICTLY below thr - the best 10th-best score of the four
  // lanes that share its query - has ten better candidates and cannot be in the query's top 10 under any tie rule; the
  // others are queued in LDS and drained through the sorted insertion once per key block.  Exactness is untouched: the
  // set that survives still contains the true top 10, and the insertion keeps the total order.
  float thr = -INFINITY;
  int qn = 0;
  auto drain = [&]() {
    for (int i = 0; __any(i < qn); ++i) {
      const vfs_u32x2q e = sQue[i * 256 + t];
      const bool on = i < qn;
      lpx_insert(tv, ti, on ? __builtin_bit_cast(float, e[0]) : -INFINITY, on ? (int)e[1] : LPX_NONE);
    }
    qn = 0;
    // lanes l and l+32 of this wave and the same lanes of the partner wave share a query; only the in-wave pair is cheap to reach
    const float m = tv[LPX_TOPK - 1];
    thr = fmaxf(m, __shfl_xor(m, 32));
  };

  // blockIdx.y = (key-frame split, sub-split): the 64-key blocks of a frame's window are dealt to nsub workgroups in contiguous runs
  const int nfs = (int)gridDim.y / nsub, fs = (int)blockIdx.y / nsub, sub = (int)blockIdx.y - fs * nsub;
  const int fpb = (a.nkeys + nfs - 1) / nfs;
  const int f_begin = fs * fpb, f_end = min(a.nkeys, f_begin + fpb);
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
    const int nkb = (nwin + BKEY - 1) / BKEY, cpb = (nkb + nsub - 1) / nsub;
    const int kb_end = min(nkb, (sub + 1) * cpb);
    for (int kb = sub * cpb; kb < kb_end; ++kb) {
      const int kk = kb * BKEY + lrow;
      const bool k_ok = kk < nwin;
      const int kc = k_ok ? kk : nwin - 1;
      const int ky = wy0 + kc / ww, kx = wx0 + kc % ww;
      const float* ksrc = a.fbank + ((size_t)slot * HW + (size_t)(ky * W + kx)) * C;
      if (lq == 0) sKC[lrow] = k_ok ? ((ky << 16) | kx) : -1;
      f32x4 rk[2], rq[2];
      auto load = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = st * BC + (lq + 4 * i) * 4;
          if (RAGGED) {
            rk[i] = (k_ok && c < C) ? ldf4(ksrc + c) : zerof4();
            rq[i] = (lq_ok && c < C) ? ldf4(qsrc + c) : zerof4();
          } else {
            rk[i] = ldf4(ksrc + c);
            rq[i] = ldf4(qsrc + c);
          }
        }
      };
      auto store = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int g = lq + 4 * i;
          *reinterpret_cast<vfs_f32x2*>(&sK[2 * g][lrow][0]) = (vfs_f32x2){rk[i][0], rk[i][1]};
          *reinterpret_cast<vfs_f32x2*>(&sK[2 * g + 1][lrow][0]) = (vfs_f32x2){rk[i][2], rk[i][3]};
          *reinterpret_cast<vfs_f32x2*>(&sQ[2 * g][lrow][0]) = (vfs_f32x2){rq[i][0], rq[i][1]};
          *reinterpret_cast<vfs_f32x2*>(&sQ[2 * g + 1][lrow][0]) = (vfs_f32x2){rq[i][2], rq[i][3]};
        }
      };
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      load(0);
      for (int st = 0; st < nst; ++st) {
        store();
        __syncthreads();
        if (st + 1 < nst) load(st + 1);
#pragma unroll
        for (int s = 0; s < BC / 2; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[s][kh * 32 + li][lk], sQ[s][qh * 32 + li][lk], acc, 0, 0, 0);
        // issue order: the operand reads of MFMA pair p+1 before the MFMAs of pair p (0x100 = DS read, 0x008 = MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int p = 0; p < BC / 4 - 2; ++p) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __syncthreads();
      }
      // scores -> circle mask -> streaming top-k: lane holds keys row(rg) = (rg&3) + 8*(rg>>2) + 4*lk of its query
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int pk = sKC[kh * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * lk];
        const int cy = pk >> 16, cx = pk & 0xffff;
        bool ok = pk >= 0 && q_in;
        if (r > 0) {
          const int dy = cy - qy, dx = cx - qx;
          ok = ok && (dy * dy + dx * dx < r * r);
        }
        const float sc = ok ? acc[rg] / a.temperature : -INFINITY;
        const int id = ok ? f * HW + cy * W + cx : LPX_NONE;
        if ((rg & 3) == 0 && __any(qn > LPX_QCAP - 4)) drain();
        if (ok && sc >= thr) {
          sQue[qn * 256 + t] = (vfs_u32x2q){__builtin_bit_cast(unsigned, sc), (unsigned)id};
          ++qn;
        }
      }
      drain();
      __syncthreads();   // sKC is rewritten by the next key block
    }
  }
  // the 4 partial lists of every query -> LDS -> one lane per query selects the split's top-10
  const int part = kh * 2 + lk;
#pragma unroll
  for (int i = 0; i < LPX_TOPK; ++i) {
    sMV[(ql * 4 + part) * LPX_TOPK + i] = tv[i];
    sMI[(ql * 4 + part) * LPX_TOPK + i] = ti[i];
  }
  __syncthreads();
  if (t < BQ) {
    const int y = qy0 + (t >> 3), x = qx0 + (t & 7);
    if (y < H && x < W) {
      float* cv = sMV + t * 4 * LPX_TOPK;
      int* ci = sMI + t * 4 * LPX_TOPK;
      float* pv = a.pval + ((size_t)blockIdx.y * HW + (y * W + x)) * LPX_TOPK;
      int* pi = a.pidx + ((size_t)blockIdx.y * HW + (y * W + x)) * LPX_TOPK;
      for (int k = 0; k < LPX_TOPK; ++k) {
        int best = 0;
        for (int c = 1; c < 4 * LPX_TOPK; ++c)
          if (lpx_better(cv[c], ci[c], cv[best], ci[best])) best = c;
        pv[k] = cv[best]; pi[k] = ci[best];
        cv[best] = -INFINITY; ci[best] = LPX_NONE;
      }
    }
  }
}

// merge the per-split lists (same total order), softmax over the top-k in sorted order, weighted sum of the values
__global__ __launch_bounds__(256) void labelprop_f32_merge_kernel(LabelPropF32Args a, int nsplit) {
  const int HW = a.H * a.W;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= HW) return;
  if (a.run_flag && *a.run_flag == 0) return;
  float bv[LPX_TOPK];
  int bi[LPX_TOPK];
#pragma unroll
  for (int k = 0; k < LPX_TOPK; ++k) { bv[k] = -INFINITY; bi[k] = LPX_NONE; }
  for (int sp = 0; sp < nsplit; ++sp) {
    const float* pv = a.pval + ((size_t)sp * HW + q) * LPX_TOPK;
    const int* pi = a.pidx + ((size_t)sp * HW + q) * LPX_TOPK;
    for (int c = 0; c < LPX_TOPK; ++c) {
      const float v = pv[c];
      const int id = pi[c];
      if (lpx_better(v, id, bv[LPX_TOPK - 1], bi[LPX_TOPK - 1])) lpx_insert(bv, bi, v, id);
    }
  }
  float e[LPX_TOPK], z = 0.f;
#pragma unroll
  for (int k = 0; k < LPX_TOPK; ++k) {
    e[k] = (k < a.topk && bi[k] != LPX_NONE && bv[k] > -INFINITY) ? vexp(bv[k] - bv[0]) : 0.f;
    z = z + e[k];
  }
  float* o = a.out + (size_t)q * a.CO;
  for (int c = 0; c < a.CO; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LPX_TOPK; ++k) {
      if (e[k] > 0.f) {
        const int fr = bi[k] / HW, px = bi[k] - fr * HW;
        s = __builtin_fmaf(e[k] / z, a.sbank[((size_t)a.kslot[fr] * HW + px) * a.CO + c], s);
      }
    }
    o[c] = s;
  }
}

int vfs_option_lpx_target = 0;   // workgroups the key frames of a query tile are split into; 0 = auto (A/B knob)
int vfs_option_lpx_wgs = 0;      // workgroups a launch should reach by ALSO splitting a key frame's window; 0 = auto (3072 for C >= 512:
                                 // R50 5.33-5.39 vs 5.45-5.50 ms per frame; R18 is faster without: 1.48 vs 1.54), < 0 = never
int vfs_option_lpx_minb = 4;     // ... with at least this many 64-key blocks per workgroup
int vfs_labelprop_f32_launch(const LabelPropF32Args& a, hipStream_t s) {
  if (a.C % 4) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32: C % 4");
  if (a.nkeys < 1 || a.nkeys > LP_MAX_KEYS) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32: 1 <= nkeys <= 64");
  if (a.topk < 1 || a.topk > LPX_TOPK) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32: 1 <= topk <= 10");
  if (a.H >= 32768 || a.W >= 65536 || (long long)a.nkeys * a.H * a.W >= 0x7fffffffLL)
    return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32: map too large for the packed candidate ids");
  if (!(a.temperature > 0.f)) return vfs_set_error(VFS_ERR_ARG, "labelprop_f32: temperature > 0");
  if (a.pval == nullptr || a.pidx == nullptr) return vfs_set_error(VFS_ERR_ARG, "labelprop_f32: partial workspace missing");
  const int tiles = ((a.H + 7) / 8) * ((a.W + 7) / 8);
  // long channel loops (ResNet-50 res4: C = 1024) balance better with one or two key frames per workgroup (A/B on MI355X:
  // 5.22 -> 5.05 ms per frame), short ones (ResNet-18: C = 256) with fewer, longer workgroups (1.46 vs 1.51)
  const int target = vfs_option_lpx_target > 0 ? vfs_option_lpx_target : (a.C >= 512 ? 2400 : 768);
  int nsplit = (target + tiles - 1) / tiles;
  if (nsplit > a.nkeys) nsplit = a.nkeys;
  if (nsplit > LP_MAX_FSPLIT) nsplit = LP_MAX_FSPLIT;
  const int fpb = (a.nkeys + nsplit - 1) / nsplit;
  nsplit = (a.nkeys + fpb - 1) / fpb;
  // a workgroup of one key frame runs ~1 ms (R50, radius 18) and a launch has only 1-3 of them per slot (the first frames of
  // a clip: fewer workgroups than CUs): the window's 64-key blocks are dealt to nsub workgroups as well
  int nsub = 1;
  const int wgs = vfs_option_lpx_wgs > 0 ? vfs_option_lpx_wgs : (vfs_option_lpx_wgs == 0 && a.C >= 512 ? 3072 : 0);
  if (wgs > 0) {
    const int wh = a.radius > 0 ? min(a.H, 8 + 2 * (a.radius - 1)) : a.H, ww = a.radius > 0 ? min(a.W, 8 + 2 * (a.radius - 1)) : a.W;
    const int blocks = (wh * ww + 63) / 64;
    nsub = (wgs + tiles * nsplit - 1) / (tiles * nsplit);
    nsub = min(nsub, max(1, blocks / max(1, vfs_option_lpx_minb)));
    nsub = max(1, min(nsub, LP_MAX_SPLIT / nsplit));
  }
  nsplit *= nsub;
  if (a.C % 32)
    hipLaunchKernelGGL(labelprop_f32_kernel<true>, dim3(tiles, nsplit), dim3(256), 0, s, a, nsub);
  else
    hipLaunchKernelGGL(labelprop_f32_kernel<false>, dim3(tiles, nsplit), dim3(256), 0, s, a, nsub);
  int rc = vfs_check_launch("labelprop_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(labelprop_f32_merge_kernel, dim3((a.H * a.W + 255) / 256), dim3(256), 0, s, a, nsplit);
  return vfs_check_launch("labelprop_f32_merge");
}

// ---------------------------------------------------------------------------------------------
// post-processing (vanilla_tracker.py:162-181), every step a single fp32 operation (oracle: xo_bilerp / xo_seg_postprocess)
__device__ __forceinline__ float bilerp_exact(const float* __restrict__ seg, int H, int W, int CO, int c, int oy, int ox, float sy,
                                              float sx) {
  float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v00 = seg[((size_t)y0 * W + x0) * CO + c], v01 = seg[((size_t)y0 * W + x1) * CO + c];
  const float v10 = seg[((size_t)y1 * W + x0) * CO + c], v11 = seg[((size_t)y1 * W + x1) * CO + c];
  const float top = hx * v00 + lx * v01, bot = hx * v10 + lx * v11;
  return hy * top + ly * bot;
}

__global__ __launch_bounds__(256) void seg_minmax_exact_kernel(const float* __restrict__ seg, float* __restrict__ partial, int H, int W,
                                                               int CO, int Ho, int Wo) {
  __shared__ float smn[256], smx[256];
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  const int total = Ho * Wo;
  for (int c = 0; c < CO; ++c) {
    float mn = INFINITY, mx = -INFINITY;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
      const float v = bilerp_exact(seg, H, W, CO, c, p / Wo, p % Wo, sy, sx);
      mn = v < mn ? v : mn; mx = v > mx ? v : mx;
    }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        const float o1 = smn[threadIdx.x + s], o2 = smx[threadIdx.x + s];
        smn[threadIdx.x] = o1 < smn[threadIdx.x] ? o1 : smn[threadIdx.x];
        smx[threadIdx.x] = o2 > smx[threadIdx.x] ? o2 : smx[threadIdx.x];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      partial[((size_t)blockIdx.x * CO + c) * 2] = smn[0];
      partial[((size_t)blockIdx.x * CO + c) * 2 + 1] = smx[0];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void seg_argmax_exact_kernel(const float* __restrict__ seg, const float* __restrict__ partial, int nblk,
                                                               uint8_t* __restrict__ label, int H, int W, int CO, int Ho, int Wo) {
  __shared__ float smn[LP_MAX_CLASSES], smx[LP_MAX_CLASSES];
  if ((int)threadIdx.x < CO) {
    float mn = INFINITY, mx = -INFINITY;
    for (int b = 0; b < nblk; ++b) {
      const float p0 = partial[((size_t)b * CO + threadIdx.x) * 2], p1 = partial[((size_t)b * CO + threadIdx.x) * 2 + 1];
      mn = p0 < mn ? p0 : mn; mx = p1 > mx ? p1 : mx;
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
      float v = bilerp_exact(seg, H, W, CO, c, oy, ox, sy, sx);
      if (smx[c] > 0.f) v = (v - smn[c]) / (smx[c] - smn[c] + 1e-12f);
      if (v > best) { best = v; bc = c; }
    }
    label[p] = (uint8_t)bc;
  }
}

int vfs_seg_postprocess_exact_launch(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                                     hipStream_t s) {
  if (CO < 1 || CO > LP_MAX_CLASSES) return vfs_set_error(VFS_ERR_SHAPE, "seg_postprocess_exact: 1 <= classes <= 256");
  const int nblk = LP_POST_BLOCKS;
  hipLaunchKernelGGL(seg_minmax_exact_kernel, dim3(nblk), dim3(256), 0, s, seg, partial, H, W, CO, Ho, Wo);
  int rc = vfs_check_launch("seg_minmax_exact");
  if (rc) return rc;
  int blocks = (Ho * Wo + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(seg_argmax_exact_kernel, dim3(blocks), dim3(256), 0, s, seg, partial, nblk, label, H, W, CO, Ho, Wo);
  return vfs_check_launch("seg_argmax_exact");
}

// F.interpolate(mode='bilinear', align_corners=False) between arbitrary layouts (element (c, y, x) of a map at
// c*sc + y*sy + x*sx): the one-hot reference map -> feature resolution, soft label maps -> original resolution
// (vanilla_tracker.py:101-111,162-166 with a 4-D ref_seg_map)
__global__ __launch_bounds__(256) void bilinear_resize_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                                  int Ho, int Wo, long long ssc, long long ssy, long long ssx, long long dsc,
                                                                  long long dsy, long long dsx) {
  const long long total = (long long)C * Ho * Wo;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long p = i / C;
  const int ox = (int)(p % Wo), oy = (int)(p / Wo);
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* s = src + (size_t)c * ssc;
  const float v00 = s[y0 * ssy + x0 * ssx], v01 = s[y0 * ssy + x1 * ssx];
  const float v10 = s[y1 * ssy + x0 * ssx], v11 = s[y1 * ssy + x1 * ssx];
  const float top = hx * v00 + lx * v01, bot = hx * v10 + lx * v11;
  dst[(size_t)c * dsc + (size_t)oy * dsy + (size_t)ox * dsx] = hy * top + ly * bot;
}
int vfs_bilinear_resize_f32_launch(const float* src, float* dst, int C, int H, int W, int Ho, int Wo, int src_nhwc, int dst_nhwc,
                                   hipStream_t s) {
  if (C < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return vfs_set_error(VFS_ERR_SHAPE, "bilinear_resize_f32: geometry");
  const long long total = (long long)C * Ho * Wo;
  const long long ssc = src_nhwc ? 1 : (long long)H * W, ssy = src_nhwc ? (long long)W * C : W, ssx = src_nhwc ? C : 1;
  const long long dsc = dst_nhwc ? 1 : (long long)Ho * Wo, dsy = dst_nhwc ? (long long)Wo * C : Wo, dsx = dst_nhwc ? C : 1;
  hipLaunchKernelGGL(bilinear_resize_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, C, H, W, Ho, Wo, ssc, ssy,
                     ssx, dsc, dsy, dsx);
  return vfs_check_launch("bilinear_resize_f32");
}

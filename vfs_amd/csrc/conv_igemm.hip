// Implicit-GEMM convolution for gfx950: forward, dgrad and the 7x7 stem share one kernel.
//
//   out[pixel][chan] = sum_k Wt[chan][k] * G[pixel][k]   (+bias[chan]) (+add[pixel][chan])
//
// Replaces the torch conv2d / linear calls of the reference's hot path
// (mmaction/models/backbones/resnet.py:51-73,163-191,267-277,425-434 via mmcv ConvModule;
// mmaction/models/heads/sim_siam_head.py:78-111) and their autograd dgrad.
//
// Tiling (one workgroup = 4 waves of 64 lanes):
//   * 128 pixels x BC channels (BC = 64|128) per workgroup, K-steps of 64
//   * MFMA v_mfma_f32_16x16x32_bf16, rows(i)=channels (A operand = packed weights),
//     cols(j)=pixels (B operand = gathered NHWC activations); each wave owns (BC/2) x 64
//   * global -> VGPR (16 B/lane, 128-byte rows fully coalesced) -> XOR-swizzled LDS,
//     double-buffered: tile kt+1 is in flight while tile kt feeds the MFMAs; one barrier/K-step
//   * epilogue: lane holds 4 consecutive channels of one pixel -> one 8-byte NHWC store;
//     optional per-channel (sum, sum of squares) of the bf16-ROUNDED outputs, reduced
//     wave-wide with shuffles and written as one deterministic partial per pixel-block
//     (consumed by bn_reduce_partials: BatchNorm batch statistics without an extra pass)
#include "vfs_conv.h"


template <int BC, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BP = 128;
  constexpr int WC = BC / 2;   // channels per wave
  constexpr int TM = WC / 16;  // 16-channel tiles per wave
  constexpr int TN = 4;        // 16-pixel tiles per wave
  constexpr int WLD = BC / 32; // weight rows loaded per thread
  __shared__ __attribute__((aligned(16))) bf16_t sW[2][BC * 64];
  __shared__ __attribute__((aligned(16))) bf16_t sX[2][BP * 64];
  __shared__ float sRed[2][BC][2];

  const ConvGeom g = a.g;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wc = wave >> 1, wp = wave & 1;
  const int ncb = (a.Cout + BC - 1) / BC;
  const int pb = blockIdx.x / ncb, cb = blockIdx.x - pb * ncb;
  const int m0 = pb * BP, c0 = cb * BC;
  const int j = t & 7, row0 = t >> 3;
  const int nk = g.Ktot >> 6;

  PixCoord pc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pc[i] = pix_decode<MODE>(g, m0 + row0 + 32 * i);

  u32x4 xr[4], wr[WLD];
  auto load_tiles = [&](int kt) {
    KStep ks = kstep_decode<MODE>(g, kt);
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = gather16<MODE>(g, a.src, pc[i], ks, j);
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      int c = c0 + row0 + 32 * i;
      wr[i] = (c < a.Cout) ? ld16(a.wgt + (size_t)c * g.Ktot + kt * 64 + j * 8) : zero16();
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st16(&sX[buf][lds_off(row0 + 32 * i, j)], xr[i]);
#pragma unroll
    for (int i = 0; i < WLD; ++i) st16(&sW[buf][lds_off(row0 + 32 * i, j)], wr[i]);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_tiles(kt + 1);
    mma_kstep<TM, TN, false>(sW[cur], sX[cur], wc * WC, wp * 64, lane, acc);
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int lr = lane & 15, lq = lane >> 4;
  const bool do_stats = a.stats != nullptr;
  float s1[TM][4], s2[TM][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[tm][r] = 0.f; s2[tm][r] = 0.f; }

#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int m = m0 + wp * 64 + tn * 16 + lr;
    const bool mok = m < g.M;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int c = c0 + wc * WC + tm * 16 + lq * 4;
      if (mok && c < a.Cout) {
        float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
        if (a.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += a.bias[c + r];
        }
        const size_t o = (size_t)m * a.Cout + c;
        if (a.add) {
          u32x2 ad = ld8(a.add + o);
          v[0] += bflo(ad.x); v[1] += bfhi(ad.x); v[2] += bflo(ad.y); v[3] += bfhi(ad.y);
        }
        u32x2 pk;
        pk.x = pack2bf(v[0], v[1]);
        pk.y = pack2bf(v[2], v[3]);
        st8(a.out + o, pk);
        if (do_stats) {
          float q0 = bflo(pk.x), q1 = bfhi(pk.x), q2 = bflo(pk.y), q3 = bfhi(pk.y);
          s1[tm][0] += q0; s2[tm][0] += q0 * q0;
          s1[tm][1] += q1; s2[tm][1] += q1 * q1;
          s1[tm][2] += q2; s2[tm][2] += q2 * q2;
          s1[tm][3] += q3; s2[tm][3] += q3 * q3;
        }
      }
    }
  }
  if (do_stats) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x1 = s1[tm][r], x2 = s2[tm][r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          x1 += __shfl_xor(x1, d);
          x2 += __shfl_xor(x2, d);
        }
        if (lr == 0) {
          int cl = wc * WC + tm * 16 + lq * 4 + r;
          sRed[wp][cl][0] = x1;
          sRed[wp][cl][1] = x2;
        }
      }
    __syncthreads();
    if (t < BC && c0 + t < a.Cout) {
      float* dst = a.stats + (size_t)pb * 2 * a.Cout;
      dst[c0 + t] = sRed[0][t][0] + sRed[1][t][0];
      dst[a.Cout + c0 + t] = sRed[0][t][1] + sRed[1][t][1];
    }
  }
}

// ------------------------------------------------------------------ host launcher
template <int BC, int MODE>
static int launch_igemm(const ConvArgs& a, hipStream_t stream) {
  int npb = (a.g.M + 127) / 128;
  int ncb = (a.Cout + BC - 1) / BC;
  hipLaunchKernelGGL((conv_igemm_kernel<BC, MODE>), dim3(npb * ncb), dim3(256), 0, stream, a);
  return vfs_check_launch("conv_igemm");
}

int vfs_conv_igemm_dispatch(const ConvArgs& a, int mode, hipStream_t stream) {
  if (a.g.Ktot % 64 != 0 || a.Cout % 4 != 0) return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: K%64 or Cout%4");
  if (mode != GATHER_STEM && a.g.C % 64 != 0) return vfs_set_error(VFS_ERR_SHAPE, "conv_igemm: C%64");
  const bool wide = (a.Cout % 128 == 0);
  switch (mode) {
    case GATHER_FWD:
      return wide ? launch_igemm<128, GATHER_FWD>(a, stream) : launch_igemm<64, GATHER_FWD>(a, stream);
    case GATHER_DGRAD:
      return wide ? launch_igemm<128, GATHER_DGRAD>(a, stream) : launch_igemm<64, GATHER_DGRAD>(a, stream);
    case GATHER_STEM:
      return launch_igemm<64, GATHER_STEM>(a, stream);
  }
  return vfs_set_error(VFS_ERR_ARG, "conv_igemm: bad mode");
}

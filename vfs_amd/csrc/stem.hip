// Dedicated kernels for the 7x7 / stride-2 / pad-3 stem (resnet.py:422-434): 32 % of all conv-output
// elements of a ResNet-18 step live here, with only 3 input channels.
//
// Forward (stem_fwd_direct): one workgroup walks spatial tiles of 8x16 output pixels.  The whole
// packed weight tensor [64 cout][7 rows][8 cols x 4 ch] (28 KB) stays in LDS; per tile the raw
// NHWC4 input patch (21 rows x 38 columns x 8 B = 6.4 KB) is staged once and the MFMA B fragments
// are read STRAIGHT from it: for kernel row r the 32 k-values of output pixel (py,px) are the 64
// contiguous bytes at patch[2py+r][2px .. 2px+7] -- no im2col copy, K = 7x32 instead of the generic
// kernel's 8x32, and HBM traffic = input once + output once.
//
// Backward (stem_wgrad_fused): weight gradient with the BatchNorm backward "apply" pass folded into
// the operand load: dY = scale*(ga - m1 - xhat*m2) is rebuilt per pixel from the raw conv output and
// the (4x smaller) pooled-gradient tensors while the tile is staged, so the full-resolution dx
// tensor is never written or re-read.  Fragments come from pixel-major LDS tiles through
// ds_read_b64_tr_b16.  No input gradient is needed (the frames are leaves).
#include "vfs_conv.h"
#include "vfs_ops.h"
#include "vfs_stem.h"

#define OOB_OFFSET 0xFFFFFFF0u
typedef __attribute__((ext_vector_type(4))) short s16x4;

// patch geometry of an 8x16 output tile
#define ST_PH 21
#define ST_PW 38
#define ST_PROW (ST_PW * 4)   // elements per patch row (38 columns x 4 channels), 304 B

__device__ __forceinline__ int stem_w_off(int r, int cout, int chunk) {   // swizzled [r][cout][4 chunks of 8]
  return (r * 64 + cout) * 32 + ((chunk ^ ((cout >> 2) & 3)) << 3);
}

#define ST_STAGE_ROW 72   // staged output pixel row: 64 channels + 16 bytes of padding
__global__ __launch_bounds__(256) void stem_fwd_direct_kernel(ConvArgs a, int tiles_per_block, int ntiles) {
  __shared__ __attribute__((aligned(16))) bf16_t sW[7 * 64 * 32];
  __shared__ __attribute__((aligned(16))) bf16_t sX[ST_PH * ST_PROW];
  __shared__ __attribute__((aligned(16))) bf16_t sOut[4 * 32 * ST_STAGE_ROW];
  // the statistics hand-off of a wave ([2][64] floats) lives at the head of that wave's OWN output slab: the wave has
  // read its slab back before it writes there, and the next slab writes come after the loop's closing barriers.
  // (A separate 2 KB array put the kernel at 55.5 KB of LDS = two workgroups per CU; 53.5 KB fits three.)
  auto sred = [&](int w) { return reinterpret_cast<float*>(sOut + w * (32 * ST_STAGE_ROW)); };
  constexpr int PL = (ST_PH * 19 + 255) / 256;     // 16-byte patch loads per thread (2)
  const ConvGeom g = a.g;               // H, W = padded input dims (NHWC4), Ho, Wo = output dims
  const int t = threadIdx.x, lane = t & 63, wp = t >> 6;   // wave = 64 cout x 32 pixels (tile rows 2wp, 2wp+1)
  const int lr = lane & 15, lq = lane >> 4;
  const int tiles_x = (g.Wo + 15) / 16, tiles_y = (g.Ho + 7) / 8, tiles_img = tiles_x * tiles_y;

  // weights [64][8][8][4] (row 7 and column 0 are zero padding) -> LDS rows r = 0..6
  for (int i = t; i < 7 * 64 * 4; i += 256) {
    const int chunk = i & 3, cout = (i >> 2) & 63, r = i >> 8;
    st16(&sW[stem_w_off(r, cout, chunk)], ld16(a.wgt + ((size_t)cout * 8 + r) * 32 + chunk * 8));
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.src, 0, (unsigned)((size_t)g.N * g.H * g.W * 4 * 2), 0x00020000);

  // patch of a tile: rows 2y0-3.., 19 chunks (2 columns) per row starting at column 2x0-4
  u32x4 pr_[PL];
  auto load_patch = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / tiles_img;
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int i = t + 256 * k;
      const int prow = i / 19, pc = i - prow * 19;
      const int y = 2 * ty * 8 - 3 + prow, x = 2 * tx * 16 - 4 + 2 * pc;
      const bool ok = i < ST_PH * 19 && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
      const unsigned off = ok ? (unsigned)((((size_t)(n * g.H + y) * g.W + x) * 4) * 2) : OOB_OFFSET;
      pr_[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int i = t + 256 * k;
      if (i < ST_PH * 19) {
        const int prow = i / 19, pc = i - prow * 19;
        st16(&sX[prow * ST_PROW + pc * 8], pr_[k]);
      }
    }
  };

  const int t_begin = blockIdx.x * tiles_per_block, t_end = min(ntiles, t_begin + tiles_per_block);
  if (t_begin >= t_end) return;
  const bool do_stats = a.stats != nullptr;
  // BatchNorm statistics accumulate in registers over a RUN of tiles of the same image (the two views
  // are different images) and are reduced once per run into the row of the run's first tile
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[tm][q] = 0.f; s2[tm][q] = 0.f; }
  int run_first = t_begin;
  load_patch(t_begin);
  store_patch();
  __syncthreads();                       // patch + weights are in LDS
  bf16_t* slab = sOut + wp * (32 * ST_STAGE_ROW);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / tiles_img;
    const int y0 = ty * 8, x0 = tx * 16;
    const bool has_next = tile + 1 < t_end;
    if (has_next) load_patch(tile + 1);  // in flight during this tile's MFMAs and epilogue
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4][2];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
        af[tm] = *reinterpret_cast<const bf16x8*>(&sW[stem_w_off(r, tm * 16 + lr, lq)]);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int py = wp * 2 + tn;
        bfr[tn] = *reinterpret_cast<const bf16x8*>(&sX[(2 * py + r) * ST_PROW + (2 * lr + 2 * lq) * 4]);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    // ---- epilogue: transpose the wave's 64 x 32 outputs through its LDS slab, whole 128-byte pixel rows out
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const bool mok = y0 + wp * 2 + tn < g.Ho && x0 + lr < g.Wo;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        u32x2 pk;
        pk.x = pack2bf(acc[tm][tn][0], acc[tm][tn][1]);
        pk.y = pack2bf(acc[tm][tn][2], acc[tm][tn][3]);
        st8(&slab[(tn * 16 + lr) * ST_STAGE_ROW + tm * 16 + lq * 4], pk);
        if (do_stats && mok) {   // statistics of the STORED (bf16) values
          const float q0 = bflo(pk.x), q1 = bfhi(pk.x), q2 = bflo(pk.y), q3 = bfhi(pk.y);
          s1[tm][0] += q0; s2[tm][0] += q0 * q0;
          s1[tm][1] += q1; s2[tm][1] += q1 * q1;
          s1[tm][2] += q2; s2[tm][2] += q2 * q2;
          s1[tm][3] += q3; s2[tm][3] += q3 * q3;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();     // no code: in-order LDS pipe; keeps the compiler (and the CPU emulator) honest
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = i * 8 + (lane >> 3), ch = lane & 7;
      const int y = y0 + wp * 2 + (p >> 4), x = x0 + (p & 15);
      if (y < g.Ho && x < g.Wo)
        st16(a.out + (((size_t)n * g.Ho + y) * g.Wo + x) * 64 + ch * 8, ld16(&slab[p * ST_STAGE_ROW + ch * 8]));
    }
    const bool run_ends = !has_next || (tile + 1) / tiles_img != n;      // uniform
    if (do_stats) {
      if (run_ends) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          f32x4 r1, r2;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            r1[q] = row16_sum(s1[tm][q]);
            r2[q] = row16_sum(s2[tm][q]);
            s1[tm][q] = 0.f; s2[tm][q] = 0.f;
          }
          if (lr == 0) {
            *reinterpret_cast<f32x4*>(sred(wp) + tm * 16 + lq * 4) = r1;
            *reinterpret_cast<f32x4*>(sred(wp) + 64 + tm * 16 + lq * 4) = r2;
          }
        }
        __syncthreads();
        if (t < 128) {
          const int st = t >> 6, cl = t & 63;
          const int o = st * 64 + cl;
          a.stats[(size_t)run_first * 128 + t] = (sred(0)[o] + sred(1)[o]) + (sred(2)[o] + sred(3)[o]);
        }
        if (tile != run_first && t >= 128) a.stats[(size_t)tile * 128 + (t - 128)] = 0.f;
        run_first = tile + 1;
      } else if (tile != run_first && t < 128) {
        a.stats[(size_t)tile * 128 + t] = 0.f;
      }
    }
    __syncthreads();                     // every wave is done with this tile's patch (and with sRed)
    if (has_next) store_patch();
    __syncthreads();
  }
}

int vfs_stem_tiles(int N, int Ho, int Wo) { return N * ((Ho + 7) / 8) * ((Wo + 15) / 16); }

int vfs_stem_fwd_direct_launch(const ConvArgs& a, hipStream_t stream) {
  const int ntiles = vfs_stem_tiles(a.g.N, a.g.Ho, a.g.Wo);
  const int maxb = vfs_option_stem_blocks > 0 ? vfs_option_stem_blocks : 2048;
  int blocks = ntiles < maxb ? ntiles : maxb;
  const int tpb = (ntiles + blocks - 1) / blocks;
  blocks = (ntiles + tpb - 1) / tpb;
  hipLaunchKernelGGL(stem_fwd_direct_kernel, dim3(blocks), dim3(256), 0, stream, a, tpb, ntiles);
  return vfs_check_launch("stem_fwd_direct");
}

// ---------------------------------------------------------------------------------------------
// dW[cout][r][s_idx*4+c] = sum_pixels dY[p][cout] * X[2py+r][2px+s_idx][c]   (s_idx = s+1; column 0
// and channel 3 are padding), dY rebuilt on the fly (stem_dx_vec).  GEMM rows = 14 tiles of 16
// k-columns (7 kernel rows x two 4-column halves), cols = 64 cout, reduction = 128 tile pixels.
#define SD_RS 72   // dY tile row stride (64 cout + 8 pad) -> conflict-free transpose reads

__global__ __launch_bounds__(256) void stem_wgrad_fused_kernel(StemBwdArgs a, const bf16_t* __restrict__ x4, int Hin, int Win,
                                                               float* __restrict__ partial, int tiles_per_block,
                                                               int ntiles) {
  // a.x = raw stem output [N][Ho][Wo][64] (a.H, a.W = Ho, Wo); x4 = NHWC4 input [N][Hin][Win][4]
  __shared__ __attribute__((aligned(16))) bf16_t sX[ST_PH * ST_PROW];
  __shared__ __attribute__((aligned(16))) bf16_t sD[128 * SD_RS];
  __shared__ float sTab[STEM_MAX_GROUPS * STEM_TAB];  // per-group BN backward coefficients
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wk = wave >> 1, wn = wave & 1;            // wave: 7 k-column tiles x 32 cout
  const int lr = lane & 15, lq = lane >> 4;
  const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;
  const float rc = (float)(1.0 / a.count);
  stem_fill_table(a, sTab, rc);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)x4, 0, (unsigned)((size_t)a.N * Hin * Win * 4 * 2), 0x00020000);

  f32x4 acc[7][2];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[i][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment bases (see conv_wgrad_halo.hip): pixel-in-16 = 4*lq + (lr>>2), 4-element run lr&3
  const int pl = 4 * lq + (lr >> 2), q4 = lr & 3;
  const int d_base = pl * SD_RS + wn * 32 + q4 * 4;
  const int x_base = (2 * pl + q4) * 4;

  const int t_begin = blockIdx.x * tiles_per_block, t_end = min(ntiles, t_begin + tiles_per_block);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int y0 = ty * 8, x0 = tx * 16;
    __syncthreads();
    for (int i = t; i < ST_PH * 19; i += 256) {
      const int pr = i / 19, pc = i - pr * 19;
      const int y = 2 * y0 - 3 + pr, x = 2 * x0 - 4 + 2 * pc;
      const bool ok = (unsigned)y < (unsigned)Hin && (unsigned)x < (unsigned)Win;
      const unsigned off = ok ? (unsigned)((((size_t)(n * Hin + y) * Win + x) * 4) * 2) : OOB_OFFSET;
      st16(&sX[pr * ST_PROW + pc * 8], __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // a wave = 8 pixels x 8 channel vectors: the pixels of one row AND one column parity (columns 0, 2, .. 14 or 1, 3, .. 15),
      // so that stem_dx_vec's "does this window exist" branches are uniform per wave
      const int v = t + 256 * i, ps = v >> 3, j = v & 7;
      const int q = ps & 15, p = (ps & ~15) | (q < 8 ? 2 * q : 2 * (q - 8) + 1);
      const int y = y0 + (p >> 4), x = x0 + (p & 15);
      float d[8];
      if (y < a.H && x < a.W) {
        stem_dx_vec(a, sTab, n, y, x, j * 8, d);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = 0.f;
      }
      st16(&sD[p * SD_RS + j * 8], pack8(d));
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 bfr[2];
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const bf16_t* b0 = sD + d_base + (32 * ks) * SD_RS + tn * 16;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0 + 16 * SD_RS));
        bfr[tn][0] = lo[0]; bfr[tn][1] = lo[1]; bfr[tn][2] = lo[2]; bfr[tn][3] = lo[3];
        bfr[tn][4] = hi[0]; bfr[tn][5] = hi[1]; bfr[tn][6] = hi[2]; bfr[tn][7] = hi[3];
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int rt = wk * 7 + i, r = rt >> 1, s0 = (rt & 1) * 4;
        // pixel rows py = 2ks (h=0) and 2ks+1 (h=1): patch row 2py + r, column 2px + s0 + q
        const bf16_t* a0 = sX + x_base + (4 * ks + r) * ST_PROW + s0 * 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * ST_PROW));
        bf16x8 af;
        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
        af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[i][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[tn], acc[i][tn], 0, 0, 0);
      }
    }
  }
  // D[k-column][cout]: lane holds 4 consecutive k (= the 4 channels of column s0+lq) of cout = lane&15
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int rt = wk * 7 + i, r = rt >> 1, s0 = (rt & 1) * 4;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int cout = wn * 32 + tn * 16 + lr;
      *reinterpret_cast<f32x4*>(partial + (((size_t)blockIdx.x * 64 + cout) * 7 + r) * 32 + (s0 + lq) * 4) = acc[i][tn];
    }
  }
}

int vfs_stem_wgrad_fused_launch(const StemBwdArgs& a, const bf16_t* x4, int Hin, int Win, float* partial, int nblocks,
                                hipStream_t stream) {
  const int ntiles = vfs_stem_tiles(a.N, a.H, a.W);
  if ((a.N + a.npg - 1) / a.npg > 8) return vfs_set_error(VFS_ERR_SHAPE, "stem_wgrad_fused: more than 8 BN groups");
  if ((long long)a.N * a.H * a.W * 64 >= (1ll << 31)) return vfs_set_error(VFS_ERR_SHAPE, "stem_wgrad_fused: 2^31 elements or more (32-bit offsets)");
  if (nblocks > ntiles) nblocks = ntiles;
  const int tpb = (ntiles + nblocks - 1) / nblocks;
  if ((ntiles + tpb - 1) / tpb != nblocks) return vfs_set_error(VFS_ERR_SHAPE, "stem_wgrad_fused: nblocks must equal ceil(ntiles / ceil(ntiles/nblocks))");
  hipLaunchKernelGGL(stem_wgrad_fused_kernel, dim3(nblocks), dim3(256), 0, stream, a, x4, Hin, Win, partial, tpb, ntiles);
  return vfs_check_launch("stem_wgrad_fused");
}

// Weight gradient of a 3x3 / stride-1 / pad-1 convolution with the HALO tile in LDS and
// hardware-transposed fragment reads.
//
//   dW[cout][(r,s),cin] = sum_pixels dY[pixel][cout] * X[pixel shifted by (r,s)][cin]
//
// conv_wgrad.hip gives every (tap, cin-chunk) its own workgroup: the same activation pixels are
// re-fetched by nine workgroups and dY by all of them, and the stage transposes through VGPRs
// because the reduction index (pixels) is the slow memory dimension.  Here a workgroup owns
// (64 cin x 64 cout x ALL nine taps) for a range of spatial tiles (8x16 pixels, or two 8x8
// images): per tile it stages the X halo patch and the dY tile ONCE, pixel-major as they lie in
// memory, and builds MFMA fragments with ds_read_b64_tr_b16, the LDS transpose read of gfx950
// (probed on the box: inside a 16-lane group lane i receives the 16-bit elements
// i, 16+i, 32+i, 48+i of the 64 consecutive elements the group's lanes point at).  The dY
// fragments of a 32-pixel k-step are reused by the nine taps; nine accumulator sets (144 VGPRs)
// stay in registers.  Split-K over tile ranges, fp32 partials in the layout wgrad_reduce expects.
#include "vfs_conv.h"
#include "vfs_wgrad_tail.h"

#define OOB_OFFSET 0xFFFFFFF0u
typedef __attribute__((ext_vector_type(4))) short s16x4;

// LDS rows here are LINEAR with a 16-byte pad (72 elements = 144 B per 64-channel row): every
// fragment address is  per-lane base + compile-time constant, so the 160 transpose reads of a
// tile use immediate offsets (no address VALU), the 8 rows a 32-lane group touches start 36 banks
// apart (conflict-free up to a 4-bank wrap) and the 16-byte staging stores stay aligned.
#define RS 72
// 8 pixels x 16 channels, transposed: returns the 8 pixel values of channel (lane&15);
// lo / hi = element offsets of this lane's 4-channel run in the pixel rows of k = 0..3 / 4..7
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int lo_off, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + lo_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + hi_off));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

extern int vfs_option_wgrad_xcd;      // conv_wgrad.hip

template <bool SMALLW, bool BNIN>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_halo_kernel(WgradArgs a, int tiles_per_split, int ntiles) {
  constexpr int TW = SMALLW ? 8 : 16, TH = 8, TI = SMALLW ? 2 : 1;
  constexpr int PW = TW + 2, PH = TH + 2;
  constexpr int PROWS = TI * PH * PW;
  constexpr int PLD = (PROWS * 8 + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t sP[PROWS * RS];   // X halo patch  [patch row][64 cin + pad]
  __shared__ __attribute__((aligned(16))) bf16_t sD[128 * RS];     // dY tile       [pixel][64 cout + pad]
  // BNIN: x is the RAW output of a plain conv-BN-ReLU unit; relu(x*scale+shift) is applied while the patch
  // is staged.  The coefficients of this workgroup's 64-channel chunk live in LDS ([group][scale|shift][64]):
  // held in registers across the MFMA loop they pushed the kernel (144 accumulators) into spills.
  __shared__ __attribute__((aligned(16))) float sBn[BNIN ? 8 * 2 * 64 : 4];

  const ConvGeom g = a.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;            // wave = 32 cin x 32 cout, all taps
  const int lr = lane & 15, lq = lane >> 4;
  const int nchunk = g.C >> 6, ncb = a.Cout >> 6;
  int b = blockIdx.x;
  if (a.xcd_swizzle) {      // XCD-aware block order (see conv_wgrad.hip): the (cin chunk, cout block) tiles of one split share an L2
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = b & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int cc = b % nchunk; b /= nchunk;
  const int cb = b % ncb; b /= ncb;
  const int split = b;
  const int j = t & 7, row0 = t >> 3;
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;   // ragged edge tiles: their dY rows load as zero

  // loader geometry (fixed per thread): patch slots and dY rows
  int p_ti[PLD], p_y[PLD], p_x[PLD];
#pragma unroll
  for (int k = 0; k < PLD; ++k) {
    const int pr = row0 + 32 * k;
    const int ti = pr / (PH * PW), rem = pr - ti * (PH * PW);
    p_ti[k] = pr < PROWS ? ti : -1;
    p_y[k] = rem / PW - 1;
    p_x[k] = rem - (rem / PW) * PW - 1;
  }
  int d_ti[4], d_y[4], d_x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = row0 + 32 * i;
    if (SMALLW) { d_ti[i] = p >> 6; d_y[i] = (p >> 3) & 7; d_x[i] = p & 7; }
    else { d_ti[i] = 0; d_y[i] = p >> 4; d_x[i] = p & 15; }
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.x, 0, (unsigned)((size_t)g.N * g.H * g.W * g.C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.dy, 0, (unsigned)((size_t)g.N * g.H * g.W * a.Cout * 2), 0x00020000);

  u32x4 pv[PLD], dv[4];
  unsigned pvalid = 0;                 // BNIN: bit k = slot k of the staged patch is inside the image
  int bn_gi = 0;                       // BNIN: statistics group of the tile in registers
  auto load_tile = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n0 = (tile / (tiles_x * tiles_y)) * TI;
    const int y0 = ty * TH, x0 = tx * TW;
    if (BNIN) bn_gi = n0 / a.in_npg;
    pvalid = 0;
#pragma unroll
    for (int k = 0; k < PLD; ++k) {
      const int y = y0 + p_y[k], x = x0 + p_x[k], n = n0 + p_ti[k];
      const bool ok = p_ti[k] >= 0 && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W && n < g.N;
      const unsigned off = ok ? (unsigned)((((size_t)(n * g.H + y) * g.W + x) * g.C + cc * 64 + j * 8) * 2) : OOB_OFFSET;
      if (BNIN) pvalid |= ok ? (1u << k) : 0u;
      pv[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + d_ti[i];
      const bool ok = n < g.N && y0 + d_y[i] < g.H && x0 + d_x[i] < g.W;     // pixels past the map contribute nothing
      const unsigned off = ok ? (unsigned)((((size_t)(n * g.H + y0 + d_y[i]) * g.W + x0 + d_x[i]) * a.Cout +
                                            cb * 64 + j * 8) * 2) : OOB_OFFSET;
      dv[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int k = 0; k < PLD; ++k) {
      const int pr = row0 + 32 * k;
      if (pr < PROWS) {
        u32x4 v = pv[k];
        if (BNIN && ((pvalid >> k) & 1u)) {   // padding stays zero
          const float* p = sBn + bn_gi * 128 + j * 8;
          v = bn_relu_vec(v, *reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4),
                          *reinterpret_cast<const f32x4*>(p + 64), *reinterpret_cast<const f32x4*>(p + 68));
        }
        st16(&sP[pr * RS + j * 8], v);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st16(&sD[(row0 + 32 * i) * RS + j * 8], dv[i]);
  };

  // fragment geometry: MFMA k index (8*lq + 4*h + e) of k-step ks <-> tile pixel
  //   p = 32*ks + 16*h + 4*lq + e        (e = lr>>2: the pixel row this lane points at)
  // so a transpose read (fixed h) covers 16 consecutive pixels and everything except the lane's
  // own (lq, e, channel run) is a compile-time constant
  const int pl = 4 * lq + (lr >> 2);                       // 0..15
  const int chq = (lr & 3) * 4;                            // this lane's 4-channel run
  const int d_base = pl * RS + wn * 32 + chq;
  const int x_base = (SMALLW ? (pl >> 3) * PW + (pl & 7) : pl) * RS + wm * 32 + chq;
  auto xrow = [](int ks, int h) {                          // patch row of pixel 32ks+16h (tap 0,0), constexpr-foldable
    const int p = 32 * ks + 16 * h;
    return SMALLW ? (p >> 6) * (PH * PW) + ((p >> 3) & 7) * PW : (p >> 4) * PW;
  };

  if (BNIN) {   // up to 8 groups x {scale, shift} x this chunk's 64 channels
    const int ngroups = min(8, (g.N + a.in_npg - 1) / a.in_npg);
    for (int i = t; i < ngroups * 128; i += 256) {
      const int gi = i >> 7, r = (i >> 6) & 1, c = i & 63;
      sBn[i] = a.in_bnp[(size_t)gi * 4 * g.C + r * g.C + cc * 64 + c];
    }
    __syncthreads();
  }
  f32x4 acc[9][2][2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tp][tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int t_begin = split * tiles_per_split;
  const int t_end = min(ntiles, t_begin + tiles_per_split);
  if (t_begin < t_end) {
    load_tile(t_begin);
    store_tile();
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
      const bool more = tile + 1 < t_end;
      if (more) load_tile(tile + 1);
      // the two-image tile needs ~20 more address / staging registers: fully unrolled it only fits one
      // wave per SIMD (472 registers) and ran at 640 TFLOP/s; rolled k-steps keep it at two waves
#pragma unroll(SMALLW ? 1 : 4)
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 bfr[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          bfr[tn] = tr_frag(sD + d_base, (32 * ks) * RS + tn * 16, (32 * ks + 16) * RS + tn * 16);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int shift = (tp / 3) * PW + (tp % 3);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            const bf16x8 af = tr_frag(sP + x_base, (xrow(ks, 0) + shift) * RS + tm * 16, (xrow(ks, 1) + shift) * RS + tm * 16);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[tn], acc[tp][tm][tn], 0, 0, 0);
          }
        }
      }
      __syncthreads();            // everyone is done reading this tile
      if (more) {
        store_tile();
        __syncthreads();
      }
    }
  }

  // D[cin][cout]: lane holds 4 consecutive cin of cout = lane&15 -> one 16-byte store
  if (!a.tickets) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int cout = cb * 64 + wn * 32 + tn * 16 + lr;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const int kc = tp * g.C + cc * 64 + wm * 32 + tm * 16 + lq * 4;
          *reinterpret_cast<f32x4*>(a.partial + ((size_t)split * a.Cout + cout) * g.Ktot + kc) = acc[tp][tm][tn];
        }
      }
    return;
  }
  // in-launch split-K reduction (vfs_wgrad_tail.h): partials as register images (36 pieces per thread); the last split of this
  // (cin chunk, cout block) tile to arrive sums all of them in split order, one tap (four pieces) at a time
  const int tile = cb * nchunk + cc, nwt = nchunk * ncb;
  const __amdgpu_buffer_rsrc_t prs = wgt_partial_rsrc(a, nwt, 36);
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) wgt_store_piece(prs, wgt_piece_off(split, tile, nwt, 36, tp * 4 + tn * 2 + tm), acc[tp][tm][tn]);
  if (!wgt_last_arriver(a, tile)) return;
  const unsigned sstride = (unsigned)((size_t)nwt * 36 * 4096);
#pragma unroll 1
  for (int tp = 0; tp < 9; ++tp) {
    unsigned off[4];
    f32x4 sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) off[i] = wgt_piece_off(0, tile, nwt, 36, tp * 4 + i);
    wgt_sum_splits<4>(a, prs, sstride, off, sum);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int cout = cb * 64 + wn * 32 + tn * 16 + lr, kc = tp * g.C + cc * 64 + wm * 32 + tm * 16 + lq * 4;
        wgt_add_grad(a, cout, kc, sum[tn * 2 + tm]);
      }
  }
}

bool vfs_wgrad_halo_eligible(const WgradArgs& a, int mode) {
  const ConvGeom& g = a.g;
  if (mode != GATHER_FWD || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return false;
  if (g.H != g.Ho || g.W != g.Wo || g.C % 64 || a.Cout % 64) return false;
  if ((size_t)g.N * g.H * g.W * g.C * 2 >= 0xFFFFFFF0ull || (size_t)g.N * g.H * g.W * a.Cout * 2 >= 0xFFFFFFF0ull) return false;
  if (vfs_small_map(g.H, g.W)) return true;
  // 8x16 tiles, ragged at the right / bottom edge when the map is not a multiple (zero dY rows), mostly full
  const long long cover = (long long)((g.H + 7) / 8 * 8) * ((g.W + 15) / 16 * 16);
  return (long long)g.H * g.W * 100 >= cover * vfs_option_halo_min_fill;
}

int vfs_wgrad_halo_dispatch(const WgradArgs& a, hipStream_t stream, int* eff_nsplit) {
  const bool smallw = vfs_small_map(a.g.H, a.g.W);
  const int TW = smallw ? 8 : 16, TI = smallw ? 2 : 1;
  const int ntiles = ((a.g.N + TI - 1) / TI) * ((a.g.H + 7) / 8) * ((a.g.W + TW - 1) / TW);
  const int nsplit = a.nsplit < ntiles ? a.nsplit : ntiles;
  const int tps = (ntiles + nsplit - 1) / nsplit;
  WgradArgs b = a;
  b.nsplit = (ntiles + tps - 1) / tps;
  *eff_nsplit = b.nsplit;
  const int blocks = (a.g.C >> 6) * (a.Cout >> 6) * b.nsplit;
  b.xcd_swizzle = vfs_option_wgrad_xcd && (a.g.C >> 6) * (a.Cout >> 6) > 1 && blocks >= 16;
  if (a.in_bnp && (a.g.N + a.in_npg - 1) / a.in_npg > 8) return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: more than 8 BatchNorm groups");
  if (b.tickets && (!b.grad || (a.g.C >> 6) * (a.Cout >> 6) > VFS_WGRAD_TICKETS || (size_t)b.nsplit * a.Cout * a.g.Ktot * 4 >= 0xFFFFFFF0ull))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: the in-launch reduction takes a gradient, at most 4096 tiles and < 4 GiB of partials");
  if (smallw) {
    if (a.in_bnp) hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, b, tps, ntiles);
    else hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, b, tps, ntiles);
  } else {
    if (a.in_bnp) hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, b, tps, ntiles);
    else hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, b, tps, ntiles);
  }
  return vfs_check_launch("conv3x3_wgrad_halo");
}

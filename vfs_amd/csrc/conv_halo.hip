// 3x3 / stride-1 / pad-1 convolution (forward and dgrad) with the input HALO TILE resident in LDS.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-fetches a pixel's 128-byte channel chunk
// once per tap: 9x the L1/L2 traffic and 9x the LDS stores of the activation operand, which on
// MI355X leaves the 64..128-channel layers bound by the vector-memory / LDS-store path, not by
// MFMA (measured 390-820 TFLOP/s).  Here one workgroup owns a SPATIAL tile of 128 output pixels
// (8 rows x 16 columns of one image, or two whole 8x8 images), stages the (TH+2)x(TW+2) halo
// patch of a 64-channel chunk ONCE, and the nine taps read it with shifted row indices:
//     B-fragment row of tap (r,s), pixel (py,px)  =  patch[(py + r') * (TW+2) + px + s']
// (r' = r forward, 2-r dgrad).  Per tap only the BC x 64 weight tile is fetched.  Same MFMA
// tiling (v_mfma_f32_16x16x32_bf16, rows = channels, cols = pixels, wave = (BC/2) x 64), same
// swizzled LDS rows and the same epilogue (bias / residual add / BatchNorm partial statistics)
// as conv_igemm.hip.
#include "vfs_conv.h"

// any offset >= num_records reads as zero; 2^31 leaves room for a scalar offset on top without wrapping
#define OOB_OFFSET 0x80000000u

// LDS rows are 64 bf16 + 16 pad = 160 bytes (32 B x odd): the ds_read_b128 fragment reads of 16
// consecutive rows are bank-conflict free on gfx950 and, being linear, every tap / k-step / MFMA
// tile is an IMMEDIATE offset from one per-lane base register (no address VALU in the main loop).
#define HALO_RS 80

// lane column lr of 16-pixel group tn of half wp -> pixel of the spatial tile.  For the 8-wide tile a
// group is two 8-pixel rows, 10 patch rows apart; the second row is permuted so that the patch rows of
// each ds_read_b128 lane group stay distinct mod 8 (conflict free).
template <bool SMALLW>
__device__ __forceinline__ void halo_pixel(int wp, int tn, int lr, int& ti, int& py, int& px) {
  if (SMALLW) {
    const int c = lr & 7;
    ti = wp;
    py = tn * 2 + (lr >> 3);
    px = lr < 8 ? c : (c < 2 ? c : (c < 4 ? c + 4 : c - 2));
  } else {
    ti = 0;
    py = wp * 4 + tn;
    px = lr;
  }
}

template <int BC, bool DGRAD, bool SMALLW>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(ConvArgs a) {
  constexpr int TW = SMALLW ? 8 : 16, TH = 8, TI = SMALLW ? 2 : 1;
  constexpr int PW = TW + 2, PH = TH + 2;
  constexpr int PROWS = TI * PH * PW;            // 180 or 200 patch rows of 64 channels
  constexpr int PLD = (PROWS * 8 + 255) / 256;   // 16-byte patch loads per thread (6 or 7)
  constexpr int WC = BC / 2, TM = WC / 16, TN = 4, WLD = BC / 32;
  constexpr int RS = HALO_RS;
  __shared__ __attribute__((aligned(16))) bf16_t sP[PROWS * RS];
  __shared__ __attribute__((aligned(16))) bf16_t sW[2 * BC * RS];
  __shared__ float sRed[2][BC][2];

  const ConvGeom g = a.g;                        // FWD: H,W,C = input; DGRAD: H,W,C = dY (same H,W)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wc = wave >> 1, wp = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int ncb = a.Cout / BC;
  const int tile = blockIdx.x / ncb, cb = blockIdx.x - tile * ncb;
  const int c0 = cb * BC;
  const int tiles_x = g.W / TW, tiles_y = g.H / TH;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tn0 = (tile / (tiles_x * tiles_y)) * TI;
  const int y0 = ty * TH, x0 = tx * TW;
  const int j = t & 7, row0 = t >> 3;
  const int nchunk = g.C >> 6;

  // ---- patch loads: slot k covers patch row (t>>3) + 32k, chunk j; offset fixed for the block
  unsigned poff[PLD];
#pragma unroll
  for (int k = 0; k < PLD; ++k) {
    const int pr = row0 + 32 * k;
    unsigned off = OOB_OFFSET;
    if (pr < PROWS) {
      const int ti = pr / (PH * PW), rem = pr - ti * (PH * PW);
      const int py = rem / PW, px = rem - py * PW;
      const int y = y0 - 1 + py, x = x0 - 1 + px, n = tn0 + ti;
      if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W && n < g.N)
        off = (unsigned)((((size_t)(n * g.H + y) * g.W + x) * g.C + j * 8) * 2);
    }
    poff[k] = off;
  }
  // weight rows c0 + row0 + 32 i, chunk j (Cout % BC == 0: never out of range)
  const unsigned wbase = (unsigned)(((size_t)(c0 + row0) * g.Ktot + j * 8) * 2);
  const unsigned wrow32 = (unsigned)(32 * g.Ktot * 2);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.src, 0, (unsigned)((size_t)g.N * g.H * g.W * g.C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.wgt, 0, (unsigned)((size_t)a.Cout * g.Ktot * 2), 0x00020000);

  u32x4 pr_[PLD], wr[WLD];
  auto load_patch = [&](int cc) {
#pragma unroll
    for (int k = 0; k < PLD; ++k) pr_[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, poff[k], cc * 128, 0);
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int k = 0; k < PLD; ++k) {
      const int pr = row0 + 32 * k;
      if (pr < PROWS) st16(&sP[pr * RS + j * 8], pr_[k]);
    }
  };
  auto load_w = [&](int cc, int tap) {
    const int wcol = (tap * g.C + cc * 64) * 2;
#pragma unroll
    for (int i = 0; i < WLD; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wbase + i * wrow32, wcol, 0);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) st16(&sW[buf * (BC * RS) + (row0 + 32 * i) * RS + j * 8], wr[i]);
  };

  // ---- per-lane fragment bases: everything else in the main loop is an immediate offset
  int pbase[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int ti, py, px;
    halo_pixel<SMALLW>(wp, tn, lr, ti, py, px);
    pbase[tn] = (ti * (PH * PW) + py * PW + px) * RS + lq * 8;
  }
  const int abase = (wc * WC + lr) * RS + lq * 8;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_patch(0);
  load_w(0, 0);
  store_patch();
  store_w(0);
  __syncthreads();
  int wbuf = 0;
  for (int cc = 0; cc < nchunk; ++cc) {
    const bool more_chunks = cc + 1 < nchunk;
    if (more_chunks) load_patch(cc + 1);          // in flight during the nine taps of this chunk
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const bool last_tap = tap == 8;
      const bool more = !last_tap || more_chunks;
      if (more) load_w(last_tap ? cc + 1 : cc, last_tap ? 0 : tap + 1);
      const int r = tap / 3, s = tap - 3 * r;
      const int shift = DGRAD ? (2 - r) * PW + (2 - s) : r * PW + s;
      const bf16_t* wsrc = sW + wbuf * (BC * RS) + abase;
      // one 64-deep K-step: A = weights (rows wc*WC..), B = patch rows shifted by the tap
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          af[tm] = *reinterpret_cast<const bf16x8*>(wsrc + tm * 16 * RS + kk * 32);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          bfr[tn] = *reinterpret_cast<const bf16x8*>(&sP[pbase[tn] + shift * RS + kk * 32]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
      }
      if (last_tap && more_chunks) {
        __syncthreads();                           // every wave is done with this chunk's patch
        store_patch();
      }
      if (more) store_w(wbuf ^ 1);
      __syncthreads();
      wbuf ^= 1;
    }
  }

  // ---------------- epilogue (as conv_igemm.hip) ----------------
  const bool do_stats = a.stats != nullptr;
  float s1[TM][4], s2[TM][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[tm][r] = 0.f; s2[tm][r] = 0.f; }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int ti, py, px;
    halo_pixel<SMALLW>(wp, tn, lr, ti, py, px);
    const int n = tn0 + ti;
    const bool mok = n < g.N;
    const size_t mdst = ((size_t)n * g.H + (y0 + py)) * g.W + (x0 + px);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int c = c0 + wc * WC + tm * 16 + lq * 4;
      if (mok) {
        float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
        if (a.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += a.bias[c + r];
        }
        const size_t o = mdst * a.Cout + c;
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
    if (t < BC) {
      float* dst = a.stats + (size_t)tile * 2 * a.Cout;   // one partial per spatial tile (128 pixels)
      dst[c0 + t] = sRed[0][t][0] + sRed[1][t][0];
      dst[a.Cout + c0 + t] = sRed[0][t][1] + sRed[1][t][1];
    }
  }
}

template <int BC, bool DGRAD, bool SMALLW>
static int launch_halo(const ConvArgs& a, hipStream_t stream) {
  const int TW = SMALLW ? 8 : 16, TI = SMALLW ? 2 : 1;
  const int tiles = ((a.g.N + TI - 1) / TI) * (a.g.H / 8) * (a.g.W / TW);
  const int ncb = a.Cout / BC;
  hipLaunchKernelGGL((conv3x3_halo_kernel<BC, DGRAD, SMALLW>), dim3(tiles * ncb), dim3(256), 0, stream, a);
  return vfs_check_launch("conv3x3_halo");
}

// eligibility: 3x3 / stride 1 / pad 1, H % 8 == 0 and (W % 16 == 0, or W == 8 with N even so that
// the 128-pixel statistics blocks coincide with the generic kernel's)
bool vfs_conv_halo_eligible(const ConvArgs& a, int mode) {
  const ConvGeom& g = a.g;
  if (mode != GATHER_FWD && mode != GATHER_DGRAD) return false;
  if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return false;
  if (a.Cout % 64 || g.C % 64) return false;
  if (g.H != g.Ho || g.W != g.Wo || g.H % 8) return false;
  if (g.W % 16 == 0) return true;
  return g.W == 8 && g.H == 8 && g.N % 2 == 0;
}

int vfs_conv_halo_dispatch(const ConvArgs& a, int mode, hipStream_t stream) {
  const bool wide = (a.Cout % 128 == 0), smallw = a.g.W == 8, dg = mode == GATHER_DGRAD;
  if (wide) {
    if (dg) return smallw ? launch_halo<128, true, true>(a, stream) : launch_halo<128, true, false>(a, stream);
    return smallw ? launch_halo<128, false, true>(a, stream) : launch_halo<128, false, false>(a, stream);
  }
  if (dg) return smallw ? launch_halo<64, true, true>(a, stream) : launch_halo<64, true, false>(a, stream);
  return smallw ? launch_halo<64, false, true>(a, stream) : launch_halo<64, false, false>(a, stream);
}

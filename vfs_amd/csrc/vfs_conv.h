// Geometry + gather shared by the implicit-GEMM convolution kernels.
//
// GEMM view (all three directions):   D[chan][pixel] = sum_k  Wt[chan][k] * G[pixel][k]
//   * pixel   : m = (n*Ho + ho)*Wo + wo  over the DESTINATION grid (Ho x Wo)
//   * k       : 64-element K-steps; K-step kt, 16-byte chunk j (8 bf16) of a pixel's
//               gathered row come from ONE contiguous 16-byte run of the NHWC source
//   * G       : never materialised (im2col on the fly), zero where the tap is padding
// Gather modes
//   FWD   : source = input activations [N,H,W,C], k = (r, s, c)           (conv forward)
//   DGRAD : source = dY [N,H,W,C=Cout], k = (r, s, cout), tap valid iff
//           (ho+pad-r) and (wo+pad-s) are multiples of stride             (conv dgrad)
//   STEM  : 7x7 stride-2 pad-3 conv on NHWC4 input (channel 3 is zero padding);
//           K-step kt covers kernel rows 2kt,2kt+1; chunk j = row (j>>2), column pair (j&3)
//           starting at column 2wo-4 (tap s=-1 and row 7 carry zero weights): K = 8*8*4 = 256
#pragma once
#include "vfs_common.h"

enum { GATHER_FWD = 0, GATHER_DGRAD = 1, GATHER_STEM = 2, GATHER_DGRAD2 = 3 };

struct ConvGeom {
  int N, H, W, C;   // gather-source tensor (NHWC), C = physical channels per pixel
  int Ho, Wo;       // destination pixel grid
  int KH, KW, stride, pad;
  int Ktot;         // GEMM K (multiple of 64)
  int M;            // N*Ho*Wo
  int dil = 1;      // tap spacing (forward gather of the implicit-GEMM kernel only: frozen dilated backbones)
};

// BatchNorm-backward statistics fused into the epilogue of the dgrad that PRODUCES the gradient g (the
// conv's output): per 128-pixel block  S1[c] = sum g*mask,  S2[c] = sum g*mask*xhat  with
// xhat = (x - mean)*invstd of the BatchNorm unit whose output this gradient belongs to - what
// bn_bwd_reduce_kernel computes in a separate pass over g and x.  mask: y > 0 (residual units, y given),
// x*scale+shift > 0 (plain conv-BN-ReLU units, relu = 1), or none.  partial == nullptr: disabled.
struct BnBwdFuse {
  const bf16_t* x;     // [M][Cout] raw conv output of that unit
  const bf16_t* y;     // [M][Cout] its activation (mask) or null; relu == VFS_MASK_BITS: bit-packed mask, slab-major uint8 [Cout/64][M][8] (vfs_common.h mask8_index)
  const float* bnp;    // [G][4][Cout] scale, shift, mean, invstd
  float* partial;      // [ceil(M/128)][2][Cout]
  int mpg;             // pixels per statistics group (selects the bnp row; blocks never straddle groups)
  int relu;
};

struct ConvArgs {
  ConvGeom g;
  const bf16_t* src;   // gather source
  const bf16_t* wgt;   // [Cout][Ktot] bf16, K contiguous, K ordered as the gather mode defines
  bf16_t* out;         // [M][Cout]
  const bf16_t* add;   // optional [M][Cout] (may alias out)
  const float* bias;   // optional [Cout]
  float* stats;        // optional [num_pixel_blocks][2][Cout]
  int Cout;
  BnBwdFuse bn;        // optional fused BatchNorm-backward statistics of the OUTPUT (dgrad only)
  // optional BatchNorm-apply + ReLU of the INPUT, folded into the operand load (forward, halo kernel): src
  // is the RAW output of a plain conv-BN-ReLU unit, in_bnp = its float[G][4][C] {scale, shift, ..},
  // in_npg = images per statistics group.  The activation tensor is never written or read.
  const float* in_bnp;
  int in_npg;
  // optional split-K (implicit-GEMM kernel, small pixel counts: the SimSiam head's Linear layers): the K loop
  // is cut into ksplit slices (blockIdx.z); ks_ws = unsigned tickets[KS_TICKETS] (zero before the first launch,
  // left at zero) followed by float partial tiles [tile][slice][...]; the LAST slice to arrive sums the
  // partials in slice order (deterministic) and runs the epilogue.
  float* ks_ws = nullptr;
  int ksplit = 1;
  // optional: `add` is gated by the bit-packed ReLU mask of its own tensor (slab-major, vfs_common.h mask8_index; add_rows =
  // its pixel count): out = conv + add * (mask bit).  The identity branch of a residual block then adds the block-output
  // gradient g itself and the masked copy g * (y > 0) is never written (Cout % 64 == 0).
  const unsigned char* add_mask = nullptr;
  long long add_rows = 0;
  // optional COARSE statistics rows (round 5): the launch's workgroups also sum their rows in groups of 2^coarse_log2 - the last
  // workgroup of a group to arrive (device-scope ticket) adds the group's rows in row order (deterministic) into
  // stats_coarse [ceil(rows / 2^L)][2][Cout] - so that the consumer (vfs_bn_act_fin: statistics finished in its prologue) sees
  // at most 128 rows per group and the separate reduction launch between the convolution and its BatchNorm disappears.
  // stats_tickets: unsigned [groups x channel blocks], zero before the first launch, left at zero.
  float* stats_coarse = nullptr;
  unsigned* stats_tickets = nullptr;
  int coarse_log2 = 0;
  int mfma_stats = 0;    // implicit-GEMM kernel: forward statistics rows by MFMA from the staged bf16 tile (set by the dispatcher)
  int xcd_swizzle = 0;   // implicit-GEMM kernel: XCD-aware logical tile order (set by the dispatcher)
};
#define KS_TICKETS 1024

struct WgradArgs {
  ConvGeom g;          // geometry of the FORWARD conv (gather source = forward input)
  const bf16_t* dy;    // [M][Cout] gradient of the raw conv output
  const bf16_t* x;     // forward input activations (gather source)
  float* partial;      // [nsplit][Cout][Ktot] fp32 partial sums
  int Cout;
  int pix_per_split;   // multiple of 64
  int nsplit;
  const float* in_bnp = nullptr;      // optional: x is a RAW conv output, relu(x*scale+shift) is applied while staging (halo kernel; 1x1 staged kernel)
  int in_npg = 0;
  int xcd_swizzle = 0; // generic kernel: XCD-aware logical block order (set by the dispatcher)
  // optional in-launch split-K reduction (round 6, vfs_wgrad_tail.h): the last workgroup of a (k-column, cout) tile to arrive sums the
  // tile's partials in split order and adds them to grad (reference OIHW layout); tickets: unsigned[VFS_WGRAD_TICKETS], zero before
  // the first launch, left at zero.  tickets == nullptr: partials only (the caller launches wgrad_reduce).
  float* grad = nullptr;
  unsigned* tickets = nullptr;
};

// relu(x*scale + shift) on one 16-byte vector (8 channels), rounded to bf16 exactly as bn_act_kernel does
__device__ __forceinline__ u32x4 bn_relu_vec(u32x4 v, const f32x4& sc0, const f32x4& sc1, const f32x4& sh0, const f32x4& sh1) {
  float x[8];
  unpack8(v, x);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] = fmaxf(x[i] * sc0[i] + sh0[i], 0.f);
    x[4 + i] = fmaxf(x[4 + i] * sc1[i] + sh1[i], 0.f);
  }
  return pack8(x);
}

int vfs_conv_igemm_dispatch(const ConvArgs& a, int mode, hipStream_t stream);

// Tail of a forward epilogue with coarse statistics rows: this workgroup has WRITTEN its fine rows (agent-scope stores, channels
// [c0, c0 + nch)); it takes a ticket of its group, and the last of the group's `arrivals` workgroups adds the group's fine rows
// [row0, row0 + nrow) in row order into coarse row `grp`.  Called by all 256 threads.
// vfs_stats_coarsen_finish: the same with the ticket already drawn by thread 0 (issued early: its round trip ran under other work).
__device__ __forceinline__ void vfs_stats_coarsen_finish(const ConvArgs& a, unsigned ticket_of_t0, int grp, int row0, int nrow, int arrivals, int ticket_idx,
                                                         int c0, int nch);
__device__ __forceinline__ void vfs_stats_coarsen_tail(const ConvArgs& a, int grp, int row0, int nrow, int arrivals, int ticket_idx, int c0, int nch) {
  vfs_release_workgroup();      // this wave's row stores have been performed
  __syncthreads();
  unsigned tk = 0;
  if (threadIdx.x == 0) tk = vfs_ticket_agent(&a.stats_tickets[ticket_idx]);
  vfs_stats_coarsen_finish(a, tk, grp, row0, nrow, arrivals, ticket_idx, c0, nch);
}
__device__ __forceinline__ void vfs_stats_coarsen_finish(const ConvArgs& a, unsigned ticket_of_t0, int grp, int row0, int nrow, int arrivals, int ticket_idx,
                                                         int c0, int nch) {
  __shared__ unsigned s_cticket;
  const int t = threadIdx.x;
  if (t == 0) s_cticket = ticket_of_t0;
  __syncthreads();
  if (s_cticket != (unsigned)arrivals - 1u) return;
  if (t == 0) vfs_store_agent(&a.stats_tickets[ticket_idx], 0u);
  for (int e = t; e < 2 * nch; e += 256) {
    const int st = e / nch, c = c0 + e - st * nch;
    const float* p = a.stats + (size_t)row0 * 2 * a.Cout + (size_t)st * a.Cout + c;
    float sum = 0.f;
    for (int j = 0; j < nrow; ++j) sum += vfs_load_agent(p + (size_t)j * 2 * a.Cout);
    a.stats_coarse[(size_t)grp * 2 * a.Cout + (size_t)st * a.Cout + c] = sum;
  }
}
// conv_pw.hip: persistent producer / consumer kernel for pure-GEMM (1x1, stride 1) problems
bool vfs_conv_pw_eligible(const ConvArgs& a, int mode);
int vfs_conv_pw_dispatch(const ConvArgs& a, int mode, hipStream_t stream);
extern int vfs_option_igemm_pw, vfs_option_igemm_pw_min_tiles, vfs_option_igemm_skinny;
bool vfs_conv_skinny_eligible(const ConvArgs& a, int mode);      // M <= 128 rows: the head's Linear layers
int vfs_conv_skinny_dispatch(const ConvArgs& a, hipStream_t stream);
bool vfs_conv_halo_eligible(const ConvArgs& a, int mode);
// maps of at most 8x8 pixels that fill most of an 8x8 tile (8x8, 7x7 with the default 70 %): the halo kernels take
// two whole images per workgroup
extern int vfs_option_halo_min_fill, vfs_option_halo_xcd, vfs_option_halo_deep_max;
static inline bool vfs_small_map(int H, int W) { return H <= 8 && W <= 8 && H * W * 100 >= 64 * vfs_option_halo_min_fill; }
int vfs_conv_halo_dispatch(const ConvArgs& a, int mode, hipStream_t stream);
bool vfs_wgrad_halo_eligible(const WgradArgs& a, int mode);
int vfs_wgrad_halo_dispatch(const WgradArgs& a, hipStream_t stream, int* eff_nsplit);
int vfs_stem_tiles(int N, int Ho, int Wo);
int vfs_stem_fwd_direct_launch(const ConvArgs& a, hipStream_t stream);
extern int vfs_option_stem_direct;
extern int vfs_option_stem_blocks;   // grid cap of the direct stem kernel (0 = default 2048); tests walk many tiles per block
extern int vfs_option_halo;   // 1: 3x3/s1 convs use the halo-tile kernel (capi: vfs_set_option)
int vfs_conv_wgrad_dispatch(const WgradArgs& a, int mode, hipStream_t stream);

struct PixCoord {
  int nH;  // n*H of the source tensor
  int hb;  // base source row   (FWD/STEM: ho*stride-pad ; DGRAD: ho+pad) ; <<0 when pixel invalid
  int wb;  // base source column
};

template <int MODE>
__device__ __forceinline__ PixCoord pix_decode(const ConvGeom& g, int m) {
  PixCoord pc;
  if (m >= g.M) {
    pc.nH = 0; pc.hb = -(1 << 24); pc.wb = -(1 << 24);
    return pc;
  }
  int hw = g.Ho * g.Wo;
  int n = m / hw;
  int rem = m - n * hw;
  int ho = rem / g.Wo;
  int wo = rem - ho * g.Wo;
  pc.nH = n * g.H;
  if (MODE == GATHER_DGRAD) {
    pc.hb = ho + g.pad; pc.wb = wo + g.pad;
  } else if (MODE == GATHER_STEM) {
    pc.hb = 2 * ho - 3; pc.wb = 2 * wo - 4;
  } else {
    pc.hb = ho * g.stride - g.pad; pc.wb = wo * g.stride - g.pad;
  }
  return pc;
}

struct KStep {  // decoded K-step (uniform per workgroup)
  int r, s, c0;
};

template <int MODE>
__device__ __forceinline__ KStep kstep_decode(const ConvGeom& g, int kt) {
  KStep ks;
  if (MODE == GATHER_STEM) {
    ks.r = 2 * kt; ks.s = 0; ks.c0 = 0;
  } else {
    int cpt = g.C >> 6;
    int tap = kt / cpt;
    ks.c0 = (kt - tap * cpt) << 6;
    ks.r = tap / g.KW;
    ks.s = tap - ks.r * g.KW;
  }
  return ks;
}

// 16-byte chunk j of pixel pc at K-step ks; zero when the tap falls on padding
template <int MODE>
__device__ __forceinline__ u32x4 gather16(const ConvGeom& g, const bf16_t* __restrict__ src,
                                          const PixCoord& pc, const KStep& ks, int j) {
  int hi, wi;
  bool ok;
  size_t off;
  if (MODE == GATHER_STEM) {
    int r = ks.r + (j >> 2);
    hi = pc.hb + r;
    wi = pc.wb + 2 * (j & 3);
    ok = (r < 7) && ((unsigned)hi < (unsigned)g.H) && ((unsigned)wi < (unsigned)g.W);
    off = ((size_t)(pc.nH + hi) * g.W + wi) * 4;
  } else if (MODE == GATHER_DGRAD) {
    int th = pc.hb - ks.r, tw = pc.wb - ks.s;
    ok = (th >= 0) && (tw >= 0);
    if (g.stride == 1) {
      hi = th; wi = tw;
    } else {
      hi = th / g.stride; wi = tw / g.stride;
      ok = ok && (hi * g.stride == th) && (wi * g.stride == tw);
    }
    ok = ok && (hi < g.H) && (wi < g.W);
    off = ((size_t)(pc.nH + hi) * g.W + wi) * g.C + ks.c0 + j * 8;
  } else {
    hi = pc.hb + ks.r;
    wi = pc.wb + ks.s;
    ok = ((unsigned)hi < (unsigned)g.H) && ((unsigned)wi < (unsigned)g.W);
    off = ((size_t)(pc.nH + hi) * g.W + wi) * g.C + ks.c0 + j * 8;
  }
  return ok ? ld16(src + off) : zero16();
}

// LDS tile: rows of 64 bf16 (128 B); the eight 16-byte chunks of a row are XOR-swizzled so
// that both the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 fragment groups
// (16 consecutive rows, same chunk) hit 16 distinct 16-byte slots of the 64-bank row.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}
// variant for tiles filled by the transposing wgrad stage (rows written with stride 8)
__device__ __forceinline__ int lds_off_t(int row, int chunk) {
  return row * 64 + ((chunk ^ (((row >> 1) ^ (row >> 4)) & 7)) << 3);
}

// ---- fused BatchNorm-backward statistics (BnBwdFuse): per-lane pieces used by the epilogues.
// A lane owns one 8-channel chunk for all the pixel rows it stores; coefficients stay in registers.
struct BnFuseLane {
  float sc[8], sh[8], mean[8], inv[8], s1[8], s2[8];
};
__device__ __forceinline__ void bnfuse_init(BnFuseLane& L, const BnBwdFuse& bn, int Cout, int gi, int c) {
  const float* bp = bn.bnp + (size_t)gi * 4 * Cout + c;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(bp + 4 * h), b = *reinterpret_cast<const f32x4*>(bp + Cout + 4 * h);
    const f32x4 m = *reinterpret_cast<const f32x4*>(bp + 2 * Cout + 4 * h), v = *reinterpret_cast<const f32x4*>(bp + 3 * Cout + 4 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) { L.sc[4 * h + i] = a[i]; L.sh[4 * h + i] = b[i]; L.mean[4 * h + i] = m[i]; L.inv[4 * h + i] = v[i]; }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { L.s1[i] = 0.f; L.s2[i] = 0.f; }
}
// gv: the 8 gradient values just stored for one pixel; xv / yv: the unit's raw output / activation at the
// same place (loaded by the caller EARLY: issued next to the use, each load costs a full HBM round trip
// per tile and the fusion is no faster than the separate reduction pass)
// 8 pixels x 16 channels of a pixel-major bf16 tile, transposed by the LDS transpose read: lane gets the 8-pixel MFMA k-run
// of channel (lane & 15) (as wg_tr_frag of conv_wgrad.hip)
typedef __attribute__((ext_vector_type(4))) short vfs_s16x4;
__device__ __forceinline__ bf16x8 tile_tr_frag(const bf16_t* tile, int lo_off, int hi_off) {
  const vfs_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) vfs_s16x4*)(tile + lo_off));
  const vfs_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) vfs_s16x4*)(tile + hi_off));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
// mask bits of the NCH (64 or 32) channels a wave owns at pixel row m of the `add` tensor: bit k <-> channel cw + k
template <int NCH>
__device__ __forceinline__ unsigned long long addmask_word(const unsigned char* bits, long long m, int cw, long long M, int C) {
  const unsigned char* p = bits + mask8_index(m, cw, M, C);
  if (NCH == 64) return *reinterpret_cast<const unsigned long long*>(p);      // cw % 64 == 0: one aligned slab row
  return (unsigned long long)*reinterpret_cast<const unsigned*>(p);           // cw % 32 == 0
}
// the ReLU-mask byte of pixel m, channels c..c+7 of the [M][Cout] output: from the bit-packed mask, or from the activation
__device__ __forceinline__ unsigned bnfuse_load_mask(const BnBwdFuse& bn, long long m, int c, long long M, int Cout) {
  if (bn.relu == VFS_MASK_BITS) return mask8_load(bn.y, m, c, M, Cout);
  return mask8_of(*reinterpret_cast<const u32x4*>(bn.y + (size_t)m * Cout + c));
}
__device__ __forceinline__ void bnfuse_accum(BnFuseLane& L, const BnBwdFuse& bn, u32x4 gv, u32x4 xv, unsigned ymask) {
  // ymask: bit i <-> element i of the unit's activation is positive (residual units: bn.y given, as bits or as the tensor)
#ifdef BNFUSE_WHATIF      // timing only (WRONG statistics): the epilogue without the per-element statistics work
  L.s1[0] += __builtin_bit_cast(float, gv.x ^ xv.x ^ ymask);
  return;
#endif
  float g[8], x[8];
  unpack8(gv, g);
  unpack8(xv, x);
  if (bn.y) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = ((ymask >> i) & 1u) ? g[i] : 0.f;
  } else if (bn.relu) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (x[i] * L.sc[i] + L.sh[i] > 0.f) ? g[i] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    L.s1[i] += g[i];
    L.s2[i] += g[i] * ((x[i] - L.mean[i]) * L.inv[i]);
  }
}
// lane -> LDS [wave][pixel group][stat][WCH channels]; the caller syncs and sums groups / pixel waves
__device__ __forceinline__ void bnfuse_spill(const BnFuseLane& L, float* sB, int wave, int grp, int ngrp, int wch, int ch) {
  float* d = sB + ((size_t)(wave * ngrp + grp) * 2) * wch + ch * 8;
  *reinterpret_cast<f32x4*>(d) = (f32x4){L.s1[0], L.s1[1], L.s1[2], L.s1[3]};
  *reinterpret_cast<f32x4*>(d + 4) = (f32x4){L.s1[4], L.s1[5], L.s1[6], L.s1[7]};
  *reinterpret_cast<f32x4*>(d + wch) = (f32x4){L.s2[0], L.s2[1], L.s2[2], L.s2[3]};
  *reinterpret_cast<f32x4*>(d + wch + 4) = (f32x4){L.s2[4], L.s2[5], L.s2[6], L.s2[7]};
}

// The same K-step with ALL fragment reads (both 32-deep halves) issued before the first MFMA: the compiler's default
// order re-uses a minimal set of fragment registers and so waits for an LDS round trip ~6 times per K-step - hidden by
// other waves at 3-4 waves per SIMD, fully exposed in the one-workgroup-per-CU DMA-ring variant (ISA inspection,
// tools: hipcc -S).  Costs 32 more VGPRs.  Same MFMA order, bit-identical results.
template <int TM, int TN>
__device__ __forceinline__ void mma_kstep_upfront(const bf16_t* __restrict__ sA, const bf16_t* __restrict__ sB,
                                                  int rowA0, int rowB0, int lane, f32x4 (&acc)[TM][TN]) {
  const int lr = lane & 15, lq = lane >> 4;
  bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) af[kk][tm] = *reinterpret_cast<const bf16x8*>(sA + lds_off(rowA0 + tm * 16 + lr, kk * 4 + lq));
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bfr[kk][tn] = *reinterpret_cast<const bf16x8*>(sB + lds_off(rowB0 + tn * 16 + lr, kk * 4 + lq));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][tm], bfr[kk][tn], acc[tm][tn], 0, 0, 0);
}

// The same K-step on v_mfma_f32_32x32x16_bf16 (round 5: tools/probe_mfma_rate.hip - in this pattern the 16x16x32 shape sustains
// 1.47 PFLOP/s chip-wide, the 32x32x16 shape 2.43): QM x QN tiles of 32 x 32, four 16-deep sub-steps.  A fragment = 32 rows
// (lane % 32) x 8 k (chunk 2 ks + lane / 32 of the swizzled 128-byte row): the 16 lanes of a ds_read_b128 group hit 16 distinct
// 16-byte slots of the 64 banks (rows r, r + 1 share a chunk column 128 bytes apart, other row pairs another column).
typedef __attribute__((ext_vector_type(16))) float vfs_f32x16;
template <int QM, int QN>
__device__ __forceinline__ void mma_kstep32(const bf16_t* __restrict__ sA, const bf16_t* __restrict__ sB, int rowA0, int rowB0, int lane,
                                            vfs_f32x16 (&acc)[QM][QN]) {
  const int l32 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bf16x8 af[QM], bfr[QN];
#pragma unroll
    for (int i = 0; i < QM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sA + lds_off(rowA0 + i * 32 + l32, ks * 2 + h));
#pragma unroll
    for (int j = 0; j < QN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sB + lds_off(rowB0 + j * 32 + l32, ks * 2 + h));
#pragma unroll
    for (int i = 0; i < QM; ++i)
#pragma unroll
      for (int j = 0; j < QN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  }
}

// one 64-deep K-step of MFMAs for a wave: acc[tm][tn] += A(rowsA + tm*16) x B(rowsB + tn*16)
template <int TM, int TN, bool TSWZ>
__device__ __forceinline__ void mma_kstep(const bf16_t* __restrict__ sA, const bf16_t* __restrict__ sB,
                                          int rowA0, int rowB0, int lane, f32x4 (&acc)[TM][TN]) {
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8 af[TM], bfr[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      int row = rowA0 + tm * 16 + lr;
      int o = TSWZ ? lds_off_t(row, kk * 4 + lq) : lds_off(row, kk * 4 + lq);
      af[tm] = *reinterpret_cast<const bf16x8*>(sA + o);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      int row = rowB0 + tn * 16 + lr;
      int o = TSWZ ? lds_off_t(row, kk * 4 + lq) : lds_off(row, kk * 4 + lq);
      bfr[tn] = *reinterpret_cast<const bf16x8*>(sB + o);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
  }
}

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
// BK = channels per chunk: BK MFMAs per wave between two barriers (16: DAVIS R50 5.21-5.22 vs 5.15-5.18 ms per frame, R18 1.45 vs 1.41)
// PF = chunks of global loads in flight in registers (1: the next chunk is requested while this one is multiplied; 2: the one after it too)
template <int BK, int PF>
__global__ __launch_bounds__(256, 5 - BK / 32 - (PF - 1)) void conv_f32_kernel(ConvF32Args a) {
  constexpr int BM = 128, BN = 64, G = BK / 4, RPP = 256 / G;   // float4 groups per row and chunk, rows per loader pass
  // LDS planes [k][row], PAD dwords of padding per plane: MFMA step s reads plane 2s + (lane >> 5), 32 consecutive dwords per
  // half-wave - conflict-free; a half-wave of the loaders' ds_write_b32 covers 32 / G rows x G float4 groups at plane 4 * group + e:
  // bank (4 * group * (rows + PAD) + row) mod 32 = 8 * group + row (BK 16, PAD 2) or 4 * group + row (BK 32, PAD 1) - 32 different
  constexpr int PAD = BK == 16 ? 2 : 1;
  constexpr int PA = BM + PAD, PB = BN + PAD;
  __shared__ float sA[BK * PA];
  __shared__ float sB[BK * PB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long M = (long long)a.N * a.Ho * a.Wo;
  // XCD-aware tile order (1-D grid): workgroup L runs on XCD L % 8; on ONE XCD consecutive workgroups take the channel tiles of
  // ONE pixel tile, so the gathered pixels are fetched from HBM once and served to the other channel tiles by that XCD's L2
  // (pixel tile fastest, as the 2-D grid had it, spread the channel tiles of a pixel tile over the whole launch: 630 MB fetched
  // per launch for 124 MB of operands, round 3 counters)
  const int NT = (a.Cout + BN - 1) / BN;
  const long long MT = (M + BM - 1) / BM;
  const long long slot = blockIdx.x >> 3;
  const long long mt = (slot / NT) * 8 + (blockIdx.x & 7);
  if (mt >= MT) return;
  const long long m0 = mt * BM;
  const int n0 = (int)(slot % NT) * BN;
  const int C4 = a.Cin >> 2;
  const int K4 = a.KH * a.KW * C4;           // float4 groups along K
  // loaders: G CONSECUTIVE LANES read the 4 * BK contiguous bytes one row has in a chunk (thread = row t / G, float4 group t % G):
  // a wave instruction touches 64 / G rows.  With one row per lane (64 different cache lines per instruction) the vector memory
  // path needed ~156 clk per wave instruction (tools/probe_vmem_rate.hip) and bounded the kernel at half the MFMA rate.
  const int lq = t % G;
  // A: pixels t / G + RPP * i
  constexpr int NA = BM / RPP, NB = BN / RPP;
  const int ap = t / G;
  // Address arithmetic of the gather, kept OUT of the chunk loop: per row the element offset of its window origin
  // ((n H + iy0) W + ix0) Cin (may be negative: signed), per chunk ONE uniform delta for the tap and channel group; the tap / channel
  // counters advance incrementally (no division per chunk).  SQ counters (round 4): 5.2 vector instructions per MFMA, and the
  // fp32-input MFMA executes on the vector lanes - every one of them is matrix time lost (59 % of the fp32 MFMA peak).
  int iy0[NA], ix0[NA];
  long long abase[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const long long am = m0 + ap + RPP * i;
    iy0[i] = -(1 << 28); ix0[i] = -(1 << 28); abase[i] = 0;      // rows past M: every tap is out of range
    if (am < M) {
      const int hw = a.Ho * a.Wo;
      const int n = (int)(am / hw);
      const int rem = (int)(am - (long long)n * hw);
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      iy0[i] = oy * a.stride - a.pad; ix0[i] = ox * a.stride - a.pad;
      abase[i] = (((long long)n * a.H + iy0[i]) * a.W + ix0[i]) * a.Cin;
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

  f32x4 ra[PF][NA], rb[PF][NB];
  // this thread's float4 group of the NEXT chunk to load: k4 = chunk * G + lq -> (kh, kw, c4), advanced by G groups per chunk
  int l_k4 = lq, l_c4 = lq % C4, l_tap = lq / C4, l_kh = l_tap / a.KW, l_kw = l_tap - l_kh * a.KW;
  auto load = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {      // called for chunk 0, 1, 2, ... in order
    const bool k_ok = l_k4 < K4;
    const int dy = l_kh * a.dil, dx = l_kw * a.dil;
    const long long delta = ((long long)dy * a.W + dx) * a.Cin + l_c4 * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x4 v = zerof4();
      if (k_ok && (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W) v = ldf4(a.x + (abase[i] + delta));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = (b_ok[i] && k_ok) ? ldf4(wrow[i] + (size_t)l_k4 * 4) : zerof4();
    l_k4 += G; l_c4 += G;
    while (l_c4 >= C4) {      // (at most once unless Cin < 32)
      l_c4 -= C4;
      if (++l_kw == a.KW) { l_kw = 0; ++l_kh; }
    }
  };
  auto store = [&](const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
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
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nchunks) load(ra[u], rb[u]);
  for (int ch = 0; ch < nchunks; ch += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (ch + u < nchunks) {
        store(ra[u], rb[u]);
        __syncthreads();
        if (ch + u + PF < nchunks) load(ra[u], rb[u]);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
          const float b = sB[(2 * s + lk) * PB + wc0 + li];
          const float a0 = sA[(2 * s + lk) * PA + wp0 + li], a1 = sA[(2 * s + lk) * PA + wp0 + 32 + li];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
        }
        __syncthreads();
      }
    }
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


// ---------------------------------------------------------------------------------------------
// The same convolution, organised around what the fp32-input MFMA is on gfx950 (tools/probe_mfma_valu_mix.hip, round 4):
// v_mfma_f32_32x32x2_f32 runs on the vector lanes - vector instructions of ANY wave of the SIMD do not hide behind it, each one takes
// about 2.5 clk (v_mul_lo 4.6) out of the 64 clk an MFMA needs, at 4 waves per SIMD - and a wave that is gathering, storing to LDS or
// standing at a barrier contributes no MFMAs.  conv_f32_kernel above spent 2.8-5.2 vector instructions per MFMA and kept every
// wave out of the matrix pipe between its two barriers per chunk (59-61 % of the fp32 MFMA peak).  Here:
//   - ONE barrier per chunk: two LDS images; chunk ch + 1 is written to the other image while chunk ch is multiplied
//   - a straight-line loop body (no conditional loads: raw buffer loads, rows / taps / channels out of range read as zeros through the
//     descriptor's range check; no end-of-loop special cases: the chunks past the end load nothing and store zeros), so the LDS stores
//     and the gather of the next chunks are scheduled BETWEEN the MFMAs of this one
//   - 32-bit offsets and incremental (kh, kw, c) counters: ~1.3 vector instructions per MFMA
// Same products, same ascending-k fmaf chain per output as conv_f32_kernel: bit-identical results (tests/test_exact_f32.py).
// The descriptor is based at the first image the pixel tile touches (offsets are 32-bit): any batch size.
__device__ __forceinline__ unsigned fdiv(unsigned n, VfsFastDiv f) {
  const unsigned t = (unsigned)(((unsigned long long)n * f.m) >> 32);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
#define CONV_F32_OOB 0xFFFFFFF0u      // >= any num_records: the load returns zeros and moves nothing
// BN = output channels per workgroup: 64 (wave tile 64 x 32, 3 workgroups per CU) or 128 (wave tile 64 x 64: four MFMAs per four LDS
// reads, the gather of a pixel row shared by twice the MFMAs; 2 workgroups per CU) - every instruction that is not an MFMA costs
// matrix time here, so the wider tile is used wherever Cout fills it.
template <bool SMALLC, int BN>      // SMALLC: Cin < 32 (the stem, Cin = 4): several taps per chunk, the tap counters advance in a loop
__global__ __launch_bounds__(256, BN == 128 ? 2 : 3) void conv_f32_db_kernel(ConvF32Args a) {
  constexpr int BM = 128, BK = 32, G = BK / 4, RPP = 256 / G, NA = BM / RPP, NB = BN / RPP, NW = BN / 64;
  constexpr int PA = BM + 1, PB = BN + 1;      // [k][row] planes, one dword of padding (see conv_f32_kernel)
  __shared__ float sA[2][BK * PA];
  __shared__ float sB[2][BK * PB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // (all divisions by launch constants go through the launcher's multipliers: the four 64-bit divisions per thread this prologue
  //  started with were a quarter of the time of a K = 256 tile; M < 2^31 is checked by the launcher)
  const unsigned hw = (unsigned)(a.Ho * a.Wo);
  const unsigned M = (unsigned)a.N * hw;
  const unsigned NT = (unsigned)(a.Cout + BN - 1) / BN;
  const unsigned MT = (M + BM - 1) / BM;
  const unsigned slot = blockIdx.x >> 3;      // XCD-aware tile order, see conv_f32_kernel
  const unsigned slot_m = fdiv(slot, a.d_nt);
  const unsigned mt = slot_m * 8 + (blockIdx.x & 7);
  if (mt >= MT) return;
  const unsigned m0 = mt * BM;
  const int n0 = (int)(slot - slot_m * NT) * BN;
  const int C4 = a.Cin >> 2;
  const int K4 = a.KH * a.KW * C4;
  const int lq = t % G, ap = t / G;

  const long long n_first = fdiv(m0, a.d_hw);
  const long long img = (long long)a.H * a.W * a.Cin;      // elements of one input image
  const long long left = ((long long)a.N - n_first) * img * 4;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + n_first * img), 0,
                                                                       (unsigned)(left < 0xFFFFFF00LL ? left : 0xFFFFFF00LL), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((size_t)a.Cout * K4 * 16), 0x00020000);

  // A: pixels ap + RPP * i - window origin (may lie outside the image: the byte offset wraps, valid taps land in range again)
  int iy0[NA], ix0[NA];
  unsigned abase[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const unsigned am = m0 + ap + RPP * i;
    iy0[i] = -(1 << 28); ix0[i] = -(1 << 28); abase[i] = 0;      // rows past M: every tap is out of range
    if (am < M) {
      const long long n = fdiv(am, a.d_hw);
      const unsigned rem = am - (unsigned)n * hw;
      const int oy = (int)fdiv(rem, a.d_wo), ox = (int)rem - oy * a.Wo;
      iy0[i] = oy * a.stride - a.pad; ix0[i] = ox * a.stride - a.pad;
      abase[i] = (unsigned)(((((n - n_first) * a.H + iy0[i]) * a.W + ix0[i]) * a.Cin) * 4);
    }
  }
  // B: output channels ap + RPP * i
  bool b_ok[NB];
  unsigned wb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int c = n0 + ap + RPP * i;
    b_ok[i] = c < a.Cout;
    wb[i] = (unsigned)((size_t)(b_ok[i] ? c : 0) * K4 * 16);
  }

  // this thread's float4 group of the NEXT chunk to load, k4 = chunk * G + lq -> (kh, kw, c4); dy, dx, delta follow it incrementally
  int l_k4 = lq, l_c4, l_kw, l_kh;
  { const int tap = (int)fdiv((unsigned)lq, a.d_c4); l_c4 = lq - tap * C4; l_kh = (int)fdiv((unsigned)tap, a.d_kw); l_kw = tap - l_kh * a.KW; }
  int dy = l_kh * a.dil, dx = l_kw * a.dil;
  unsigned delta = (unsigned)((((long long)dy * a.W + dx) * a.Cin + l_c4 * 4) * 4);
  const unsigned d_tap = (unsigned)((a.dil - 1) * a.Cin * 4);      // channel wrap onto the next tap of the row: + dil * Cin - Cin elements
  const unsigned d_row = (unsigned)((((long long)a.dil * a.W - (long long)(a.KW - 1) * a.dil) * a.Cin - a.Cin) * 4);   // onto the next row
  f32x4 ra[NA], rb[NB];
  auto load = [&]() {      // chunk 0, 1, 2, ... in order; past the end: nothing is moved
    const bool k_ok = l_k4 < K4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = k_ok && (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? abase[i] + delta : CONV_F32_OOB, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (b_ok[i] && k_ok) ? wb[i] + (unsigned)l_k4 * 16u : CONV_F32_OOB, 0, 0));
    l_k4 += G; l_c4 += G; delta += G * 16;
    if (SMALLC) {
      while (l_c4 >= C4) {
        l_c4 -= C4;
        if (++l_kw == a.KW) { l_kw = 0; ++l_kh; delta += d_row; dx = 0; dy += a.dil; }
        else { delta += d_tap; dx += a.dil; }
      }
    } else {      // at most one wrap per chunk (C4 >= G): selects, no branch
      const bool wrap = l_c4 >= C4;
      const bool wrap2 = wrap && l_kw + 1 == a.KW;
      l_c4 -= wrap ? C4 : 0;
      l_kw = wrap2 ? 0 : l_kw + (wrap ? 1 : 0);
      delta += wrap ? (wrap2 ? d_row : d_tap) : 0u;
      dx = wrap2 ? 0 : dx + (wrap ? a.dil : 0);
      dy += wrap2 ? a.dil : 0;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < NA; ++i) sA[buf][(4 * lq + e) * PA + ap + RPP * i] = ra[i][e];
#pragma unroll
      for (int i = 0; i < NB; ++i) sB[buf][(4 * lq + e) * PB + ap + RPP * i] = rb[i][e];
    }
  };

  f32x16 acc[2][NW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wp0 = (wave & 1) * 64, wc0 = (wave >> 1) * (BN / 2);
  const int li = lane & 31, lk = lane >> 5;
  const int nchunks = (K4 + G - 1) / G;
  load();
  store(0);
  load();
  __syncthreads();
  for (int ch = 0; ch < nchunks; ch += 2) {      // two chunks per trip: the LDS image of every access is a constant
#pragma unroll
    for (int cur = 0; cur < 2; ++cur) {
      if (cur == 1 && ch + 1 >= nchunks) break;
      if (!(a.dbg & 2)) store(cur ^ 1);      // chunk ch + cur + 1 (zeros past the end), requested one chunk ago
      if (!(a.dbg & 1)) load();              // chunk ch + cur + 2
      __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler sinks the requests below the MFMAs: nothing would hide their latency)
      if (!(a.dbg & 4))
#pragma unroll
      for (int s = 0; s < BK / 2; ++s) {
        const float a0 = sA[cur][(2 * s + lk) * PA + wp0 + li], a1 = sA[cur][(2 * s + lk) * PA + wp0 + 32 + li];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const float b = sB[cur][(2 * s + lk) * PB + wc0 + 32 * j + li];
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][j], 0, 0, 0);
        }
      }
      __syncthreads();      // image cur read by all, image cur ^ 1 written by all
    }
  }
  // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31].  Outputs and identity values go through descriptors based at the
  // tile's first row whose range check drops the rows past M and the channels past Cout: one 32-bit add per element, no compare
  // (the per-element 64-bit address arithmetic and row tests were 325 vector instructions per wave and tile)
  const unsigned rows = M - m0 < (unsigned)BM ? M - m0 : (unsigned)BM;
  const unsigned row_b = (unsigned)a.Cout * 4u;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)m0 * a.Cout), 0, rows * row_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)((a.res ? a.res : a.y) + (size_t)m0 * a.Cout), 0, rows * row_b, 0x00020000);
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int co = n0 + wc0 + 32 * j + li;
    const unsigned off0 = co < a.Cout ? (unsigned)(wp0 + 4 * lk) * row_b + (unsigned)co * 4u : 0x80000000u;
    const float sc = (a.scale && co < a.Cout) ? a.scale[co] : 1.f, sh = (a.scale && co < a.Cout) ? a.shift[co] : 0.f;
    float rv[2][16];
    if (a.res && !(a.dbg & 32)) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[pt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, off0 + (unsigned)(pt * 32 + (r & 3) + 8 * (r >> 2)) * row_b, 0, 0));
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[pt][j][r];
        if (a.scale) v = __builtin_fmaf(v, sc, sh);
        if (a.res && !(a.dbg & 32)) v = v + rv[pt][r];
        if (a.relu) v = v > 0.f ? v : 0.f;
        if (!(a.dbg & 16) || v == 12345.678f)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, off0 + (unsigned)(pt * 32 + (r & 3) + 8 * (r >> 2)) * row_b, 0, 0);
      }
  }
}

int vfs_option_conv_f32_variant = 0;      // A/B knob: 321 = conv_f32_kernel (two barriers per chunk)
int vfs_option_conv_f32_dbg = 0;
int vfs_conv_f32_launch(const ConvF32Args& a_in, hipStream_t s) {
  ConvF32Args a = a_in;
  a.dbg = vfs_option_conv_f32_dbg;
  if (a.Cin % 4) return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: Cin % 4 (pad the 3-channel input to NHWC4)");
  if (a.N < 1 || a.Ho < 1 || a.Wo < 1 || a.Cout < 1 || a.stride < 1 || a.dil < 1) return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: geometry");
  if (a.Ho != (a.H + 2 * a.pad - a.dil * (a.KH - 1) - 1) / a.stride + 1 || a.Wo != (a.W + 2 * a.pad - a.dil * (a.KW - 1) - 1) / a.stride + 1)
    return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: output size does not match (H + 2 pad - dil (K - 1) - 1) / stride + 1");
  if (a.scale && !a.shift) return vfs_set_error(VFS_ERR_ARG, "conv_f32: scale without shift");
  if (a.Cin < 4 || a.KH < 1 || a.KW < 1 || a.H < 1 || a.W < 1) return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: geometry");
  const long long M = (long long)a.N * a.Ho * a.Wo;
  a.d_hw = vfs_fastdiv((unsigned)(a.Ho * a.Wo)); a.d_wo = vfs_fastdiv((unsigned)a.Wo);
  a.d_c4 = vfs_fastdiv((unsigned)(a.Cin / 4)); a.d_kw = vfs_fastdiv((unsigned)a.KW);
  // channel tile 128: where Cout fills it, K is long enough to amortise the twice larger epilogue and the tiles still fill two
  // workgroups per CU (per-layer table of the DAVIS ResNet-50 pass, MEASUREMENTS.md round 4: res4 conv1 / conv2 -5 %, every other
  // layer +2..13 %).  A/B knob conv_f32_variant: 64 / 128 force one, 321 = the two-barrier kernel
  const bool wide = a.Cout % 128 == 0 && (long long)a.KH * a.KW * a.Cin >= 512 && ((M + 127) / 128) * (a.Cout / 128) >= 384;
  const int bn = vfs_option_conv_f32_variant == 64 || vfs_option_conv_f32_variant == 321 ? 64 : vfs_option_conv_f32_variant == 128 ? 128 : wide ? 128 : 64;
  const long long mt8 = ((M + 127) / 128 + 7) / 8 * 8, nt = (a.Cout + bn - 1) / bn;      // pixel tiles padded to the 8 XCDs
  a.d_nt = vfs_fastdiv((unsigned)nt);
  if (mt8 * nt > 0x7fffffffLL) return vfs_set_error(VFS_ERR_SHAPE, "conv_f32: too many tiles");
  const dim3 grid((unsigned)(mt8 * nt));
  // images one pixel tile can touch: 32-bit offsets from the first of them
  const long long span = (127 / ((long long)a.Ho * a.Wo) + 2) * a.H * a.W * a.Cin * 4;
  if (vfs_option_conv_f32_variant == 321 || M >= 0x7fffff00LL || (long long)a.Cout * 512 >= 0x7fffffffLL || span >= 0xFFFFFF00LL || (long long)a.Cout * a.KH * a.KW * a.Cin * 4 >= 0xFFFFFF00LL)
    hipLaunchKernelGGL((conv_f32_kernel<32, 1>), grid, dim3(256), 0, s, a);
  else if (a.Cin < 32 && bn == 64) hipLaunchKernelGGL((conv_f32_db_kernel<true, 64>), grid, dim3(256), 0, s, a);
  else if (a.Cin < 32) hipLaunchKernelGGL((conv_f32_db_kernel<true, 128>), grid, dim3(256), 0, s, a);
  else if (bn == 64) hipLaunchKernelGGL((conv_f32_db_kernel<false, 64>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_f32_db_kernel<false, 128>), grid, dim3(256), 0, s, a);
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

// grid (LP_POST_BLOCKS, CO): one class per workgroup (round 6: the 64 workgroups of the first version walked the classes one after
// the other - a quarter of the CUs, 37 us per 480 x 854 frame with four classes).  min / max do not depend on the order: same bits.
__global__ __launch_bounds__(256) void seg_minmax_exact_kernel(const float* __restrict__ seg, float* __restrict__ partial, int H, int W,
                                                               int CO, int Ho, int Wo) {
  __shared__ float smn[256], smx[256];
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  const int total = Ho * Wo, c = blockIdx.y;
  const int stride = gridDim.x * 256, dy = stride / Wo, dx = stride - dy * Wo;
  float mn = INFINITY, mx = -INFINITY;
  int p = blockIdx.x * 256 + threadIdx.x, oy = p / Wo, ox = p - oy * Wo;
  for (; p < total; p += stride) {
    const float v = bilerp_exact(seg, H, W, CO, c, oy, ox, sy, sx);
    mn = v < mn ? v : mn; mx = v > mx ? v : mx;
    oy += dy; ox += dx;
    if (ox >= Wo) { ox -= Wo; ++oy; }
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
  hipLaunchKernelGGL(seg_minmax_exact_kernel, dim3(nblk, CO), dim3(256), 0, s, seg, partial, H, W, CO, Ho, Wo);
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

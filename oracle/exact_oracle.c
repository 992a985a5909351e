/* TEST INFRASTRUCTURE -- CPU restatement, in plain C, of the reference's fp32 EVALUATION path
 * (VanillaTracker.forward_test: ResNet in eval mode -> L2-normalised feature bank ->
 * masked_attention_efficient -> upsample / min-max / argmax), written operation by operation so that
 * its results are defined down to the last bit:
 *
 *   - every dot product is ONE ascending chain  acc = fmaf(a_k, b_k, acc)  starting at +0
 *     (exactly what v_mfma_f32_32x32x2_f32 computes, MI355X guide "FP32-input MFMA": bitwise a
 *     k-ordered fmaf chain), everything else is a single correctly rounded fp32 operation
 *     (compiled with -ffp-contract=off: the compiler may not fuse or reassociate);
 *   - exp() of the softmax is the explicit polynomial below (the product kernels carry the same
 *     sequence), ties of the top-k are broken by the lowest candidate index.
 *
 * The product path (vfs_amd/csrc/exact_f32.hip) must reproduce these results BIT FOR BIT on the GPU;
 * this file in turn is pinned against vectors captured from the reference itself
 * (tests/golden/forward_test_r18*.npz, resnet*_dilated_eval.npz, masked_attention.npz) to fp32
 * summation-order accuracy (tests/test_exact_oracle.py).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline may load it.
 *
 * Reference lines restated: mmaction/models/backbones/resnet.py:15-232,555-575 (blocks, forward),
 * mmcv ConvModule (conv -> BN(eval) -> ReLU), mmaction/models/common/local_attention.py:237-348,
 * common/affinity_utils.py:144-156, trackers/vanilla_tracker.py:150-181.
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (oracle/exact_oracle.py). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XO_TOPK_MAX 10

/* ---- conv2d, NHWC fp32, weights [KH][KW][Cin][Cout] (the chain runs kh, kw, cin ascending) ------
 * y = [relu]( fmaf(acc, scale[co], shift[co]) [+ res] );  scale == NULL: y = acc [+ res]
 * out-of-range taps contribute fmaf(0, w, acc) as in the kernel (zero-filled operand tile) */
#define XO_PB 8
void xo_conv2d(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y, int N,
               int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int dil, int relu) {
  const long long rows = (long long)N * Ho;
  const int nxb = (Wo + XO_PB - 1) / XO_PB;
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * XO_PB * (size_t)Cout);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (long long row = 0; row < rows; ++row) {
      for (int xb = 0; xb < nxb; ++xb) {
        const int n = (int)(row / Ho), oy = (int)(row % Ho);
        const int ox0 = xb * XO_PB;
        const int np = Wo - ox0 < XO_PB ? Wo - ox0 : XO_PB;
        memset(acc, 0, sizeof(float) * XO_PB * (size_t)Cout);
        for (int kh = 0; kh < KH; ++kh) {
          const int iy = oy * stride - pad + kh * dil;
          for (int kw = 0; kw < KW; ++kw) {
            const float* xp[XO_PB];
            for (int p = 0; p < XO_PB; ++p) {
              const int ix = (ox0 + p) * stride - pad + kw * dil;
              xp[p] = (p < np && iy >= 0 && iy < H && ix >= 0 && ix < W) ? x + (((size_t)n * H + iy) * W + ix) * Cin : NULL;
            }
            const float* wp = w + ((size_t)(kh * KW + kw) * Cin) * Cout;
            for (int c = 0; c < Cin; ++c) {
              const float* wc = wp + (size_t)c * Cout;
              for (int p = 0; p < XO_PB; ++p) {
                const float xv = xp[p] ? xp[p][c] : 0.0f;
                float* a = acc + (size_t)p * Cout;
                for (int co = 0; co < Cout; ++co) a[co] = __builtin_fmaf(xv, wc[co], a[co]);
              }
            }
          }
        }
        for (int p = 0; p < np; ++p) {
          const size_t o = (((size_t)n * Ho + oy) * Wo + ox0 + p) * Cout;
          for (int co = 0; co < Cout; ++co) {
            float v = acc[(size_t)p * Cout + co];
            if (scale) v = __builtin_fmaf(v, scale[co], shift[co]);
            if (res) v = v + res[o + co];
            if (relu) v = v > 0.0f ? v : 0.0f;
            y[o + co] = v;
          }
        }
      }
    }
    free(acc);
  }
}

/* nn.MaxPool2d(3, 2, 1) (resnet.py:435), NHWC */
void xo_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo) {
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox)
        for (int c = 0; c < C; ++c) {
          float m = -INFINITY;
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              const float v = x[(((size_t)n * H + iy) * W + ix) * C + c];
              m = v > m ? v : m;
            }
          y[(((size_t)n * Ho + oy) * Wo + ox) * C + c] = m;
        }
}

/* F.normalize(p=2, dim=channel, eps=1e-12) of rows [P][C] (local_attention.py:277-279).  Summation
 * order of the kernel: lane l of 64 owns the float4 groups l, l+64, ... (one fmaf chain), then a
 * butterfly over the lanes (xor 32, 16, 8, 4, 2, 1); y = x / max(sqrt(ss), eps). */
void xo_l2norm_rows(const float* x, float* y, long long P, int C) {
#pragma omp parallel for
  for (long long r = 0; r < P; ++r) {
    const float* s = x + (size_t)r * C;
    float part[64];
    for (int l = 0; l < 64; ++l) {
      float a = 0.0f;
      for (int g = l; g * 4 < C; g += 64)
        for (int e = 0; e < 4; ++e) a = __builtin_fmaf(s[g * 4 + e], s[g * 4 + e], a);
      part[l] = a;
    }
    for (int d = 32; d >= 1; d >>= 1) {
      float nxt[64];
      for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ d];
      memcpy(part, nxt, sizeof(part));
    }
    float nrm = sqrtf(part[0]);
    nrm = nrm > 1e-12f ? nrm : 1e-12f;
    for (int c = 0; c < C; ++c) y[(size_t)r * C + c] = s[c] / nrm;
  }
}

/* exp(x) for x <= 0 (softmax after subtracting the maximum): n = rint(x*log2(e)); r = x - n*ln2 in two
 * fmaf steps; degree-6 Horner polynomial in fmaf; scale by 2^n.  The kernels carry the same sequence. */
float xo_exp(float x) {
  if (x < -87.0f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  p = __builtin_fmaf(p, r * r, r);
  p = p + 1.0f;
  return ldexpf(p, (int)n);
}

/* total order of the top-k: larger score first, equal scores: smaller candidate id first */
static inline int xo_better(float s, int id, float ts, int tid) { return s > ts || (s == ts && id < tid); }

/* masked_attention_efficient with the circular spatial_neighbor mask for ONE query frame.
 * fbank [frames][HW][C] L2-normalised, sbank [frames][HW][CO], kslot[nkeys] bank indices of the key frames
 * in the reference's order (first frame first, duplicates allowed), out [HW][CO].
 * score = chain_c fmaf(key[c], query[c]) / temperature; candidate id = f*HW + pixel; radius <= 0: no mask */
void xo_labelprop(const float* fbank, const float* sbank, float* out, int qframe, const int* kslot, int nkeys, int H, int W,
                  int C, int CO, int radius_all, int non_mask_len, int topk, float temperature) {
  const int HW = H * W;
#pragma omp parallel for schedule(dynamic, 8)
  for (int q = 0; q < HW; ++q) {
    const int qy = q / W, qx = q % W;
    const float* qv = fbank + ((size_t)qframe * HW + q) * C;
    float tv[XO_TOPK_MAX];
    int ti[XO_TOPK_MAX];
    for (int i = 0; i < XO_TOPK_MAX; ++i) { tv[i] = -INFINITY; ti[i] = 0x7fffffff; }
    for (int f = 0; f < nkeys; ++f) {
      /* the first non_mask_len key frames are not masked (local_attention.py:303-309) */
      const int radius = f < non_mask_len ? 0 : radius_all;
      int y0 = 0, y1 = H - 1, x0 = 0, x1 = W - 1;
      if (radius > 0) {
        y0 = qy - (radius - 1) > 0 ? qy - (radius - 1) : 0; y1 = qy + radius - 1 < H - 1 ? qy + radius - 1 : H - 1;
        x0 = qx - (radius - 1) > 0 ? qx - (radius - 1) : 0; x1 = qx + radius - 1 < W - 1 ? qx + radius - 1 : W - 1;
      }
      const float* kb = fbank + (size_t)kslot[f] * HW * C;
      for (int ky = y0; ky <= y1; ++ky)
        for (int kx = x0; kx <= x1; ++kx) {
          const int dy = ky - qy, dx = kx - qx;
          if (radius > 0 && dy * dy + dx * dx >= radius * radius) continue;
          const float* kv = kb + (size_t)(ky * W + kx) * C;
          float acc = 0.0f;
          for (int c = 0; c < C; ++c) acc = __builtin_fmaf(kv[c], qv[c], acc);
          const float s = acc / temperature;
          const int id = f * HW + ky * W + kx;
          if (!xo_better(s, id, tv[topk - 1], ti[topk - 1])) continue;
          int j = topk - 1;
          while (j > 0 && xo_better(s, id, tv[j - 1], ti[j - 1])) { tv[j] = tv[j - 1]; ti[j] = ti[j - 1]; --j; }
          tv[j] = s; ti[j] = id;
        }
    }
    float e[XO_TOPK_MAX], z = 0.0f;
    for (int j = 0; j < topk; ++j) {
      e[j] = (ti[j] != 0x7fffffff && tv[j] > -INFINITY) ? xo_exp(tv[j] - tv[0]) : 0.0f;
      z = z + e[j];
    }
    for (int c = 0; c < CO; ++c) {
      float acc = 0.0f;
      for (int j = 0; j < topk; ++j) {
        if (!(e[j] > 0.0f)) continue;
        const int f = ti[j] / HW, px = ti[j] - f * HW;
        acc = __builtin_fmaf(e[j] / z, sbank[((size_t)kslot[f] * HW + px) * CO + c], acc);
      }
      out[(size_t)q * CO + c] = acc;
    }
  }
}

/* vanilla_tracker.py:162-181: F.interpolate(bilinear, align_corners=False) -> per-channel min-max
 * normalisation where max > 0 -> argmax (first maximum) -> uint8.  seg [H][W][CO] */
static inline float xo_bilerp(const float* seg, int H, int W, int CO, int c, int oy, int ox, float sy, float sx) {
  float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.0f ? 0.0f : fy; fx = fx < 0.0f ? 0.0f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float v00 = seg[((size_t)y0 * W + x0) * CO + c], v01 = seg[((size_t)y0 * W + x1) * CO + c];
  const float v10 = seg[((size_t)y1 * W + x0) * CO + c], v11 = seg[((size_t)y1 * W + x1) * CO + c];
  const float top = hx * v00 + lx * v01, bot = hx * v10 + lx * v11;
  return hy * top + ly * bot;
}

void xo_seg_postprocess(const float* seg, uint8_t* label, int H, int W, int CO, int Ho, int Wo) {
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  float* mn = (float*)malloc(sizeof(float) * CO);
  float* mx = (float*)malloc(sizeof(float) * CO);
  for (int c = 0; c < CO; ++c) {
    float a = INFINITY, b = -INFINITY;
#pragma omp parallel for reduction(min : a) reduction(max : b)
    for (int p = 0; p < Ho * Wo; ++p) {
      const float v = xo_bilerp(seg, H, W, CO, c, p / Wo, p % Wo, sy, sx);
      a = v < a ? v : a; b = v > b ? v : b;
    }
    mn[c] = a; mx[c] = b;
  }
#pragma omp parallel for
  for (int p = 0; p < Ho * Wo; ++p) {
    float best = -INFINITY;
    int bc = 0;
    for (int c = 0; c < CO; ++c) {
      float v = xo_bilerp(seg, H, W, CO, c, p / Wo, p % Wo, sy, sx);
      if (mx[c] > 0.0f) v = (v - mn[c]) / (mx[c] - mn[c] + 1e-12f);
      if (v > best) { best = v; bc = c; }
    }
    label[p] = (uint8_t)bc;
  }
  free(mn);
  free(mx);
}

/* F.interpolate(bilinear, align_corners=False) between layouts: element (c, y, x) at c*sc + y*sy + x*sx */
void xo_bilinear_resize(const float* src, float* dst, int C, int H, int W, int Ho, int Wo, long long ssc, long long ssy,
                        long long ssx, long long dsc, long long dsy, long long dsx) {
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
#pragma omp parallel for
  for (int oy = 0; oy < Ho; ++oy)
    for (int ox = 0; ox < Wo; ++ox) {
      float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
      fy = fy < 0.0f ? 0.0f : fy; fx = fx < 0.0f ? 0.0f : fx;
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float hy = 1.0f - ly, hx = 1.0f - lx;
      for (int c = 0; c < C; ++c) {
        const float* s = src + (size_t)c * ssc;
        const float v00 = s[y0 * ssy + x0 * ssx], v01 = s[y0 * ssy + x1 * ssx];
        const float v10 = s[y1 * ssy + x0 * ssx], v11 = s[y1 * ssy + x1 * ssx];
        const float top = hx * v00 + lx * v01, bot = hx * v10 + lx * v11;
        dst[(size_t)c * dsc + (size_t)oy * dsy + (size_t)ox * dsx] = hy * top + ly * bot;
      }
    }
}

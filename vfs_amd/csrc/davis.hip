// DAVIS-2017 semi-supervised J&F evaluation: the integer ingredients of region similarity J and
// boundary measure F for every object and evaluated frame of a sequence, on label maps that are
// already in HBM (the propagation kernels produce them there).
//
// Reference call site: DavisDataset.davis_evaluate (datasets/davis_dataset.py:68-140) hands PNG files
// to the un-vendored davis2017.evaluation package; the algorithm restated here is that package's
// db_eval_iou / db_eval_boundary / _seg2bmap / f_measure (see oracle/davis_jf.py for the restatement
// this file is tested against, bit-exact on the counts).
//
// Objects are ONE-HOT bit words: pixel label p (1..K, not void) = bit p-1.  A boundary map of all K
// objects is then three XORs of neighbouring words (east, south, south-east), and the dilated match
// of f_measure is an OR of the other side's boundary words over the disk - one pass serves all objects.
// Byte / bit arithmetic, HBM- and latency-bound: coalesced rows, no MFMA.
#include "vfs_common.h"
#include "vfs_ops.h"

__device__ __forceinline__ unsigned davis_onehot(unsigned label, bool is_void, int nobj) {
  return (label >= 1u && label <= (unsigned)nobj && !is_void) ? (1u << (label - 1u)) : 0u;
}

// pass 1: boundary words of prediction and ground truth + intersection / union counts.
// grid (pixel blocks, evaluated frames); counts[f][k][0..1] += ...
__global__ __launch_bounds__(256) void davis_bmap_kernel(DavisArgs a) {
  __shared__ int hist[32][2];
  const int t = threadIdx.x, f = blockIdx.y;           // evaluated frame f = sequence frame f + 1
  for (int i = t; i < 64; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  const size_t fo = (size_t)(f + 1) * a.H * a.W;
  const uint8_t* P = a.pred + fo;
  const uint8_t* G = a.gt + fo;
  for (int i = blockIdx.x * 256 + t; i < a.H * a.W; i += gridDim.x * 256) {
    const int y = i / a.W, x = i - y * a.W;
    const bool xe = x + 1 < a.W, ys = y + 1 < a.H;
    auto words = [&](int yy, int xx, unsigned& mp, unsigned& mg) {
      const unsigned g = G[yy * a.W + xx];
      const bool v = a.use_void && g == 255u;
      mp = davis_onehot(P[yy * a.W + xx], v, a.nobj);
      mg = davis_onehot(g, v, a.nobj);
    };
    unsigned mp, mg, ep = 0, eg = 0, sp = 0, sg = 0, dp = 0, dg = 0;
    words(y, x, mp, mg);
    if (xe) words(y, x + 1, ep, eg);
    if (ys) words(y + 1, x, sp, sg);
    if (xe && ys) words(y + 1, x + 1, dp, dg);
    unsigned bp, bg;
    if (!ys && !xe) { bp = 0; bg = 0; }                       // bottom-right corner
    else if (!ys) { bp = mp ^ ep; bg = mg ^ eg; }             // last row: along the edge only
    else if (!xe) { bp = mp ^ sp; bg = mg ^ sg; }             // last column
    else { bp = (mp ^ ep) | (mp ^ sp) | (mp ^ dp); bg = (mg ^ eg) | (mg ^ sg) | (mg ^ dg); }
    a.bp[(size_t)f * a.H * a.W + i] = bp;
    a.bg[(size_t)f * a.H * a.W + i] = bg;
    unsigned un = mp | mg, in = mp & mg;
    while (un) {                                              // at most two bits
      const int k = __builtin_ctz(un);
      un &= un - 1;
      atomicAdd(&hist[k][1], 1);
      if ((in >> k) & 1u) atomicAdd(&hist[k][0], 1);
    }
  }
  __syncthreads();
  if (t < 2 * a.nobj) {
    const int k = t >> 1, s = t & 1;
    if (hist[k][s]) atomicAdd(&a.counts[((size_t)f * a.nobj + k) * 6 + s], hist[k][s]);
  }
}

// pass 2: boundary pixel counts and matches within the disk (integer atomics: order-independent)
__global__ __launch_bounds__(256) void davis_match_kernel(DavisArgs a) {
  __shared__ int hist[32][4];
  const int t = threadIdx.x, f = blockIdx.y;
  for (int i = t; i < 128; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  const unsigned* BP = a.bp + (size_t)f * a.H * a.W;
  const unsigned* BG = a.bg + (size_t)f * a.H * a.W;
  const int r = a.radius, r2 = r * r;
  for (int i = blockIdx.x * 256 + t; i < a.H * a.W; i += gridDim.x * 256) {
    const unsigned bp = BP[i], bg = BG[i];
    if ((bp | bg) == 0u) continue;
    const int y = i / a.W, x = i - y * a.W;
    unsigned pd = 0, gd = 0;                                   // dilated words at this pixel
    const int y0 = max(0, y - r), y1 = min(a.H - 1, y + r);
    for (int yy = y0; yy <= y1 && ((bp & ~gd) | (bg & ~pd)); ++yy) {
      const int dy = yy - y;
      int dx = 0;
      while ((dx + 1) * (dx + 1) + dy * dy <= r2) ++dx;        // half width of the disk row
      const int x0 = max(0, x - dx), x1 = min(a.W - 1, x + dx);
      for (int xx = x0; xx <= x1; ++xx) {
        pd |= BP[yy * a.W + xx];
        gd |= BG[yy * a.W + xx];
      }
    }
    unsigned all = bp | bg;
    while (all) {
      const int k = __builtin_ctz(all);
      all &= all - 1;
      const unsigned m = 1u << k;
      if (bp & m) { atomicAdd(&hist[k][0], 1); if (gd & m) atomicAdd(&hist[k][2], 1); }
      if (bg & m) { atomicAdd(&hist[k][1], 1); if (pd & m) atomicAdd(&hist[k][3], 1); }
    }
  }
  __syncthreads();
  if (t < 4 * a.nobj) {
    const int k = t >> 2, s = t & 3;
    if (hist[k][s]) atomicAdd(&a.counts[((size_t)f * a.nobj + k) * 6 + 2 + s], hist[k][s]);
  }
}

int vfs_davis_counts_launch(const DavisArgs& a, hipStream_t s) {
  if (a.nobj < 0 || a.nobj > 32) return vfs_set_error(VFS_ERR_SHAPE, "davis_counts: 0..32 objects");
  if (a.radius < 0 || a.radius > 64) return vfs_set_error(VFS_ERR_SHAPE, "davis_counts: radius 0..64");
  const int F = a.T - 2;
  if (F <= 0 || a.nobj == 0) return VFS_OK;
  if (hipMemsetAsync(a.counts, 0, (size_t)F * a.nobj * 6 * sizeof(int), s) != hipSuccess)
    return vfs_set_error(VFS_ERR_LAUNCH, "davis_counts: memset");
  int bx = (a.H * a.W + 255) / 256;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(davis_bmap_kernel, dim3(bx, F), dim3(256), 0, s, a);
  hipLaunchKernelGGL(davis_match_kernel, dim3(bx, F), dim3(256), 0, s, a);
  return vfs_check_launch("davis_counts");
}

// Training input pipeline on the GPU: RandomResizedCrop -> Resize(224, bilinear) -> Flip
// -> Normalize -> FormatShape('NCTHW') of the reference's train_pipeline (configs/r*_*.py:48-91;
// pipelines/augmentations.py:171-334 crop, :487-596 resize, :600-707 flip, :711-794 normalize;
// pipelines/formating.py:222-309) fused into ONE pass over the decoded uint8 frames: every output pixel
// reads its (at most) four source pixels inside the frame's crop box and writes the normalised value
// straight into the fp32 [B][V][3][T][H][W] tensor train_step takes and / or the bf16 NHWC4 frame buffer
// the stem kernel reads.  The random decisions (crop boxes, flips) are drawn on the host with the
// reference's rules and arrive as small per-frame arrays.
//
// The image arithmetic of the reference lives in mmcv -> OpenCV (absent here): restated from the
// published implementation - cv2.resize INTER_LINEAR on 8-bit data is FIXED POINT (11-bit coefficients,
// two passes, the vertical pass on (value >> 4) with a 2-bit rounding), cv2.flip, and mmcv.imnormalize_
// = cv2.subtract / cv2.multiply of a float32 image with float64 scalars (computed in double, stored as
// float32 after each step).  Byte work, HBM-bound; parity with oracle/pipeline_oracle.py is bit-exact.
#include "vfs_common.h"
#include "vfs_ops.h"

// cv2 resize (resize.cpp, INTER_LINEAR, 8-bit): source index and 11-bit weights of destination index d for an
// axis of n samples scaled to m.  Columns (XAXIS): an index outside [0, n-1) is clamped AND its fraction
// zeroed; rows: the two row indices are clamped when they are fetched, the weights stay.
template <bool XAXIS>
__device__ __forceinline__ void cv_linear_coef(int d, int n, int m, int& s0, int& s1, int& w0, int& w1) {
  const double scale = 1.0 / ((double)m / (double)n);     // cv::resize: scale = 1 / inv_scale
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (XAXIS) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
  }
  s0 = min(max(s, 0), n - 1);
  s1 = min(max(s + 1, 0), n - 1);
  w0 = (int)rintf((1.f - f) * 2048.f);     // saturate_cast<short>: round half to even
  w1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void crop_resize_flip_norm_kernel(PipelineArgs a) {
  const long long total = (long long)a.B * a.V * a.T * a.Ho * a.Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % a.Wo);
    long long r = i / a.Wo;
    const int y = (int)(r % a.Ho); r /= a.Ho;
    const int f = (int)r;                          // frame index in pipeline order: (b, v, t)
    const int t = f % a.T, v = (f / a.T) % a.V, b = f / (a.T * a.V);
    // boxes come from the host: clamp them to the frame so a bad box can never read outside it
    const int left = min(max(a.boxes[4 * f], 0), a.Ws - 1), top = min(max(a.boxes[4 * f + 1], 0), a.Hs - 1);
    const int right = min(max(a.boxes[4 * f + 2], left + 1), a.Ws), bottom = min(max(a.boxes[4 * f + 3], top + 1), a.Hs);
    const int cw = right - left, ch = bottom - top;
    const int xr = a.flips[f] ? a.Wo - 1 - x : x;  // cv2.flip(img, 1) AFTER the resize
    int sx, sx1, ax0, ax1, sy, sy1, by0, by1;
    cv_linear_coef<true>(xr, cw, a.Wo, sx, sx1, ax0, ax1);
    cv_linear_coef<false>(y, ch, a.Ho, sy, sy1, by0, by1);
    const uint8_t* img = a.src + (size_t)f * a.Hs * a.Ws * 3;
    const uint8_t* p00 = img + ((size_t)(top + sy) * a.Ws + left + sx) * 3;
    const uint8_t* p01 = img + ((size_t)(top + sy) * a.Ws + left + sx1) * 3;
    const uint8_t* p10 = img + ((size_t)(top + sy1) * a.Ws + left + sx) * 3;
    const uint8_t* p11 = img + ((size_t)(top + sy1) * a.Ws + left + sx1) * 3;
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = p00[c] * ax0 + p01[c] * ax1;          // horizontal pass, 8 + 11 bits
      const int h1 = p10[c] * ax0 + p11[c] * ax1;
      const int u = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;   // vertical pass
      const float d = (float)((double)u - a.mean[c]);       // cv2.subtract(float32 image, float64 scalar)
      o[c] = (float)((double)d * a.stdinv[c]);              // cv2.multiply(..., 1/std)
    }
    if (a.imgs) {
      const size_t plane = (size_t)a.T * a.Ho * a.Wo;
      float* q = a.imgs + (((size_t)b * a.V + v) * 3) * plane + ((size_t)t * a.Ho + y) * a.Wo + x;
      q[0] = o[0]; q[plane] = o[1]; q[2 * plane] = o[2];
    }
    if (a.x4) {   // frame order of the backbone batch: (view, b, t), as vfs_imgs_to_nhwc4
      const size_t fr = ((size_t)v * a.B + b) * a.T + t;
      u32x2 pk;
      pk.x = pack2bf(o[0], o[1]);
      pk.y = pack2bf(o[2], 0.f);
      st8(a.x4 + ((fr * a.Ho + y) * a.Wp + x) * 4, pk);
      if (a.Wp > a.Wo && x == a.Wo - 1) st8(a.x4 + ((fr * a.Ho + y) * a.Wp + a.Wo) * 4, (u32x2){0u, 0u});
    }
  }
}

int vfs_crop_resize_flip_norm_launch(const PipelineArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.V <= 0 || a.T <= 0 || a.Ho <= 0 || a.Wo <= 0) return vfs_set_error(VFS_ERR_SHAPE, "pipeline: empty batch");
  const long long total = (long long)a.B * a.V * a.T * a.Ho * a.Wo;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(crop_resize_flip_norm_kernel, dim3((int)blocks), dim3(256), 0, s, a);
  return vfs_check_launch("crop_resize_flip_norm");
}

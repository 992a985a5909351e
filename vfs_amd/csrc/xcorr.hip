// SiamFC cross-correlation head (§8f rank 4): out[m][i][j] = scale * sum_{u,v,c} z[m % nz][u][v][c] * x[m][i+u][j+v][c]
// - what `_fast_xcorr` computes with a grouped conv2d (projects/siamfc-pytorch/siamfc/heads.py:16-23,51-58:
// x.view(-1, nz*c, h, w) convolved with z, groups = nz, i.e. search feature m is correlated with exemplar m % nz).
// NHWC bf16 operands, fp32 accumulation and output.  The op is tiny (3 x 18 x 18 responses of a 15 x 15 x 512 filter:
// 0.2 GFLOP) and reads the 3 MB search feature from L2: one wave per response element, 16-byte loads along the
// channels, a wave reduction at the end - no MFMA reshaping of what is a bandwidth / latency problem.
#include "vfs_common.h"
#include "vfs_ops.h"

__global__ __launch_bounds__(256) void xcorr_fwd_kernel(XcorrArgs a) {
  const int lane = threadIdx.x & 63;
  const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per output element
  const int ho = a.H - a.Hz + 1, wo = a.W - a.Wz + 1;
  const long long total = (long long)a.nx * ho * wo;
  if (e >= total) return;
  const int j = (int)(e % wo), i = (int)((e / wo) % ho), m = (int)(e / ((long long)wo * ho));
  const bf16_t* z = a.z + (size_t)(m % a.nz) * a.Hz * a.Wz * a.C;
  const bf16_t* x = a.x + ((size_t)m * a.H * a.W + (size_t)i * a.W + j) * a.C;
  const int cv = a.C >> 3;                       // 16-byte chunks per pixel
  const int rowv = a.Wz * cv;                    // chunks of one filter row: contiguous in z, and in x (window row)
  float acc = 0.f;
  for (int u = 0; u < a.Hz; ++u) {
    const bf16_t* zr = z + (size_t)u * a.Wz * a.C;
    const bf16_t* xr = x + (size_t)u * a.W * a.C;
    for (int k = lane; k < rowv; k += 64) {
      float fz[8], fx[8];
      unpack8(ld16(zr + (size_t)k * 8), fz);
      unpack8(ld16(xr + (size_t)k * 8), fx);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += fz[q] * fx[q];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) a.out[e] = acc * a.scale;
}

int vfs_xcorr_fwd_launch(const XcorrArgs& a, hipStream_t s) {
  if (a.nz <= 0 || a.nx <= 0 || a.nx % a.nz) return vfs_set_error(VFS_ERR_SHAPE, "xcorr: nx must be a multiple of nz");
  if (a.C % 8 || a.Hz > a.H || a.Wz > a.W || a.Hz <= 0 || a.Wz <= 0) return vfs_set_error(VFS_ERR_SHAPE, "xcorr: C % 8, filter <= search size");
  const long long total = (long long)a.nx * (a.H - a.Hz + 1) * (a.W - a.Wz + 1);
  if (total >= (1ll << 31)) return vfs_set_error(VFS_ERR_SHAPE, "xcorr: too many outputs");
  hipLaunchKernelGGL(xcorr_fwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a);
  return vfs_check_launch("xcorr_fwd");
}

// One-off hardware probes (run on the GPU box): semantics of ds_read_b64_tr_b16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void probe_tr16(int mode, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int off;   // in shorts
  if (mode == 0) off = l * 4;                                   // lane-linear 8-byte chunks
  else if (mode == 1) off = 0;                                  // uniform address
  else if (mode == 2) off = (l & 15) * 64 + (l >> 4) * 4;       // 16 rows of 64 shorts (128 B), lane group -> column block
  else off = (l & 3) * 4 + ((l >> 2) & 3) * 64 + (l >> 4) * 256;  // 4x4 blocks of 8 B with row stride 128 B
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
  short h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe_tr16, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d MHz mem %zu GB\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.totalGlobalMem >> 30);
  return 0;
}

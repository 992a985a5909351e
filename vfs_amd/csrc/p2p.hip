// SyncBN statistic exchange over xGMI without a collective library call.
//
// The reference runs SyncBN (configs/r*_*.py:9,15 `norm_cfg=dict(type='SyncBN')`, wrapped by MMDistributedDataParallel,
// mmaction/apis/train.py:58-66): every BatchNorm layer all-reduces a few KB of statistics, twice per step (forward sums,
// backward sums) - 114 dependent, latency-bound collectives per ResNet-50 step.  Through RCCL each costs a host call, a
// kernel launch and ~20 us of GPU-side protocol.  Here the exchange is ONE small kernel per BatchNorm and direction:
//
//   window (per rank; fine-grained device memory, mapped into every peer by hipIpc*):
//       u64    flag[SLOTS][MAXW]        epoch stamps, flag[s][q] written by rank q
//       double data[SLOTS][MAXW][MAXN]  data[s][q][:] written by rank q
//   rank r, exchange number e (slot e % SLOTS):
//       push : store its n doubles into data[s][r] of EVERY rank's window (xGMI stores), fence, then flag[s][r] = e there
//       wait : spin on its OWN flag[s][q] == e for all q (local memory), then buf[i] = sum_q data[s][q][i] in rank order
//   Every rank adds the same numbers in the same order: the result is bit-identical on all ranks (deterministic).
//   The exchange counter lives in device memory (`state[0]`), so a recorded launch chain (command tape / hipGraph) replays
//   it unchanged.  A slot is rewritten SLOTS exchanges later; a rank can only push exchange e + 1 after it completed e,
//   which needs every peer's push of e, which they issue after completing e - 1: no rank is ever more than one exchange
//   ahead of a reader, so SLOTS >= 2 is safe (4 used).
//   A bounded spin (spin_limit polls, then state[1] = 1 and garbage out) keeps a lost peer from hanging the GPU.
#include <string.h>

#include "vfs_p2p.h"

// phase: 1 = push, 2 = wait + reduce, 3 = both (the product path; the split exists for single-threaded protocol tests)
__global__ __launch_bounds__(256) void p2p_allreduce_f64_kernel(double* __restrict__ buf, int n, void* const* __restrict__ peers, int rank,
                                                                int world, unsigned long long* __restrict__ state, int phase,
                                                                unsigned long long spin_limit) {
  __shared__ int failed;
  p2p_exchange_body(buf, n, peers, rank, world, state, phase, spin_limit, &failed);
}

// one launch at the head of every launch chain that contains folded exchanges (vfs_p2p.h): state[3] counts the chains
__global__ void p2p_chain_start_kernel(unsigned long long* state) { state[3] = state[3] + 1ull; }
int vfs_p2p_chain_start_launch(unsigned long long* state, hipStream_t s) {
  hipLaunchKernelGGL(p2p_chain_start_kernel, dim3(1), dim3(1), 0, s, state);
  return vfs_check_launch("p2p_chain_start");
}

int vfs_p2p_window_bytes_host(long long* bytes, int* max_doubles, int* max_world) {
  *bytes = (long long)p2p_window_bytes();
  *max_doubles = P2P_MAXN;
  *max_world = P2P_MAXW;
  return VFS_OK;
}

int vfs_p2p_alloc_host(void** ptr) {
  void* p = nullptr;
  // fine-grained: stores from a peer (another process, another GPU) become visible to a RUNNING kernel of this one
  if (hipExtMallocWithFlags(&p, p2p_window_bytes(), hipDeviceMallocFinegrained) != hipSuccess || !p)
    return vfs_set_error(VFS_ERR_LAUNCH, "p2p_alloc: hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed");
  // the zero fill must have LANDED before the handle leaves this process: hipMemset is asynchronous, and a peer's first push
  // arriving before it would have its flag or data cleared (time-out in the self-test -> silent fall-back for the whole run)
  if (hipMemset(p, 0, p2p_window_bytes()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return vfs_set_error(VFS_ERR_LAUNCH, "p2p_alloc: hipMemset / hipDeviceSynchronize failed");
  }
  *ptr = p;
  return VFS_OK;
}

int vfs_p2p_free_host(void* ptr) { return hipFree(ptr) == hipSuccess ? VFS_OK : vfs_set_error(VFS_ERR_LAUNCH, "p2p_free: hipFree failed"); }

int vfs_p2p_export_host(void* ptr, void* handle64) {
  hipIpcMemHandle_t h;
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (hipIpcGetMemHandle(&h, ptr) != hipSuccess) return vfs_set_error(VFS_ERR_LAUNCH, "p2p_export: hipIpcGetMemHandle failed");
  memcpy(handle64, &h, 64);
  return VFS_OK;
}

int vfs_p2p_import_host(const void* handle64, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p)
    return vfs_set_error(VFS_ERR_LAUNCH, "p2p_import: hipIpcOpenMemHandle failed");
  *ptr = p;
  return VFS_OK;
}

int vfs_p2p_unimport_host(void* ptr) {
  return hipIpcCloseMemHandle(ptr) == hipSuccess ? VFS_OK : vfs_set_error(VFS_ERR_LAUNCH, "p2p_unimport: hipIpcCloseMemHandle failed");
}

int vfs_p2p_allreduce_f64_launch(double* buf, int n, void* const* peers, int rank, int world, unsigned long long* state, int phase,
                                 unsigned long long spin_limit, hipStream_t s) {
  if (n <= 0 || n > P2P_MAXN) return vfs_set_error(VFS_ERR_SHAPE, "p2p_allreduce: 1 <= n <= 8192 doubles");
  if (world < 1 || world > P2P_MAXW || rank < 0 || rank >= world) return vfs_set_error(VFS_ERR_ARG, "p2p_allreduce: rank / world (<= 8)");
  hipLaunchKernelGGL(p2p_allreduce_f64_kernel, dim3(1), dim3(256), 0, s, buf, n, peers, rank, world, state, phase, spin_limit);
  return vfs_check_launch("p2p_allreduce_f64");
}

// SyncBN statistic exchange through IPC-mapped windows: layout + the device-side exchange (see csrc/p2p.hip for the protocol).
// Shared by the standalone kernel (p2p.hip) and by the BatchNorm reduction kernels (bn.hip), whose LAST workgroup runs the
// exchange as a tail - no extra launch between "local sums" and "sums over all ranks".
#pragma once
#include "vfs_common.h"

#define P2P_SLOTS 4
#define P2P_MAXW 8
#define P2P_MAXN 8192

__host__ __device__ inline size_t p2p_flag_bytes() { return sizeof(unsigned long long) * P2P_SLOTS * P2P_MAXW; }
__host__ __device__ inline size_t p2p_window_bytes() { return p2p_flag_bytes() + sizeof(double) * P2P_SLOTS * P2P_MAXW * P2P_MAXN; }

__device__ __forceinline__ unsigned long long* p2p_flags(void* win) { return reinterpret_cast<unsigned long long*>(win); }
__device__ __forceinline__ double* p2p_data(void* win) { return reinterpret_cast<double*>(reinterpret_cast<char*>(win) + p2p_flag_bytes()); }

// what a reduction kernel needs to run the exchange as its tail (peers == nullptr: no exchange)
struct P2PTail {
  void* const* peers = nullptr;    // device array of `world` window pointers (peers[rank] = own window)
  unsigned long long* state = nullptr;   // {exchange counter, error flag, "workgroups done" ticket counter (low 32 bits), -}
  unsigned long long spin_limit = 0;
  int rank = 0, world = 1;
};

// One workgroup of 256 threads, all of them call this.  buf[0..n) <- sum over ranks in rank order.  phase: 1 push, 2 wait + reduce.
// buf may have been written by OTHER workgroups of the same launch (agent-scope stores + their release, then this workgroup's
// ticket): it is read with agent-scope loads.
__device__ __forceinline__ void p2p_exchange_body(double* __restrict__ buf, int n, void* const* __restrict__ peers, int rank, int world,
                                                  unsigned long long* __restrict__ state, int phase, unsigned long long spin_limit,
                                                  int* s_failed) {
  // The own contribution never travels through a window (it is read from buf); a thread keeps PB elements in flight: the
  // system-scope accesses are uncached round trips, issued back to back instead of one dependent chain per element
  // (first version: 14-20 us per BatchNorm for 4-8 K doubles; profiles/r03_d_r50_p2p_kernel_stats.csv).
  constexpr int PB = 4;
  const int tid = threadIdx.x;
  const unsigned long long epoch = state[0] + 1;
  const int slot = (int)(epoch % P2P_SLOTS);
  if ((phase & 1) && world > 1) {
    for (int base = tid; base < n; base += 256 * PB) {
      double mine[PB];
#pragma unroll
      for (int j = 0; j < PB; ++j) mine[j] = (base + 256 * j < n) ? vfs_load_agent(buf + base + 256 * j) : 0.0;
      for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        double* dst = p2p_data(peers[p]) + ((size_t)slot * P2P_MAXW + rank) * P2P_MAXN;
#pragma unroll
        for (int j = 0; j < PB; ++j)
          if (base + 256 * j < n) vfs_store_system(dst + base + 256 * j, mine[j]);
      }
    }
    vfs_fence_system();          // this thread's payload stores are performed at system scope ...
    __syncthreads();             // ... for every thread of the workgroup, before any flag goes out
    if (tid < world && tid != rank) vfs_store_system_release(p2p_flags(peers[tid]) + slot * P2P_MAXW + rank, epoch);
  }
  if (phase & 2) {
    if (tid == 0) *s_failed = 0;
    __syncthreads();
    if (tid < world && tid != rank) {
      const unsigned long long* f = p2p_flags(peers[rank]) + slot * P2P_MAXW + tid;
      unsigned long long polls = 0;
      while (vfs_load_system_acquire(f) != epoch) {
        if (++polls > spin_limit) { *s_failed = 1; break; }
        vfs_spin_pause();
      }
    }
    __syncthreads();
    // a peer that never arrived: the sums are POISONED with NaN instead of left as garbage - the statistics, the loss and
    // every gradient of the step become NaN (visible), NaN statistics are kept out of the running statistics (bn.hip) and
    // the guarded optimizer step (vfs_sgd_step's skip word = state[1]) leaves the weights alone
    const bool poisoned = *s_failed != 0;
    if (world > 1) {
      const double* src = p2p_data(peers[rank]) + (size_t)slot * P2P_MAXW * P2P_MAXN;
      for (int base = tid; base < n; base += 256 * PB) {
        double v[P2P_MAXW][PB];
#pragma unroll
        for (int q = 0; q < P2P_MAXW; ++q)
#pragma unroll
          for (int j = 0; j < PB; ++j) {
            const int i = base + 256 * j;
            v[q][j] = (q < world && i < n) ? (q == rank ? vfs_load_agent(buf + i) : vfs_load_system(src + (size_t)q * P2P_MAXN + i)) : 0.0;
          }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
          const int i = base + 256 * j;
          if (i >= n) continue;
          double acc = v[0][j];                      // rank order: the same sum, bit for bit, on every rank
#pragma unroll
          for (int q = 1; q < P2P_MAXW; ++q)
            if (q < world) acc += v[q][j];
          buf[i] = poisoned ? __builtin_nan("") : acc;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      state[0] = epoch;
      if (*s_failed) state[1] = 1;
    }
  }
}

// SyncBN statistic exchange through IPC-mapped windows: layout + the device-side exchange (see csrc/p2p.hip for the protocol).
// Shared by the standalone kernel (p2p.hip) and by the BatchNorm reduction kernels (bn.hip), whose LAST workgroup runs the
// exchange as a tail - no extra launch between "local sums" and "sums over all ranks".
#pragma once
#include "vfs_common.h"

#define P2P_SLOTS 4
#define P2P_MAXW 8
#define P2P_MAXN 8192

#define P2P_MAXSLAB 64      // round 6: per-slab stamps of the exchange folded into the BatchNorm apply kernels (64-channel slabs: C <= 4096)
#define P2P_STATE_WORDS (4 + P2P_MAXSLAB)      // state of the folded exchange: {counter, error, ticket, -, slab_ready[P2P_MAXSLAB]}

// flag[slot][rank][sub]: sub 0 = the whole-buffer exchange (p2p_exchange_body), 1 + y = slab y of a folded exchange
__host__ __device__ inline size_t p2p_flag_bytes() { return sizeof(unsigned long long) * P2P_SLOTS * P2P_MAXW * (1 + P2P_MAXSLAB); }
// data[2 P2P_SLOTS][MAXW][MAXN]: slots 0 .. P2P_SLOTS - 1 = the whole-buffer exchanges, P2P_SLOTS .. = the folded ones (their own ring
// and their own epoch sequence: the two kinds never share a word)
__host__ __device__ inline size_t p2p_window_bytes() { return p2p_flag_bytes() + sizeof(double) * 2 * P2P_SLOTS * P2P_MAXW * P2P_MAXN; }

__device__ __forceinline__ unsigned long long* p2p_flag(void* win, int slot, int rank, int sub = 0) {
  return reinterpret_cast<unsigned long long*>(win) + ((size_t)slot * P2P_MAXW + rank) * (1 + P2P_MAXSLAB) + sub;
}
__device__ __forceinline__ double* p2p_data(void* win) { return reinterpret_cast<double*>(reinterpret_cast<char*>(win) + p2p_flag_bytes()); }

// what a reduction kernel needs to run the exchange as its tail (peers == nullptr: no exchange)
struct P2PTail {
  void* const* peers = nullptr;    // device array of `world` window pointers (peers[rank] = own window)
  unsigned long long* state = nullptr;   // {exchange counter, error flag, "workgroups done" ticket counter (low 32 bits), -}
  unsigned long long spin_limit = 0;
  int rank = 0, world = 1;
  int seq = 0;                     // folded exchanges: number of this exchange inside its launch chain (assigned by the host, the same on every rank)
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
    if (tid < world && tid != rank) vfs_store_system_release(p2p_flag(peers[tid], slot, rank), epoch);
  }
  if (phase & 2) {
    if (tid == 0) *s_failed = 0;
    __syncthreads();
    if (tid < world && tid != rank) {
      const unsigned long long* f = p2p_flag(peers[rank], slot, tid);
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

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the exchange FOLDED into the BatchNorm apply kernels (bn.hip: bn_act_kernel<FIN> / bn_bwd_apply_kernel<FIN>).
// Rounds 3-5 ran it as the tail of a reduction kernel of its own between the convolution and the apply pass - one dependent
// launch more per BatchNorm and direction than the single-GPU path on the 16x16 / 8x8 stages (92 per ResNet-50 step, +12.5 % on one
// GPU).  Here the apply kernel's slab LEAD (the first workgroup of a 64-channel slab, which sums the statistics rows of every
// group anyway) exchanges its slab's n = G x 2 x cslab <= 256 local sums with the peers' leads of the same slab - its own stamp
// flag[slot][rank][1 + slab], its own range of data[P2P_SLOTS + slot][rank][..] - writes the sums over the ranks to `sums` and
// releases the other workgroups of the slab through slab_ready[slab] = epoch (state[4 + slab], device scope); they spin on that
// word (one lane, s_sleep between polls, bounded) instead of summing rows.
// Epoch of a folded exchange = state[3] * 4096 + seq + 1: state[3] counts launch CHAINS (vfs_p2p_chain_start, one tiny launch at
// the head of the forward and of the backward chain - recorded on the command tape like any other call), seq numbers the
// exchanges inside a chain (host-assigned, identical on every rank because every rank issues the same calls in the same order).
// Every workgroup reads a word that an EARLIER launch wrote: no counting of workgroups, no atomics.  (First version: epoch =
// state[0] + 1 published by the last workgroup to start - one device-scope atomic per workgroup on one address, thousands per
// launch: the ResNet-50 step went from 9.1 to 10.8 ms.)  The folded exchanges have their own slot ring and stamps, so the safety
// argument of p2p.hip holds for each kind separately (a rank pushes exchange e + 1 only after ALL its leads completed e).
struct P2PSlab {
  unsigned long long epoch;
  int slab, cslab, C;      // element t of the slab <-> (g, r, ch) = (t / (2 cslab), (t / cslab) & 1, t % cslab) <-> sums[(2 g + r) C + slab cslab + ch]
};
__device__ __forceinline__ int p2p_slab_index(const P2PSlab& sl, int t) {
  const int g = t / (2 * sl.cslab), r = (t / sl.cslab) & 1, ch = t % sl.cslab;
  return (2 * g + r) * sl.C + sl.slab * sl.cslab + ch;
}
__device__ __forceinline__ unsigned long long p2p_fold_epoch(const P2PTail& x) { return x.state[3] * 4096ull + (unsigned long long)x.seq + 1ull; }
// the slab lead, all 256 threads: vals[0..n) (LDS, local sums) <- sums over the ranks in rank order (NaN if a peer never arrived)
__device__ __forceinline__ void p2p_exchange_slab(double* vals, int n, const P2PSlab& sl, const P2PTail& x, int* s_failed) {
  const int tid = threadIdx.x, slot = P2P_SLOTS + (int)(sl.epoch % P2P_SLOTS), fslot = (int)(sl.epoch % P2P_SLOTS), sub = 1 + sl.slab;
  const int gidx = tid < n ? p2p_slab_index(sl, tid) : 0;
  const double mine = tid < n ? vals[tid] : 0.0;
  if (tid == 0) *s_failed = 0;
  if (x.world > 1) {
    if (tid < n)
      for (int p = 0; p < x.world; ++p)
        if (p != x.rank) vfs_store_system(p2p_data(x.peers[p]) + ((size_t)slot * P2P_MAXW + x.rank) * P2P_MAXN + gidx, mine);
    vfs_fence_system();
    __syncthreads();
    if (tid < x.world && tid != x.rank) vfs_store_system_release(p2p_flag(x.peers[tid], fslot, x.rank, sub), sl.epoch);
    if (tid < x.world && tid != x.rank) {
      const unsigned long long* f = p2p_flag(x.peers[x.rank], fslot, tid, sub);
      unsigned long long polls = 0;
      while (vfs_load_system_acquire(f) != sl.epoch) {
        if (++polls > x.spin_limit) { *s_failed = 1; break; }
        vfs_spin_pause();
      }
    }
  }
  __syncthreads();
  const bool poisoned = *s_failed != 0;
  if (tid < n) {
    const double* src = p2p_data(x.peers[x.rank]) + (size_t)slot * P2P_MAXW * P2P_MAXN;
    double acc = 0.0;
    for (int q = 0; q < x.world; ++q) {      // rank order: the same sum, bit for bit, on every rank
      const double v = q == x.rank ? mine : vfs_load_system(src + (size_t)q * P2P_MAXN + gidx);
      acc = q == 0 ? v : acc + v;
    }
    vals[tid] = poisoned ? __builtin_nan("") : acc;
  }
  if (tid == 0 && poisoned) __hip_atomic_store(x.state + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
}
// lead: after it has written the slab's `sums` with agent-scope stores (all threads call); others: p2p_slab_wait
__device__ __forceinline__ void p2p_slab_release(const P2PTail& x, const P2PSlab& sl) {
  vfs_release_workgroup();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(x.state + 4 + sl.slab, sl.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void p2p_slab_wait(const P2PTail& x, const P2PSlab& sl) {
  if (threadIdx.x == 0) {
    unsigned long long polls = 0;
    // (the lead's own wait for the peers is bounded by spin_limit polls of ~0.3 us; these polls are ~4 x as far apart and the lead
    // poisons + releases when it gives up, so this loop ends on its own - the bound only guards against a lead that never ran)
    while (__hip_atomic_load(x.state + 4 + sl.slab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sl.epoch) {
      if (++polls > x.spin_limit) { __hip_atomic_store(x.state + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  __syncthreads();
}

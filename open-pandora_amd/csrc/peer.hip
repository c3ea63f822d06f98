// Peer mailboxes: the latency-class exchanges of the frame-sharded U-Net forward as ONE kernel launch per
// exchange - no RCCL call, no host round trip, capturable inside a HIP graph.
//
// A frame-sharded forward (open-pandora_amd/frame_parallel.py) has 88 temporal-conv stages that each need (i) the
// (T,H,W)-GroupNorm partial sums of every rank of the frame group (256 B) and (ii) the raw boundary frames of the two
// neighbour ranks, plus 17 more GroupNorms that need (i) only.  As RCCL calls these were 2(N-1)+4 point-to-point ops
// per rank and stage, each behind a host-visible wait.  Here every rank owns a MAILBOX in
// fine-grained device memory that its peers map through hipIpc; one kernel per exchange
//   1. writes its partial sums into EVERY peer's mailbox and its first / last frame into the two neighbours'
//      mailboxes (peer writes over xGMI: reads stay local),
//   2. publishes them: every storing workgroup fences at system scope and adds 1 to the receiver's arrival counter,
//   3. polls its OWN mailbox until every source has arrived (bounded: a timeout raises the error word, never hangs),
//   4. sums the partial sums in rank order (bitwise the same on every rank) and copies the halo frames out.
// Two slots alternate by an epoch counter kept in the mailbox itself (a captured graph replays with frozen
// arguments): a peer can be at most one exchange ahead, because its exchange k+1 needs this rank's arrival of k+1,
// which is sent by the kernel AFTER the one that read slot k & 1.
//
// The reference has no counterpart (its only collective, lvdm/common.py:8-14, is never called).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "common.hpp"

namespace pm {

constexpr int PEER_MAX = 16;
constexpr int PEER_CTRL_BYTES = 256;

struct PeerCtrl {
  uint32_t epoch;   // exchanges completed by this rank
  uint32_t error;   // 1: a poll timed out (results of that exchange are garbage, nothing hung)
  uint32_t done;    // ticket of the workgroups that have finished reading (the last one closes the exchange)
  uint32_t pad;
};

struct PeerLayout {  // byte offsets inside a mailbox; identical on every rank
  int64_t slot_bytes, arrive_off, stats_off, lo_off, hi_off;
};

__host__ __device__ inline PeerLayout peer_layout(int world, int64_t nstat_max, int64_t halo_max) {
  PeerLayout l;
  l.arrive_off = 0;                                       // uint32 arrive[world] (own 256-byte line)
  l.stats_off = 256;                                      // float stats[world][nstat_max]
  const int64_t stats_bytes = ((int64_t)world * nstat_max * 4 + 255) & ~(int64_t)255;
  l.lo_off = l.stats_off + stats_bytes;                   // frame before my first (from rank - 1)
  const int64_t hb = (halo_max + 255) & ~(int64_t)255;
  l.hi_off = l.lo_off + hb;                               // frame after my last (from rank + 1)
  l.slot_bytes = l.hi_off + hb;
  return l;
}

struct PeerParams {
  char* mine;
  char* peer[PEER_MAX];
  int rank, world;
  const float* stats;  // [nstat] this rank's partial sums
  int nstat;
  const char* first;   // this rank's first / last frame (halo_bytes each), or nullptr: statistics only
  const char* last;
  int64_t halo_bytes;
  float* totals;       // [nstat] out
  char* lo_out;        // out (nullptr at the clip start / statistics only)
  char* hi_out;
  int64_t nstat_max, halo_max;
  uint32_t timeout_ticks;  // of s_memrealtime (100 MHz)
};

__device__ __forceinline__ void copy16(char* dst, const char* src, int64_t bytes, int tid, int nthreads) {
  const int64_t n16 = bytes >> 4;
  for (int64_t i = tid; i < n16; i += nthreads)
    reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
}

__global__ __launch_bounds__(256) void peer_exchange_kernel(const PeerParams p) {
  __shared__ uint32_t s_flag;
  PeerCtrl* ctrl = reinterpret_cast<PeerCtrl*>(p.mine);
  const PeerLayout L = peer_layout(p.world, p.nstat_max, p.halo_max);
  const uint32_t epoch = __hip_atomic_load(&ctrl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t slot = PEER_CTRL_BYTES + (int64_t)(epoch & 1u) * L.slot_bytes;
  const int G = gridDim.x, b = blockIdx.x;
  const int gtid = b * 256 + threadIdx.x, gthreads = G * 256;
  const bool halo = p.first != nullptr;
  const bool has_lo = halo && p.rank > 0, has_hi = halo && p.rank < p.world - 1;

  // ---- 1. send ----
  if (b == 0) {  // partial sums -> every peer
    for (int r = 0; r < p.world; ++r) {
      if (r == p.rank) continue;
      float* dst = reinterpret_cast<float*>(p.peer[r] + slot + L.stats_off) + (int64_t)p.rank * p.nstat_max;
      for (int j = threadIdx.x; j < p.nstat; j += 256) dst[j] = p.stats[j];
    }
  }
  if (has_lo) copy16(p.peer[p.rank - 1] + slot + L.hi_off, p.first, p.halo_bytes, gtid, gthreads);  // my first = their "after last"
  if (has_hi) copy16(p.peer[p.rank + 1] + slot + L.lo_off, p.last, p.halo_bytes, gtid, gthreads);   // my last = their "before first"
  // ---- 2. publish: every storing thread's stores are released at system scope, then one arrival per workgroup ----
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int r = 0; r < p.world; ++r) {
      if (r == p.rank) continue;
      const bool nb = halo && (r == p.rank - 1 || r == p.rank + 1);
      if (b == 0 || nb) {
        uint32_t* arr = reinterpret_cast<uint32_t*>(p.peer[r] + slot + L.arrive_off) + p.rank;
        __hip_atomic_fetch_add(arr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  // ---- 3. wait for every source (own mailbox: local polls) ----
  if (threadIdx.x == 0) {
    // (fail fast: after one timed-out poll the clip is invalid anyway - pm_peer_status reports it - so later exchanges
    // of the same mailbox do not wait again: a dead peer costs ONE poll bound per clip, not one per exchange)
    uint32_t ok = __hip_atomic_load(&ctrl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < p.world && ok; ++r) {
      if (r == p.rank) continue;
      const bool nb = halo && (r == p.rank - 1 || r == p.rank + 1);
      const uint32_t want = nb ? (uint32_t)G : 1u;
      const uint32_t* arr = reinterpret_cast<const uint32_t*>(p.mine + slot + L.arrive_off) + r;
      while (__hip_atomic_load(arr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if ((uint32_t)(__builtin_amdgcn_s_memrealtime() - t0) > p.timeout_ticks) {
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (!ok) __hip_atomic_store(&ctrl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_flag = ok;
  }
  __syncthreads();
  __threadfence_system();  // (acquire side for the threads that did not poll)
  // ---- 4. consume ----
  if (b == 0) {
    for (int j = threadIdx.x; j < p.nstat; j += 256) {
      float acc = 0.f;
      for (int r = 0; r < p.world; ++r) {  // rank order on every rank: bitwise identical totals
        const float v = (r == p.rank)
                            ? p.stats[j]
                            : __builtin_nontemporal_load(reinterpret_cast<const float*>(p.mine + slot + L.stats_off) +
                                                         (int64_t)r * p.nstat_max + j);
        acc = (r == 0) ? v : acc + v;
      }
      p.totals[j] = acc;
    }
  }
  if (has_lo && p.lo_out) copy16(p.lo_out, p.mine + slot + L.lo_off, p.halo_bytes, gtid, gthreads);
  if (has_hi && p.hi_out) copy16(p.hi_out, p.mine + slot + L.hi_off, p.halo_bytes, gtid, gthreads);
  // ---- close: the last workgroup to finish reading re-arms this slot and advances the epoch ----
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(&ctrl->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (uint32_t)G - 1) {
      uint32_t* arr = reinterpret_cast<uint32_t*>(p.mine + slot + L.arrive_off);
      for (int r = 0; r < p.world; ++r) __hip_atomic_store(arr + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&ctrl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctrl->epoch, epoch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" size_t pm_peer_mailbox_bytes(int world, int64_t nstat_max, int64_t halo_bytes_max) {
  if (world < 1 || world > PEER_MAX || nstat_max < 1 || halo_bytes_max < 0) return 0;
  return (size_t)(PEER_CTRL_BYTES + 2 * peer_layout(world, nstat_max, halo_bytes_max).slot_bytes);
}

// Allocates a zeroed mailbox on the current device and fills `handle` (64 bytes) for the peers.  Fine-grained device
// memory (system-scope atomics and fences act on it without a kernel boundary) where the runtime shares it through
// hipIpc; plain device memory otherwise (*fine_grained tells which).
extern "C" int pm_peer_create(size_t bytes, void** base, void* handle, int* fine_grained) {
  if (!base || !handle || bytes < PEER_CTRL_BYTES) return PM_E_NULL;
  void* p = nullptr;
  int fg = 1;
  hipIpcMemHandle_t h;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipGetLastError();
    if (p) (void)hipFree(p);
    p = nullptr;
    fg = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) return PM_E_WORKSPACE;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
      (void)hipFree(p);
      return PM_E_LAUNCH;
    }
  }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return PM_E_LAUNCH;
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  memcpy(handle, &h, sizeof(h));
  *base = p;
  if (fine_grained) *fine_grained = fg;
  return PM_OK;
}

extern "C" int pm_peer_open(const void* handle, void** base) {
  if (!handle || !base) return PM_E_NULL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  return hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? PM_OK : PM_E_LAUNCH;
}

extern "C" int pm_peer_close(void* base) { return hipIpcCloseMemHandle(base) == hipSuccess ? PM_OK : PM_E_LAUNCH; }
extern "C" int pm_peer_destroy(void* base) { return hipFree(base) == hipSuccess ? PM_OK : PM_E_LAUNCH; }

// Synchronous read of {epoch, error} of this rank's mailbox (tests, and the sampler once per clip).
extern "C" int pm_peer_status(const void* base, int* epoch, int* error) {
  PeerCtrl c;
  if (!base || hipMemcpy(&c, base, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return PM_E_LAUNCH;
  if (epoch) *epoch = (int)c.epoch;
  if (error) *error = (int)c.error;
  return PM_OK;
}

extern "C" int pm_peer_exchange(void* mine, const void* const* peers, int rank, int world, const float* stats,
                                int64_t nstat, const void* first, const void* last, int64_t halo_bytes, float* totals,
                                void* lo_out, void* hi_out, int64_t nstat_max, int64_t halo_bytes_max,
                                double timeout_s, void* stream) {
  if (!mine || !peers || !stats || !totals) return PM_E_NULL;
  if (world < 2 || world > PEER_MAX || rank < 0 || rank >= world) return PM_E_SHAPE;
  if (nstat < 1 || nstat > nstat_max || halo_bytes < 0 || halo_bytes > halo_bytes_max || (halo_bytes & 15)) return PM_E_SHAPE;
  if ((first == nullptr) != (last == nullptr) || (first && halo_bytes == 0)) return PM_E_SHAPE;
  PeerParams p{};
  p.mine = reinterpret_cast<char*>(mine);
  for (int r = 0; r < world; ++r) {
    if (r != rank && !peers[r]) return PM_E_NULL;
    p.peer[r] = reinterpret_cast<char*>(const_cast<void*>(peers[r]));
  }
  p.rank = rank; p.world = world; p.stats = stats; p.nstat = (int)nstat;
  p.first = reinterpret_cast<const char*>(first); p.last = reinterpret_cast<const char*>(last);
  p.halo_bytes = halo_bytes; p.totals = totals;
  p.lo_out = reinterpret_cast<char*>(lo_out); p.hi_out = reinterpret_cast<char*>(hi_out);
  p.nstat_max = nstat_max; p.halo_max = halo_bytes_max;
  const double ticks = (timeout_s > 0 ? timeout_s : 2.0) * 1e8;
  p.timeout_ticks = ticks > 4.0e9 ? 4000000000u : (uint32_t)ticks;
  // one workgroup per 256 KiB of halo payload (both frames), at least one, at most 64: the copies are
  // xGMI- / HBM-bound, the statistics ride with workgroup 0
  int grid = first ? (int)((2 * halo_bytes + (256 << 10) - 1) / (256 << 10)) : 1;
  grid = grid < 1 ? 1 : (grid > 64 ? 64 : grid);
  hipLaunchKernelGGL(peer_exchange_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch();
}

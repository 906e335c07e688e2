// HBM row cache for EmbeddingLocation.MANAGED_CACHING tables (gfx950).
//
// Reference surface: torchrec/distributed/embedding_types.py:57-76 maps the
// `batched_fused_uvm_caching` compute kernel to EmbeddingLocation.MANAGED_CACHING;
// torchrec/distributed/batched_embedding_kernel.py:563,664 call `emb_module.flush()` before
// reading weights; planner/constants.py:26 gives the default cache_load_factor (0.2).  The cache
// itself lives in fbgemm_gpu, which is absent from the reference tree: what is restated here is
// its contract — the table stays in host memory, a software-managed set-associative cache in HBM
// holds the rows in use, results are identical to an uncached table, flush() writes back.
//
// Design for MI355X (wave = 64 lanes):
//  * a set is 64 ways = one wave-wide coalesced tag load (512 B) / LRU load (256 B);
//  * the TBE forward / backward kernels are NOT changed: a prefetch step rewrites the batch's ids of
//    cached features into SLOT numbers of one pseudo-table [cache slots | staging slots] in HBM, and
//    the per-feature metadata of those features points at that pseudo-table.  Duplicates of a row
//    map to the same slot, so the backward's exact (coalescing, deterministic) update is unchanged;
//  * rows that find no evictable way (every way of the set is needed by this very batch) are held
//    in a per-batch staging area behind the cache and written back after the backward, so
//    correctness never depends on capacity (a 1-set cache works);
//  * no float atomics, no locks: a way is claimed by one compare-and-swap on its LRU word
//    (old value -> current iteration); ways touched in this iteration are never victims.
// Prefetch pipeline (all on `stream`, no host sync):
//   linearize (id -> cached-row key) -> stable radix sort -> head flags + scan (dense unique index)
//   -> lookup (16-lane group per unique key probes its set) -> insert (wave per miss: claim a way,
//   write the victim back to host, load the new row) -> remap (slot of every position).
#include <algorithm>

#include "common.hpp"
#include "radix_sort.hpp"

namespace tbe {

constexpr int kWays = 64;
constexpr int kCntStaging = 0, kCntHits = 1, kCntMisses = 2, kCntEvictions = 3, kCntUnique = 4, kCntNumMiss = 5;

struct CacheDev {
  int64_t* tags;
  int32_t* lru;
  float* rows;
  float* state;  // rowwise optimizer state per slot, or nullptr
  int64_t* staging_keys;
  int32_t* counters;
  const int64_t* tab_key_base;
  const uint64_t* tab_weights;
  const uint64_t* tab_state;
  const int32_t* tab_D;
  int32_t num_sets;
  int32_t row_stride;
  int32_t staging_cap;
  int32_t Tc;
};

__device__ __forceinline__ int64_t set_base(uint64_t key, int num_sets) {
  const uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 32;
  return static_cast<int64_t>(h % static_cast<uint64_t>(num_sets)) * kWays;
}

// position p -> feature (largest f with offsets[f*B] <= p), key of cached rows, pass-through of the rest
__global__ __launch_bounds__(256) void cache_linearize_kernel(
    const int64_t* __restrict__ indices, const int64_t* __restrict__ offsets, const int32_t* __restrict__ feat_ctab,
    const int64_t* __restrict__ feat_rows, const int64_t* __restrict__ feat_window,
    const int64_t* __restrict__ tab_key_base, int F, int B, int64_t N,
    uint64_t sentinel, uint64_t* __restrict__ keys, uint64_t* __restrict__ payload, int64_t* __restrict__ remapped) {
  extern __shared__ int64_t fb[];
  for (int i = threadIdx.x; i <= F; i += blockDim.x) fb[i] = offsets[static_cast<int64_t>(i) * B];
  __syncthreads();
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < N;
       p += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int lo = 0, hi = F;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (fb[mid] <= p) lo = mid; else hi = mid;
    }
    const int f = lo;
    const int64_t idx = indices[p];
    const int tc = feat_ctab[f];
    uint64_t key = sentinel;
    if (tc < 0) {
      remapped[p] = idx;  // not cached: the lookup kernels apply the feature's window themselves
    } else {
      int64_t lidx;
      const int cls = classify_id(load_window(feat_rows, feat_window, f), idx, lidx);
      if (cls == kIdLocal) {
        key = static_cast<uint64_t>(tab_key_base[tc] + lidx);
      } else {
        // another shard's row: skipped silently; out of range: the lookup kernels count it (zero row either way)
        remapped[p] = cls == kIdForeign ? TBE_ID_SKIP : -1;
      }
    }
    keys[p] = key;
    payload[p] = static_cast<uint64_t>(p);
  }
}

__global__ __launch_bounds__(256) void cache_flag_kernel(const uint64_t* __restrict__ skeys, int64_t N, uint64_t sentinel,
                                                         int32_t* __restrict__ flags) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const uint64_t k = skeys[j];
  flags[j] = (k != sentinel && (j == 0 || skeys[j - 1] != k)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void cache_compact_kernel(const uint64_t* __restrict__ skeys, const int32_t* __restrict__ flags,
                                                            const int32_t* __restrict__ runidx, int64_t N,
                                                            int64_t* __restrict__ ukey, int32_t* __restrict__ counters) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= N) return;
  if (flags[j]) ukey[runidx[j] - 1] = static_cast<int64_t>(skeys[j]);
  if (j == N - 1) counters[kCntUnique] = runidx[j];
}

// One 16-lane group per unique key: 4 tags per lane = the 64 ways of the key's set.
__global__ __launch_bounds__(256) void cache_lookup_kernel(CacheDev c, const int64_t* __restrict__ ukey,
                                                           int32_t* __restrict__ uslot, int32_t* __restrict__ miss_u,
                                                           int32_t iter) {
  const int nuniq = c.counters[kCntUnique];
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, gl = lane & 15;
  const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  int hits = 0;
  for (int64_t u0 = wave_id * 4; u0 < nuniq; u0 += nwaves * 4) {  // wave-uniform trip count
    const int64_t u = u0 + g;
    const bool active = u < nuniq;
    const int64_t key = active ? ukey[u] : 0;
    const int64_t base = set_base(static_cast<uint64_t>(key), c.num_sets);
    int hit = -1;
    if (active) {
      const int64_t* t = c.tags + base + gl * 4;
      const longlong2 a = *reinterpret_cast<const longlong2*>(t);
      const longlong2 b = *reinterpret_cast<const longlong2*>(t + 2);
      if (a.x == key) hit = 0;
      else if (a.y == key) hit = 1;
      else if (b.x == key) hit = 2;
      else if (b.y == key) hit = 3;
    }
    const uint64_t ball = __ballot(hit >= 0);
    const unsigned gmask = static_cast<unsigned>((ball >> (g * 16)) & 0xffffu);
    int slot = -1;
    if (gmask != 0) {
      const int src = __builtin_ctz(gmask);
      const int h = __shfl(hit, g * 16 + src, kWave);
      slot = static_cast<int>(base) + src * 4 + h;
    }
    if (active && gl == 0) {
      uslot[u] = slot;
      if (slot >= 0) {
        c.lru[slot] = iter;  // protects the row from eviction by this iteration's inserts
        ++hits;
      } else {
        miss_u[atomicAdd(&c.counters[kCntNumMiss], 1)] = static_cast<int32_t>(u);
      }
    }
  }
  if (hits > 0) atomicAdd(&c.counters[kCntHits], hits);
}

__device__ __forceinline__ int table_of(const CacheDev& c, int64_t key, int lane) {
  // Tc is small: lanes test one table each, 64 tables per step
  for (int t0 = 0; t0 < c.Tc; t0 += kWave) {
    const int t = t0 + lane;
    const bool in = t < c.Tc && c.tab_key_base[t] <= key && key < c.tab_key_base[t + 1];
    const uint64_t b = __ballot(in);
    if (b != 0) return t0 + __builtin_ctzll(b);
  }
  return 0;
}

// copies one row (and its rowwise state) between a cache/staging slot and the host table
__device__ __forceinline__ void copy_row(const CacheDev& c, int64_t key, int64_t slot, bool to_host, int lane) {
  const int tc = table_of(c, key, lane);
  const int64_t local = key - c.tab_key_base[tc];
  const int D = c.tab_D[tc];
  float* host = reinterpret_cast<float*>(c.tab_weights[tc]) + local * D;
  float* dev = c.rows + slot * c.row_stride;
  const bool vec = ((D & 3) == 0) && ((c.row_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(host) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(c.rows) & 15) == 0);
  if (vec) {
    for (int d = lane * 4; d < D; d += kWave * 4) {
      if (to_host) st4(host + d, ld4(dev + d)); else st4(dev + d, ld4(host + d));
    }
  } else {
    for (int d = lane; d < D; d += kWave) {
      if (to_host) host[d] = dev[d]; else dev[d] = host[d];
    }
  }
  if (c.state != nullptr && lane == 0) {
    float* hs = reinterpret_cast<float*>(c.tab_state[tc]) + local;
    if (to_host) *hs = c.state[slot]; else c.state[slot] = *hs;
  }
}

// One wave per missed key.
__global__ __launch_bounds__(256) void cache_insert_kernel(CacheDev c, const int64_t* __restrict__ ukey,
                                                           int32_t* __restrict__ uslot, const int32_t* __restrict__ miss_u,
                                                           int32_t iter) {
  const int nmiss = c.counters[kCntNumMiss];
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  int evictions = 0, misses = 0;
  for (int64_t m = wave_id; m < nmiss; m += nwaves) {
    const int u = miss_u[m];
    const int64_t key = ukey[u];
    const int64_t base = set_base(static_cast<uint64_t>(key), c.num_sets);
    int64_t slot = -1;
    for (int attempt = 0; attempt < 4 * kWays; ++attempt) {
      const int l = __hip_atomic_load(c.lru + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // candidates: ways not touched in this iteration; order by (last use, lane)
      uint32_t ord = (l < iter) ? ((static_cast<uint32_t>(l + 1) << 6) | static_cast<uint32_t>(lane)) : 0xffffffffu;
      uint32_t best = ord;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) best = min(best, static_cast<uint32_t>(__shfl_xor(static_cast<int>(best), o, kWave)));
      if (best == 0xffffffffu) break;  // every way is in use by this batch -> staging
      const int v = static_cast<int>(best & 63u);
      int ok = 0;
      if (lane == v) ok = atomicCAS(c.lru + base + v, l, iter) == l ? 1 : 0;
      ok = __shfl(ok, v, kWave);
      if (ok) {
        slot = base + v;
        break;
      }
    }
    ++misses;
    if (slot >= 0) {
      const int64_t old = c.tags[slot];  // written before this kernel: a claimed way is never re-claimed within it
      if (old >= 0) {
        copy_row(c, old, slot, /*to_host=*/true, lane);
        ++evictions;
      }
      copy_row(c, key, slot, /*to_host=*/false, lane);
      if (lane == 0) c.tags[slot] = key;
    } else {
      int k = 0;
      if (lane == 0) k = atomicAdd(&c.counters[kCntStaging], 1);
      k = __shfl(k, 0, kWave);
      if (k < c.staging_cap) {  // the host sizes staging_cap >= ids per batch, so this always holds
        slot = static_cast<int64_t>(c.num_sets) * kWays + k;
        copy_row(c, key, slot, /*to_host=*/false, lane);
        if (lane == 0) c.staging_keys[k] = key;
      }
    }
    if (lane == 0) uslot[u] = static_cast<int32_t>(slot);
  }
  if (lane == 0) {
    if (misses) atomicAdd(&c.counters[kCntMisses], misses);
    if (evictions) atomicAdd(&c.counters[kCntEvictions], evictions);
  }
}

__global__ __launch_bounds__(256) void cache_remap_kernel(const uint64_t* __restrict__ skeys, const uint64_t* __restrict__ spay,
                                                          const int32_t* __restrict__ runidx, const int32_t* __restrict__ uslot,
                                                          int64_t N, uint64_t sentinel, int64_t* __restrict__ remapped) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= N) return;
  if (skeys[j] == sentinel) return;
  remapped[spay[j]] = static_cast<int64_t>(uslot[runidx[j] - 1]);
}

// staging rows of the finished batch -> host
__global__ __launch_bounds__(256) void cache_staging_writeback_kernel(CacheDev c) {
  const int n = min(c.counters[kCntStaging], c.staging_cap);
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  for (int64_t k = wave_id; k < n; k += nwaves)
    copy_row(c, c.staging_keys[k], static_cast<int64_t>(c.num_sets) * kWays + k, /*to_host=*/true, lane);
}

__global__ __launch_bounds__(256) void cache_flush_kernel(CacheDev c, int invalidate) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  const int64_t nslots = static_cast<int64_t>(c.num_sets) * kWays;
  for (int64_t s = wave_id; s < nslots; s += nwaves) {
    const int64_t tag = c.tags[s];
    if (tag >= 0) copy_row(c, tag, s, /*to_host=*/true, lane);
    if (invalidate && lane == 0) {
      c.tags[s] = -1;
      c.lru[s] = -1;
    }
  }
}

struct PrefetchWorkspace {
  uint64_t *k0, *k1, *p0, *p1;
  int32_t *flags, *runidx, *uslot, *miss_u;
  int64_t* ukey;
  void* scan_ws;
  size_t scan_bytes;
  RadixWorkspace sort;
  size_t total;
};

static void carve_prefetch(void* ws, int64_t N, int key_bits, PrefetchWorkspace* w) {
  Carver c(ws);
  w->k0 = c.take<uint64_t>(N);
  w->k1 = c.take<uint64_t>(N);
  w->p0 = c.take<uint64_t>(N);
  w->p1 = c.take<uint64_t>(N);
  w->flags = c.take<int32_t>(N);
  w->runidx = c.take<int32_t>(N);
  w->uslot = c.take<int32_t>(N);
  w->miss_u = c.take<int32_t>(N);
  w->ukey = c.take<int64_t>(N);
  w->scan_bytes = tbe_cumsum_workspace_bytes(N);
  w->scan_ws = c.take_bytes(w->scan_bytes);
  const size_t sort_bytes = radix_carve(nullptr, N, key_bits).bytes;
  w->sort = radix_carve(c.take_bytes(sort_bytes), N, key_bits);
  w->total = c.total();
}

static int to_dev(const tbe_cache_desc* d, CacheDev* c) {
  TBE_REQUIRE(d != nullptr, "tbe_cache: null descriptor");
  TBE_REQUIRE(d->tags && d->lru && d->rows && d->staging_keys && d->counters, "tbe_cache: null cache storage");
  TBE_REQUIRE(d->tab_key_base && d->tab_weights && d->tab_D && d->num_tables > 0, "tbe_cache: null table metadata");
  TBE_REQUIRE(d->num_sets > 0 && d->row_stride > 0 && d->staging_cap >= 0, "tbe_cache: bad geometry");
  TBE_REQUIRE(static_cast<int64_t>(d->num_sets) * kWays + d->staging_cap < (1ll << 31), "tbe_cache: too many slots");
  TBE_REQUIRE(d->state == nullptr || d->tab_state != nullptr, "tbe_cache: state cache without host state");
  c->tags = d->tags;
  c->lru = d->lru;
  c->rows = d->rows;
  c->state = d->state;
  c->staging_keys = d->staging_keys;
  c->counters = d->counters;
  c->tab_key_base = d->tab_key_base;
  c->tab_weights = d->tab_weights;
  c->tab_state = d->tab_state;
  c->tab_D = d->tab_D;
  c->num_sets = d->num_sets;
  c->row_stride = d->row_stride;
  c->staging_cap = d->staging_cap;
  c->Tc = d->num_tables;
  return TBE_OK;
}

}  // namespace tbe

using namespace tbe;

extern "C" size_t tbe_cache_prefetch_workspace_bytes(int64_t N, int32_t key_bits) {
  if (N <= 0) return 256;
  if (N >= kSortMaxPairs) return 0;  // not sortable in one call (tbe_cache_prefetch says why)
  PrefetchWorkspace w;
  carve_prefetch(nullptr, N, key_bits, &w);
  return w.total;
}

extern "C" int tbe_cache_prefetch(const tbe_cache_desc* desc, const int32_t* feat_cached_table, const int64_t* feat_rows,
                                  int32_t F, int32_t B, const int64_t* indices, int64_t N, const int64_t* offsets,
                                  int32_t key_bits, int32_t iteration, int64_t* remapped_indices, void* workspace,
                                  size_t workspace_bytes, const int64_t* feat_window, void* stream) {
  CacheDev c;
  int rc = to_dev(desc, &c);
  if (rc != TBE_OK) return rc;
  TBE_REQUIRE(F > 0 && B >= 0 && N >= 0, "tbe_cache_prefetch: bad sizes");
  TBE_REQUIRE(N < kSortMaxPairs, "tbe_cache_prefetch: N = %lld ids in one call; the limit is 2^29 - 1 (the pair sort's count field)",
              static_cast<long long>(N));
  TBE_REQUIRE(key_bits >= 1 && key_bits <= 62, "tbe_cache_prefetch: key_bits=%d", key_bits);
  TBE_REQUIRE(iteration >= 0, "tbe_cache_prefetch: iteration < 0");
  TBE_REQUIRE(c.staging_cap >= N, "tbe_cache_prefetch: staging_cap (%d) must be >= N (%lld)", c.staging_cap, (long long)N);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // staging count + per-batch miss counter restart; hit / miss / eviction totals accumulate
  if (hipMemsetAsync(c.counters + kCntStaging, 0, sizeof(int32_t), st) != hipSuccess ||
      hipMemsetAsync(c.counters + kCntUnique, 0, 2 * sizeof(int32_t), st) != hipSuccess) {
    set_error("tbe_cache_prefetch: hipMemsetAsync failed");
    return TBE_ERR_LAUNCH;
  }
  if (N == 0 || B == 0) return TBE_OK;
  TBE_REQUIRE(feat_cached_table && feat_rows && indices && offsets && remapped_indices && workspace,
              "tbe_cache_prefetch: null pointer");
  TBE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "tbe_cache_prefetch: workspace must be 256-B aligned");
  PrefetchWorkspace w;
  carve_prefetch(workspace, N, key_bits, &w);
  if (w.total > workspace_bytes) {
    set_error("tbe_cache_prefetch: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return TBE_ERR_WORKSPACE;
  }
  const uint64_t sentinel = (1ull << key_bits) - 1ull;
  const size_t lds = (static_cast<size_t>(F) + 1) * sizeof(int64_t);
  TBE_REQUIRE(lds <= 60000, "tbe_cache_prefetch: too many features (%d)", F);
  const unsigned gridN = static_cast<unsigned>((N + 255) / 256);
  hipLaunchKernelGGL(cache_linearize_kernel, dim3(std::min<unsigned>(gridN, 256 * 16)), dim3(256), lds, st, indices, offsets,
                     feat_cached_table, feat_rows, feat_window, c.tab_key_base, F, B, N, sentinel, w.k0, w.p0, remapped_indices);
  TBE_CHECK_LAUNCH("tbe_cache_prefetch linearize");
  const int where = radix_sort_pairs<uint64_t, uint64_t>(w.k0, w.k1, w.p0, w.p1, N, key_bits, w.sort, st);
  if (where < 0) return where;
  const bool in_second = (radix_passes(key_bits) & 1) != 0;
  const uint64_t* sk = in_second ? w.k1 : w.k0;
  const uint64_t* sp = in_second ? w.p1 : w.p0;
  hipLaunchKernelGGL(cache_flag_kernel, dim3(gridN), dim3(256), 0, st, sk, N, sentinel, w.flags);
  TBE_CHECK_LAUNCH("tbe_cache_prefetch flags");
  rc = tbe_cumsum(w.flags, w.runidx, N, 4, /*inclusive*/ 1, w.scan_ws, w.scan_bytes, stream);
  if (rc != TBE_OK) return rc;
  hipLaunchKernelGGL(cache_compact_kernel, dim3(gridN), dim3(256), 0, st, sk, w.flags, w.runidx, N, w.ukey, c.counters);
  TBE_CHECK_LAUNCH("tbe_cache_prefetch compact");
  const unsigned lgrid = static_cast<unsigned>(std::min<int64_t>((N + 15) / 16, 256 * 8));
  hipLaunchKernelGGL(cache_lookup_kernel, dim3(lgrid), dim3(256), 0, st, c, w.ukey, w.uslot, w.miss_u, iteration);
  TBE_CHECK_LAUNCH("tbe_cache_prefetch lookup");
  const unsigned igrid = static_cast<unsigned>(std::min<int64_t>((N + 3) / 4, 256 * 8));
  hipLaunchKernelGGL(cache_insert_kernel, dim3(igrid), dim3(256), 0, st, c, w.ukey, w.uslot, w.miss_u, iteration);
  TBE_CHECK_LAUNCH("tbe_cache_prefetch insert");
  hipLaunchKernelGGL(cache_remap_kernel, dim3(gridN), dim3(256), 0, st, sk, sp, w.runidx, w.uslot, N, sentinel,
                     remapped_indices);
  TBE_CHECK_LAUNCH("tbe_cache_prefetch remap");
  return TBE_OK;
}

extern "C" int tbe_cache_writeback_staging(const tbe_cache_desc* desc, void* stream) {
  CacheDev c;
  int rc = to_dev(desc, &c);
  if (rc != TBE_OK) return rc;
  if (c.staging_cap == 0) return TBE_OK;
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((static_cast<int64_t>(c.staging_cap) + 3) / 4, 256 * 4));
  hipLaunchKernelGGL(cache_staging_writeback_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), c);
  TBE_CHECK_LAUNCH("tbe_cache_writeback_staging");
  return TBE_OK;
}

extern "C" int tbe_cache_flush(const tbe_cache_desc* desc, int32_t invalidate, void* stream) {
  CacheDev c;
  int rc = to_dev(desc, &c);
  if (rc != TBE_OK) return rc;
  const int64_t nslots = static_cast<int64_t>(c.num_sets) * kWays;
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>((nslots + 3) / 4, 256 * 8));
  hipLaunchKernelGGL(cache_flush_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), c, invalidate);
  TBE_CHECK_LAUNCH("tbe_cache_flush");
  return TBE_OK;
}

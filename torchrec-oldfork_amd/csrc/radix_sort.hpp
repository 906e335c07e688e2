// Hand-written stable LSD radix sort of (key, payload) pairs for gfx950: ONE launch per digit pass.
// Used by the TBE backward to group a batch's contributions by table row (tbe_backward.hip) and by
// the row cache's prefetch (tbe_cache.hip).
//
// Digits of up to 10 bits: passes = ceil(key_bits / 10), digit width = ceil(key_bits / passes)
// (28-bit Criteo row keys: 3 passes of 10 bits).  Launches per sort: 1 memset + 1 totals + `passes`:
//
//   radix_totals_kernel : digit totals of pass 0 (LDS counters with a one-lane fast path when a whole wave
//                         holds one digit, as the ids of 3-row tables do), flushed with integer global
//                         atomics.  The totals of pass p + 1 are counted by pass p's kernel on the keys
//                         it holds anyway (totals do not depend on the order of the keys).
//   radix_pass_kernel   : one workgroup (8 waves) per SEGMENT of the input (<= 256 segments).  It
//     1. takes a ticket (segment index = arrival order, so a workgroup only ever waits on
//        workgroups that already run: forward progress needs no co-residency of the grid),
//     2. loads its keys into registers (started before the ticket returns, for the segment the
//        ticket will most likely name) and ranks them per wave with a match-any built from one ballot
//        per digit bit + one returning LDS add per peer group (stable: order = wave, round, lane),
//     3. publishes its digit histogram as ONE row of 4-byte {pass tag | count} words (write-through
//        stores; the word is its own ready flag, so there is no fence and no separate flag),
//     4. sums its predecessors' rows in two levels (groups of 16 segments; a group's last segment
//        publishes the group sum): <= 30 rows, 8-B L1-bypassing loads all in flight at once, re-polling
//        the words whose tag is not this pass's yet -- two dependent hops, not the serial chain that
//        makes a decoupled look-back slow when the whole grid is resident at once,
//     5. places its pairs in LDS in digit order and writes them out as contiguous runs.
//   Segments longer than one tile (N > 256 * 8192) are histogrammed first and re-read tile by tile.
//
// Equal keys keep their input order, which is what makes the backward's summation order a function of
// the input only.  No float atomics anywhere; integer atomics only on counters whose final value is
// order-independent.
#pragma once
#include <algorithm>
#include <type_traits>

#include "common.hpp"

namespace tbe {

constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kSortMaxBlocks = 256;  // segments per pass
constexpr int kSortGroup = 16;       // segments per group of the two-level predecessor sum
constexpr int kMaxDigitBits = 10;
constexpr int kMaxRadix = 1 << kMaxDigitBits;
constexpr int kMaxPasses = 7;
constexpr int kTagShift = 29;  // word = (pass + 1) << 29 | count ; count < 2^29
constexpr uint32_t kCountMask = (1u << kTagShift) - 1u;
constexpr int64_t kSortMaxPairs = 1ll << kTagShift;  // pairs per sort (the count field of a histogram word)
constexpr int kSpinLimit = 1 << 20;  // polls per thread before it gives up (~1 s)

// development aid: when set (tbe_debug_set_sort_stamps), every segment writes 8 wall-clock stamps (100 MHz) per pass
static uint64_t* g_sort_stamps = nullptr;
#define TBE_SORT_STAMP(i)                                                                          \
  do {                                                                                             \
    if (a.stamps != nullptr && threadIdx.x == 0)                                                   \
      a.stamps[(static_cast<size_t>(a.pass) * kSortMaxBlocks + b) * 8 + (i)] = wall_clock64();     \
  } while (0)

// Spin-wait give-ups (a predecessor never published within kSpinLimit polls): nothing hangs, but the prefix sums — and
// with them the sorted order — are garbage then.  Every give-up is written to the library's FAULT WORD (error.cpp: a
// line of GPU-mapped host memory), which the host side reads without a sync (tbe_fault_status) and turns into an
// exception at its next check point (fbgemm_gpu/_lib.py raise_on_faults): wrong results are never silent.
uint32_t* fault_word_device();  // error.cpp
__device__ __forceinline__ void report_sort_giveup(uint32_t* fault) {
  __hip_atomic_store(&fault[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_fetch_add(&fault[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct RadixPlan {
  int passes;
  int bits;  // digit width of every pass
};
inline RadixPlan radix_plan(int key_bits) {
  RadixPlan p;
  p.passes = (key_bits + kMaxDigitBits - 1) / kMaxDigitBits;
  p.bits = (key_bits + p.passes - 1) / p.passes;
  return p;
}
inline int radix_passes(int key_bits) { return radix_plan(key_bits).passes; }

struct RadixGeom {
  int rounds;           // rounds of 64 keys per wave in a tile: 4, 6, 8, 10, 12, 14 or 16
  int64_t ntiles;
  int blocks;           // segments
  int tiles_per_block;
};
inline RadixGeom radix_geom(int64_t N, size_t pair_bytes) {
  const int max_rounds = pair_bytes <= 8 ? 16 : 8;  // LDS staging of one tile: <= 64 KB
  RadixGeom g;
  g.rounds = 4;  // the smallest tile that needs no more than kSortMaxBlocks segments: every CU gets work
  while (g.rounds < max_rounds && (N + kSortThreads * g.rounds - 1) / (kSortThreads * g.rounds) > kSortMaxBlocks)
    g.rounds += 2;
  const int64_t tile = static_cast<int64_t>(kSortThreads) * g.rounds;
  g.ntiles = (N + tile - 1) / tile;
  g.tiles_per_block = static_cast<int>((g.ntiles + kSortMaxBlocks - 1) / kSortMaxBlocks);
  if (g.tiles_per_block < 1) g.tiles_per_block = 1;
  g.blocks = static_cast<int>((g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
  return g;
}

struct RadixWorkspace {
  uint32_t* state;    // [tickets 16 | totals passes*radix | group table | table], zeroed per sort
  uint32_t* tickets;  // [16]
  uint32_t* totals;   // [passes][radix]
  uint32_t* group_table;  // [kSortMaxBlocks / kSortGroup][radix]
  uint32_t* table;    // [kSortMaxBlocks][radix]
  size_t bytes;
};

// words of the state block radix_sort_pairs needs zeroed (it does so itself unless told otherwise)
inline size_t radix_state_words(int64_t N, int key_bits, size_t pair_bytes) {
  const RadixPlan pl = radix_plan(key_bits);
  const RadixGeom g = radix_geom(N, pair_bytes);
  return 16 + (static_cast<size_t>(1) << pl.bits) * (pl.passes + kSortMaxBlocks / kSortGroup + g.blocks);
}
constexpr int kSortStateZeroed = 1;  // flag: the caller has zeroed radix_state_words() words of ws.state on the stream

inline RadixWorkspace radix_carve(void* base, int64_t N, int key_bits) {
  (void)N;
  const RadixPlan pl = radix_plan(key_bits);
  const size_t radix = static_cast<size_t>(1) << pl.bits;
  Carver c(base);
  RadixWorkspace w;
  constexpr size_t kGroups = kSortMaxBlocks / kSortGroup;
  w.state = c.take<uint32_t>(16 + radix * pl.passes + radix * kGroups + radix * kSortMaxBlocks);
  w.tickets = w.state;
  w.totals = w.state ? w.state + 16 : nullptr;
  w.group_table = w.state ? w.state + 16 + radix * pl.passes : nullptr;
  w.table = w.state ? w.state + 16 + radix * (pl.passes + kGroups) : nullptr;
  w.bytes = c.total();
  return w;
}

// Lanes of this wave whose `digit` (< 2^kMaxDigitBits) equals mine, as two 32-bit halves (invalid lanes: 0).
// One ballot per digit bit; per bit: v_bfe_i32 (bit -> 0 / ~0), v_cmp (ballot), and one v_bitop3 per half
// computing p & ~(ballot ^ mine).  The trip count is fixed so that the rounds of a tile unroll and interleave.
__device__ __forceinline__ void match_digit(unsigned digit, bool valid, uint32_t& plo, uint32_t& phi) {
  const unsigned long long vm = __ballot(valid);
  plo = static_cast<uint32_t>(vm);
  phi = static_cast<uint32_t>(vm >> 32);
#pragma unroll
  for (int bit = 0; bit < kMaxDigitBits; ++bit) {
    const int m = __builtin_amdgcn_sbfe(static_cast<int>(digit), bit, 1);
    const unsigned long long b = __ballot(m < 0);
    plo = __builtin_amdgcn_bitop3_b32(plo, static_cast<uint32_t>(b), static_cast<uint32_t>(m), 0x90);
    phi = __builtin_amdgcn_bitop3_b32(phi, static_cast<uint32_t>(b >> 32), static_cast<uint32_t>(m), 0x90);
  }
  if (!valid) plo = phi = 0u;
}

// Digit totals of the first `passes` passes (independent of key order, so keys are read as 16-B vectors in
// any order, several loads in flight); LDS atomics, then one integer global atomic per non-zero counter.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_totals_kernel(const KeyT* __restrict__ keys, int64_t N, int passes,
                                                                   int bits, int key_bits, uint32_t* __restrict__ totals) {
  __shared__ uint32_t h[kMaxPasses * kMaxRadix];
  constexpr int V = 16 / sizeof(KeyT);  // keys per 16-B vector
  constexpr int U = 4;                  // vectors in flight per thread
  const int radix = 1 << bits;
  for (int i = threadIdx.x; i < passes * radix; i += kSortThreads) h[i] = 0;
  __syncthreads();
  auto count = [&](KeyT k, bool valid) {
    if (!valid) return;
    for (int p = 0; p < passes; ++p) {
      // the last pass only looks at the bits below key_bits
      const int width = min(bits, key_bits - bits * p);
      atomicAdd(&h[p * radix + static_cast<unsigned>((k >> (bits * p)) & static_cast<KeyT>((1u << width) - 1u))], 1u);
    }
  };
  const int64_t nvec = (reinterpret_cast<uintptr_t>(keys) & 15) == 0 ? N / V : 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kSortThreads;
  using Vec = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
  const Vec* kv = reinterpret_cast<const Vec*>(keys);
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kSortThreads; base < nvec; base += stride * U) {  // wave-uniform
    Vec x[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * stride + threadIdx.x;
      ok[u] = i < nvec;
      if (ok[u]) x[u] = kv[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (sizeof(KeyT) == 4) {
        count(x[u].x, ok[u]);
        count(x[u].y, ok[u]);
        count(x[u].z, ok[u]);
        count(x[u].w, ok[u]);
      } else {
        count(x[u].x, ok[u]);
        count(x[u].y, ok[u]);
      }
    }
  }
  // the keys behind the last whole vector (all of them if the array is not 16-B aligned)
  for (int64_t base = nvec * V + static_cast<int64_t>(blockIdx.x) * kSortThreads; base < N; base += stride) {
    const int64_t i = base + threadIdx.x;
    const bool valid = i < N;
    count(valid ? keys[i] : static_cast<KeyT>(0), valid);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < passes * radix; i += kSortThreads) {
    const uint32_t c = h[i];
    if (c != 0u) atomicAdd(&totals[i], c);
  }
}

// Exclusive scan over `radix` (<= 1024) values held two per thread (entries 2t, 2t+1); returns the
// exclusive prefix of entry 2t (entry 2t+1's is that + a).  Contains two workgroup barriers.
__device__ __forceinline__ uint32_t block_excl_scan_pairs(uint32_t a, uint32_t b, uint32_t* s_scan) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t s = a + b;
  uint32_t inc = s;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o, kWave);
    if (lane >= o) inc += y;
  }
  __syncthreads();  // s_scan reuse
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  uint32_t add = 0;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w)
    if (w < wave) add += s_scan[w];
  return add + inc - s;
}

template <typename KeyT, typename ValT>
struct RadixPassArgs {
  const KeyT* keys_in;
  const ValT* vals_in;
  KeyT* keys_out;
  ValT* vals_out;
  int64_t N;
  int shift;
  int bits;
  int width;       // bits of this pass's digit that lie below key_bits (<= bits)
  int next_width;  // the same for the NEXT pass's digit, 0 if this is the last pass
  int pass;
  int tiles_per_block;
  uint32_t* table;         // [blocks][radix] level 0: one histogram row per segment
  uint32_t* group_table;   // [blocks / kSortGroup][radix] level 1: one sum row per complete group
  const uint32_t* totals;  // [radix] digit totals of this pass
  uint32_t* next_totals;   // [radix] digit totals of the next pass, accumulated by this one (or nullptr)
  uint32_t* ticket;        // this pass's ticket counter
  uint64_t* stamps;        // debug: [passes][kSortMaxBlocks][8] or nullptr
  uint32_t* fault;         // the library's fault word (never nullptr)
};

template <typename KeyT, typename ValT, int ROUNDS>
__global__ __launch_bounds__(kSortThreads) void radix_pass_kernel(const RadixPassArgs<KeyT, ValT> a) {
  constexpr int TILE = kSortThreads * ROUNDS;
  __shared__ uint32_t wave_cnt[kSortWaves][kMaxRadix];  // per-wave digit counts, then tile-local wave offsets
  __shared__ uint32_t s_hist[kMaxRadix];                // digit counts of the current tile
  __shared__ uint32_t s_seg[kMaxRadix];                 // digit counts of the whole segment (T > 1)
  __shared__ uint32_t s_next[kMaxRadix];                // this segment's counts of the NEXT pass's digit
  __shared__ uint32_t s_gbase[kMaxRadix];               // global position of this block's next key of digit d
  __shared__ uint32_t s_goff[kMaxRadix];                // s_gbase - (tile-local start of digit d)
  __shared__ uint32_t s_part[2 * kSortThreads];         // partial predecessor sums [nsplit][radix]
  __shared__ uint32_t s_scan[kSortWaves];
  __shared__ uint32_t s_block;
  __shared__ KeyT stage_k[TILE];
  __shared__ ValT stage_v[TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int radix = 1 << a.bits;
  const KeyT dmask = static_cast<KeyT>((1u << a.width) - 1u);
  const KeyT nmask = static_cast<KeyT>((1u << a.next_width) - 1u);
  const int nshift = a.shift + a.bits;
  const bool count_next = a.next_totals != nullptr;
  const uint32_t epoch = static_cast<uint32_t>(a.pass + 1);
  const int T = a.tiles_per_block;

  KeyT k[ROUNDS];
  ValT v[ROUNDS];
  uint32_t wrank[ROUNDS];  // rank of the key among the keys of its digit in this wave's share of the tile

  auto load_tile = [&](int64_t tile_begin, int64_t end) {
    const int64_t wbase = tile_begin + static_cast<int64_t>(wave) * (ROUNDS * kWave);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int64_t i = wbase + r * kWave + lane;
      const bool valid = i < end;
      k[r] = valid ? a.keys_in[i] : static_cast<KeyT>(0);
      v[r] = valid ? a.vals_in[i] : static_cast<ValT>(0);
    }
  };

  // ---- ticket: segment index = arrival order ----------------------------------------------------------
  const uint64_t t_start = a.stamps != nullptr ? wall_clock64() : 0ull;
  if (tid == 0) s_block = atomicAdd(a.ticket, 1u);
  for (int d = tid; d < radix; d += kSortThreads) {
    s_seg[d] = 0;
    s_next[d] = 0;
  }
  for (int d = lane; d < radix; d += kWave) wave_cnt[wave][d] = 0;
  // digit base inputs: this pass's totals (complete: written by the previous launch)
  const uint32_t t0 = (2 * tid < radix) ? a.totals[2 * tid] : 0u;
  const uint32_t t1 = (2 * tid + 1 < radix) ? a.totals[2 * tid + 1] : 0u;
  __syncthreads();
  const uint32_t b = s_block;
  TBE_SORT_STAMP(0);
  if (a.stamps != nullptr && tid == 0) a.stamps[(static_cast<size_t>(a.pass) * kSortMaxBlocks + b) * 8 + 7] = t_start;
  const int64_t seg_begin = static_cast<int64_t>(b) * T * TILE;
  const int64_t seg_end = min(a.N, seg_begin + static_cast<int64_t>(T) * TILE);
  if (T == 1) load_tile(seg_begin, seg_end);

  // ranks the tile held in k[]: per-wave digit counts -> tile histogram s_hist, wave_cnt[w][d] = number of
  // keys of digit d in waves < w, wrank[r] = rank inside the wave.  (wave_cnt must be zero on entry.)
  auto rank_tile = [&](int64_t tile_begin) {
    const int64_t wbase = tile_begin + static_cast<int64_t>(wave) * (ROUNDS * kWave);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const bool valid = wbase + r * kWave + lane < seg_end;
      const unsigned digit = static_cast<unsigned>((k[r] >> a.shift) & dmask);
      uint32_t plo, phi;
      match_digit(digit, valid, plo, phi);
      // my rank among this round's peers; the leader (rank 0) adds the group's size and learns how many
      // keys of the digit the wave's earlier rounds held; same-wave LDS atomics execute in program order
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
      uint32_t before = 0;
      if (valid && rank == 0u) before = atomicAdd(&wave_cnt[wave][digit], static_cast<uint32_t>(__popc(plo) + __popc(phi)));
      const int leader = valid ? (plo != 0u ? __builtin_ctz(plo) : 32 + __builtin_ctz(phi)) : lane;
      before = static_cast<uint32_t>(__shfl(static_cast<int>(before), leader, kWave));
      wrank[r] = before + rank;
      if (count_next && T == 1 && valid) atomicAdd(&s_next[static_cast<unsigned>((k[r] >> nshift) & nmask)], 1u);
    }
    __syncthreads();
    for (int d = tid; d < radix; d += kSortThreads) {
      uint32_t off = 0;
#pragma unroll
      for (int w = 0; w < kSortWaves; ++w) {
        const uint32_t c = wave_cnt[w][d];
        wave_cnt[w][d] = off;
        off += c;
      }
      s_hist[d] = off;
    }
    __syncthreads();
  };

  if (T == 1) {
    rank_tile(seg_begin);
  } else {
    for (int64_t i0 = seg_begin; i0 < seg_end; i0 += kSortThreads) {  // block-uniform trip count
      const int64_t i = i0 + tid;
      const bool valid = i < seg_end;
      if (valid) {
        const KeyT key = a.keys_in[i];
        atomicAdd(&s_seg[static_cast<unsigned>((key >> a.shift) & dmask)], 1u);
        if (count_next) atomicAdd(&s_next[static_cast<unsigned>((key >> nshift) & nmask)], 1u);
      }
    }
    __syncthreads();
  }
  const uint32_t* seg_hist = (T == 1) ? s_hist : s_seg;
  TBE_SORT_STAMP(1);

  // ---- publish this segment's histogram row: the word is its own ready flag ------------------------
  uint32_t* my_row = a.table + static_cast<size_t>(b) * radix;
  for (int d = tid; d < radix; d += kSortThreads)
    __hip_atomic_store(&my_row[d], (epoch << kTagShift) | seg_hist[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- sum the rows of all predecessors (segments 0 .. b-1), two levels -----------------------------
  // Segments form groups of kSortGroup; the last segment of a group publishes the group's sum as a row
  // of the level-1 table as soon as it has read its group.  A segment then needs the level-1 rows of
  // the groups before its own and the level-0 rows of its own group before itself: <= 2 * kSortGroup
  // rows, all requested at once, two dependent hops on the critical path (not one per predecessor).
  {
    const int npairs = radix >> 1;
    const int nsplit = kSortThreads / npairs;
    const int j = tid % npairs;
    const int s = tid / npairs;
    int spins = 0;  // per thread over the whole wait: one give-up ends all its waiting
    auto sum_rows = [&](const uint32_t* tab32, uint32_t first, uint32_t count, uint32_t& acc0, uint32_t& acc1) {
      constexpr int U = kSortGroup;
      const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(tab32);
      for (uint32_t r0 = s; r0 < count; r0 += nsplit * U) {
        unsigned long long w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t r = r0 + u * nsplit;
          w[u] = 0ull;
          if (r < count) w[u] = __hip_atomic_load(tab + static_cast<size_t>(first + r) * npairs + j, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t r = r0 + u * nsplit;
          if (r < count) {
            while (static_cast<uint32_t>(w[u]) >> kTagShift != epoch || static_cast<uint32_t>(w[u] >> 32) >> kTagShift != epoch) {
              if (++spins > kSpinLimit) {
                if (spins == kSpinLimit + 1) report_sort_giveup(a.fault);
                break;
              }
              __builtin_amdgcn_s_sleep(2);
              w[u] = __hip_atomic_load(tab + static_cast<size_t>(first + r) * npairs + j, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            acc0 += static_cast<uint32_t>(w[u]) & kCountMask;
            acc1 += static_cast<uint32_t>(w[u] >> 32) & kCountMask;
          }
        }
      }
    };
    // sums the nsplit partial accumulators of digit pair `tid` through LDS (result valid for 2*tid < radix)
    auto reduce_split = [&](uint32_t acc0, uint32_t acc1, uint32_t& p0, uint32_t& p1) {
      if (nsplit == 1) {  // radix 1024: thread tid holds the whole sum of its pair already
        p0 = acc0;
        p1 = acc1;
        return;
      }
      s_part[s * radix + 2 * j] = acc0;
      s_part[s * radix + 2 * j + 1] = acc1;
      __syncthreads();
      p0 = p1 = 0;
      if (2 * tid < radix) {
        for (int q = 0; q < nsplit; ++q) {
          p0 += s_part[q * radix + 2 * tid];
          p1 += s_part[q * radix + 2 * tid + 1];
        }
      }
      __syncthreads();  // s_part is reused
    };
    const uint32_t grp = b / kSortGroup;
    const uint32_t in_grp = b % kSortGroup;
    uint32_t a0 = 0, a1 = 0, own0, own1, up0, up1;
    sum_rows(a.table, grp * kSortGroup, in_grp, a0, a1);
    reduce_split(a0, a1, own0, own1);
    TBE_SORT_STAMP(2);
    if (in_grp == kSortGroup - 1 && 2 * tid < radix) {  // block-uniform condition: this segment closes its group
      uint32_t* grow = a.group_table + static_cast<size_t>(grp) * radix;
      __hip_atomic_store(&grow[2 * tid], (epoch << kTagShift) | (own0 + seg_hist[2 * tid]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&grow[2 * tid + 1], (epoch << kTagShift) | (own1 + seg_hist[2 * tid + 1]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    a0 = a1 = 0;
    sum_rows(a.group_table, 0, grp, a0, a1);
    reduce_split(a0, a1, up0, up1);
    TBE_SORT_STAMP(3);
    // the next pass's digit totals: fire-and-forget integer atomics behind the waits, complete at the end of
    // this launch
    if (count_next) {
      for (int d = tid; d < radix; d += kSortThreads) {
        const uint32_t c = s_next[d];
        if (c != 0u) atomicAdd(&a.next_totals[d], c);
      }
    }
    // digit base (exclusive scan of this pass's totals) + predecessors
    const uint32_t base0 = block_excl_scan_pairs(t0, t1, s_scan);
    if (2 * tid < radix) {
      s_gbase[2 * tid] = base0 + own0 + up0;
      s_gbase[2 * tid + 1] = base0 + t0 + own1 + up1;
    }
    __syncthreads();
    TBE_SORT_STAMP(4);
  }

  // ---- place: tile by tile, LDS-staged so that the global writes are contiguous runs ----------------
  for (int t = 0; t < T; ++t) {
    const int64_t tile_begin = seg_begin + static_cast<int64_t>(t) * TILE;
    if (tile_begin >= seg_end) break;  // block-uniform
    if (T > 1) {
      load_tile(tile_begin, seg_end);
      for (int d = lane; d < radix; d += kWave) wave_cnt[wave][d] = 0;
      rank_tile(tile_begin);
    }
    {
      const uint32_t h0 = (2 * tid < radix) ? s_hist[2 * tid] : 0u;
      const uint32_t h1 = (2 * tid + 1 < radix) ? s_hist[2 * tid + 1] : 0u;
      const uint32_t l0 = block_excl_scan_pairs(h0, h1, s_scan);
      if (2 * tid < radix) {
        // wave_cnt[w][d] becomes the tile-local position of wave w's first key of digit d
        const uint32_t l1 = l0 + h0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
          wave_cnt[w][2 * tid] += l0;
          wave_cnt[w][2 * tid + 1] += l1;
        }
        s_goff[2 * tid] = s_gbase[2 * tid] - l0;
        s_goff[2 * tid + 1] = s_gbase[2 * tid + 1] - l1;
        s_gbase[2 * tid] += h0;
        s_gbase[2 * tid + 1] += h1;
      }
      __syncthreads();
    }
    const int64_t wbase = tile_begin + static_cast<int64_t>(wave) * (ROUNDS * kWave);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (wbase + r * kWave + lane < seg_end) {
        const unsigned digit = static_cast<unsigned>((k[r] >> a.shift) & dmask);
        const uint32_t pos = wave_cnt[wave][digit] + wrank[r];
        stage_k[pos] = k[r];
        stage_v[pos] = v[r];
      }
    }
    __syncthreads();
    TBE_SORT_STAMP(5);
    const int count = static_cast<int>(min<int64_t>(TILE, seg_end - tile_begin));
    for (int i = tid; i < count; i += kSortThreads) {
      const KeyT key = stage_k[i];
      const uint32_t pos = s_goff[static_cast<unsigned>((key >> a.shift) & dmask)] + static_cast<uint32_t>(i);
      if (pos < a.N) {  // always true unless a wait gave up (garbage prefix, reported): never write out of bounds
        a.keys_out[pos] = key;
        a.vals_out[pos] = stage_v[i];
      }
    }
    __syncthreads();
    TBE_SORT_STAMP(6);
  }
}

// Sorts N pairs on the low `key_bits` bits.  Buffers ping-pong between (k0, v0) and (k1, v1);
// returns 0 if the result is in (k0, v0), 1 if in (k1, v1), negative TBE_ERR_* on failure.
template <typename KeyT, typename ValT>
inline int radix_sort_pairs(KeyT* k0, KeyT* k1, ValT* v0, ValT* v1, int64_t N, int key_bits, const RadixWorkspace& ws,
                            hipStream_t st, int flags = 0) {
  if (N >= kSortMaxPairs) {
    set_error("radix_sort_pairs: N must be < 2^29");
    return TBE_ERR_UNSUPPORTED;
  }
  const RadixPlan pl = radix_plan(key_bits);
  if (pl.passes > kMaxPasses) {
    set_error("radix_sort_pairs: key_bits too large");
    return TBE_ERR_UNSUPPORTED;
  }
  if (N <= 0) return pl.passes & 1;
  uint32_t* const fault = fault_word_device();
  if (fault == nullptr) {  // a give-up could not be reported: refuse to sort rather than risk a silent wrong order
    set_error("radix_sort_pairs: the fault word (pinned host memory) could not be allocated");
    return TBE_ERR_LAUNCH;
  }
  const int radix = 1 << pl.bits;
  const RadixGeom g = radix_geom(N, sizeof(KeyT) + sizeof(ValT));
  const size_t state_words = radix_state_words(N, key_bits, sizeof(KeyT) + sizeof(ValT));
  if (!(flags & kSortStateZeroed) && hipMemsetAsync(ws.state, 0, state_words * sizeof(uint32_t), st) != hipSuccess) {
    set_error("radix_sort_pairs: hipMemsetAsync failed");
    return TBE_ERR_LAUNCH;
  }
  const unsigned tgrid = static_cast<unsigned>(std::min<int64_t>(kSortMaxBlocks, (N + 4 * kSortThreads - 1) / (4 * kSortThreads)));
  // digit totals of pass 0 only: every pass kernel accumulates the totals of the pass after it
  hipLaunchKernelGGL((radix_totals_kernel<KeyT>), dim3(tgrid), dim3(kSortThreads), 0, st, k0, N, 1, pl.bits, key_bits, ws.totals);
  int cur = 0;
  for (int p = 0; p < pl.passes; ++p) {
    RadixPassArgs<KeyT, ValT> a;
    a.keys_in = cur ? k1 : k0;
    a.vals_in = cur ? v1 : v0;
    a.keys_out = cur ? k0 : k1;
    a.vals_out = cur ? v0 : v1;
    a.N = N;
    a.shift = pl.bits * p;
    a.bits = pl.bits;
    a.width = std::min(pl.bits, key_bits - pl.bits * p);
    a.next_width = p + 1 < pl.passes ? std::min(pl.bits, key_bits - pl.bits * (p + 1)) : 0;
    a.next_totals = p + 1 < pl.passes ? ws.totals + static_cast<size_t>(radix) * (p + 1) : nullptr;
    a.pass = p;
    a.tiles_per_block = g.tiles_per_block;
    a.table = ws.table;
    a.group_table = ws.group_table;
    a.totals = ws.totals + static_cast<size_t>(radix) * p;
    a.ticket = ws.tickets + p;
    a.stamps = g_sort_stamps;
    a.fault = fault;
    const dim3 grid(static_cast<unsigned>(g.blocks)), block(kSortThreads);
    bool launched = true;
#define TBE_SORT_PASS(R)                                                                  \
  case R:                                                                                 \
    hipLaunchKernelGGL((radix_pass_kernel<KeyT, ValT, R>), grid, block, 0, st, a);        \
    break;
    if constexpr (sizeof(KeyT) + sizeof(ValT) <= 8) {
      switch (g.rounds) {
        TBE_SORT_PASS(4) TBE_SORT_PASS(6) TBE_SORT_PASS(8) TBE_SORT_PASS(10) TBE_SORT_PASS(12) TBE_SORT_PASS(14)
        TBE_SORT_PASS(16)
        default: launched = false;
      }
    } else {
      switch (g.rounds) {
        TBE_SORT_PASS(4) TBE_SORT_PASS(6) TBE_SORT_PASS(8)
        default: launched = false;
      }
    }
#undef TBE_SORT_PASS
    if (!launched) {
      set_error("radix_sort_pairs: internal geometry error");
      return TBE_ERR_UNSUPPORTED;
    }
    cur ^= 1;
  }
  if (hipGetLastError() != hipSuccess) {
    set_error("radix_sort_pairs: launch failed");
    return TBE_ERR_LAUNCH;
  }
  return cur;
}

}  // namespace tbe

// Hand-written stable LSD radix sort of (key, 64-bit payload) pairs for gfx950, used by the TBE
// backward to group a batch's contributions by table row (tbe_backward.hip).
//
// Digits of up to 10 bits: passes = ceil(key_bits / 10), digit width = ceil(key_bits / passes)
// (28-bit Criteo row keys: 3 passes of 10 bits).  3 launches per pass + 2 up front, no
// inter-workgroup synchronisation inside a launch (nothing to dead-lock, nothing stale):
//   0. radix_totals_kernel : once per sort: digit totals of EVERY pass (they do not depend on the order
//                            of the keys), per-workgroup partial counts, no global atomics;
//      radix_totals_reduce : sums the partials.
//   1. radix_hist_kernel   : per-tile histogram (LDS atomics) -> hist[digit][tile].
//   2. radix_offsets_kernel: one workgroup per digit: digit base = sum of lower digits' totals, then
//                            an exclusive scan of that digit's row over the tiles (in place).
//   3. radix_scatter_kernel: a tile = 4 waves x R rounds x 64 keys in input order.  Rank inside a
//                            round comes from a match-any built of one wave ballot per digit bit;
//                            per-wave digit counters in LDS carry the running offset between
//                            rounds; a cross-wave prefix per digit orders the waves.  Equal keys
//                            therefore keep their input order (stable), which is what makes the
//                            backward's summation order a function of the input only.
#pragma once
#include "common.hpp"

namespace tbe {

constexpr int kSortThreads = 256;
constexpr int kSortRounds = 8;                                   // rounds of 64 keys per wave
constexpr int kSortTile = kSortThreads * kSortRounds;            // keys per workgroup
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kTotalsBlocks = 256;
constexpr int kMaxDigitBits = 10;
constexpr int kMaxRadix = 1 << kMaxDigitBits;
constexpr int kMaxPasses = 7;

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

struct RadixWorkspace {
  uint32_t* hist;         // [radix][ntiles]
  uint32_t* totals_part;  // [kTotalsBlocks][passes][radix] per-workgroup partial digit totals
  uint32_t* totals;       // [passes][radix]
  size_t bytes;
};

inline int radix_passes(int key_bits) { return radix_plan(key_bits).passes; }
inline int64_t radix_tiles(int64_t N) { return (N + kSortTile - 1) / kSortTile; }

inline RadixWorkspace radix_carve(void* base, int64_t N, int key_bits) {
  const RadixPlan pl = radix_plan(key_bits);
  const size_t radix = static_cast<size_t>(1) << pl.bits;
  Carver c(base);
  RadixWorkspace w;
  w.hist = c.take<uint32_t>(radix * static_cast<size_t>(radix_tiles(N)));
  w.totals_part = c.take<uint32_t>(static_cast<size_t>(kTotalsBlocks) * radix * pl.passes);
  w.totals = c.take<uint32_t>(radix * pl.passes);
  w.bytes = c.total();
  return w;
}

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const KeyT* __restrict__ keys, int64_t N, int shift,
                                                                 int bits, uint32_t* __restrict__ hist, int64_t ntiles) {
  __shared__ uint32_t h[kMaxRadix];
  const int radix = 1 << bits;
  for (int i = threadIdx.x; i < radix; i += kSortThreads) h[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t i = base + r * kSortThreads + threadIdx.x;
    if (i < N) atomicAdd(&h[(keys[i] >> shift) & (radix - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < radix; i += kSortThreads) hist[static_cast<int64_t>(i) * ntiles + blockIdx.x] = h[i];
}

// Digit totals of all passes, once per sort (independent of key order).  grid = kTotalsBlocks.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_totals_kernel(const KeyT* __restrict__ keys, int64_t N, int passes,
                                                                   int bits, uint32_t* __restrict__ totals_part) {
  __shared__ uint32_t h[kMaxPasses * kMaxRadix];
  const int radix = 1 << bits;
  for (int i = threadIdx.x; i < passes * radix; i += kSortThreads) h[i] = 0;
  __syncthreads();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSortThreads + threadIdx.x; i < N;
       i += static_cast<int64_t>(gridDim.x) * kSortThreads) {
    const KeyT k = keys[i];
    for (int p = 0; p < passes; ++p) atomicAdd(&h[p * radix + static_cast<int>((k >> (bits * p)) & (radix - 1))], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < passes * radix; i += kSortThreads)
    totals_part[static_cast<int64_t>(blockIdx.x) * passes * radix + i] = h[i];
}

// grid = passes * radix workgroups: workgroup (p, d) sums the kTotalsBlocks partial counts of digit d.
static __global__ __launch_bounds__(kSortThreads) void radix_totals_reduce_kernel(const uint32_t* __restrict__ totals_part,
                                                                          int row, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_tot[kSortWaves];
  const int pd = blockIdx.x;  // p * radix + d
  uint32_t s = threadIdx.x < kTotalsBlocks ? totals_part[static_cast<int64_t>(threadIdx.x) * row + pd] : 0u;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) totals[pd] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// grid = radix workgroups (one per digit)
static __global__ __launch_bounds__(kSortThreads) void radix_offsets_kernel(uint32_t* __restrict__ hist,
                                                                    const uint32_t* __restrict__ totals, int64_t ntiles) {
  __shared__ uint32_t wave_tot[kSortWaves];
  __shared__ uint32_t digit_base;
  const int d = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // digit base = sum of totals of lower digits
  uint32_t s = 0u;
  for (int i = threadIdx.x; i < d; i += kSortThreads) s += totals[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (lane == 0) wave_tot[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) digit_base = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  __syncthreads();
  uint32_t carry = digit_base;
  uint32_t* row = hist + static_cast<int64_t>(d) * ntiles;
  for (int64_t base = 0; base < ntiles; base += kSortThreads) {
    const int64_t i = base + threadIdx.x;
    const uint32_t x = i < ntiles ? row[i] : 0u;
    uint32_t inc = x;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = __shfl_up(inc, o, kWave);
      if (lane >= o) inc += y;
    }
    __syncthreads();  // wave_tot reuse
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const uint32_t t = wave_tot[w];
      if (w < wave) add += t;
      tot += t;
    }
    if (i < ntiles) row[i] = carry + add + inc - x;
    carry += tot;
  }
}

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const KeyT* __restrict__ keys_in,
                                                                    const uint64_t* __restrict__ vals_in,
                                                                    KeyT* __restrict__ keys_out,
                                                                    uint64_t* __restrict__ vals_out, int64_t N, int shift,
                                                                    int bits, const uint32_t* __restrict__ hist,
                                                                    int64_t ntiles) {
  __shared__ uint32_t wave_cnt[kSortWaves][kMaxRadix];
  const int radix = 1 << bits;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t wbase = static_cast<int64_t>(blockIdx.x) * kSortTile + static_cast<int64_t>(wave) * (kSortRounds * kWave);
  for (int d = lane; d < radix; d += kWave) wave_cnt[wave][d] = 0;
  KeyT k[kSortRounds];
  uint64_t v[kSortRounds];
  unsigned long long peers[kSortRounds];
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t i = wbase + r * kWave + lane;
    const bool valid = i < N;
    k[r] = valid ? keys_in[i] : static_cast<KeyT>(0);
    v[r] = valid ? vals_in[i] : 0ull;
  }
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const bool valid = wbase + r * kWave + lane < N;
    const unsigned digit = static_cast<unsigned>((k[r] >> shift) & (radix - 1));
    unsigned long long p = __ballot(valid);
    for (int bit = 0; bit < bits; ++bit) {  // match-any: lanes holding the same digit
      const bool one = (digit >> bit) & 1u;
      const unsigned long long b = __ballot(one);
      p &= one ? b : ~b;
    }
    peers[r] = valid ? p : 0ull;
    // leader (lowest lane of the peer group) accumulates the group's size; digits differ between leaders
    if (valid && (p & lt_mask) == 0ull) wave_cnt[wave][digit] += static_cast<uint32_t>(__popcll(p));
  }
  __syncthreads();
  // cross-wave exclusive prefix per digit on top of this tile's global base for the digit
  for (int d = threadIdx.x; d < radix; d += kSortThreads) {
    uint32_t off = hist[static_cast<int64_t>(d) * ntiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const uint32_t c = wave_cnt[w][d];
      wave_cnt[w][d] = off;
      off += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const unsigned long long p = peers[r];
    const unsigned digit = static_cast<unsigned>((k[r] >> shift) & (radix - 1));
    if (p != 0ull) {
      const uint32_t pos = wave_cnt[wave][digit] + static_cast<uint32_t>(__popcll(p & lt_mask));
      keys_out[pos] = k[r];
      vals_out[pos] = v[r];
    }
    // every lane has read wave_cnt for this round before any leader bumps it (same wave: LDS ops of one
    // instruction stream complete in order; the fences keep the compiler from reordering them)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (p != 0ull && (p & lt_mask) == 0ull) wave_cnt[wave][digit] += static_cast<uint32_t>(__popcll(p));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Sorts N pairs on the low `key_bits` bits.  Buffers ping-pong between (k0, v0) and (k1, v1);
// returns 0 if the result is in (k0, v0), 1 if in (k1, v1), negative TBE_ERR_* on failure.
template <typename KeyT>
inline int radix_sort_pairs(KeyT* k0, KeyT* k1, uint64_t* v0, uint64_t* v1, int64_t N, int key_bits,
                            const RadixWorkspace& ws, hipStream_t st) {
  if (N >= (1ll << 32)) {
    set_error("radix_sort_pairs: N must be < 2^32");
    return TBE_ERR_UNSUPPORTED;
  }
  const RadixPlan pl = radix_plan(key_bits);
  if (pl.passes > kMaxPasses) {
    set_error("radix_sort_pairs: key_bits too large");
    return TBE_ERR_UNSUPPORTED;
  }
  const int radix = 1 << pl.bits;
  const int64_t ntiles = radix_tiles(N);
  hipLaunchKernelGGL((radix_totals_kernel<KeyT>), dim3(kTotalsBlocks), dim3(kSortThreads), 0, st, k0, N, pl.passes, pl.bits,
                     ws.totals_part);
  hipLaunchKernelGGL(radix_totals_reduce_kernel, dim3(pl.passes * radix), dim3(kSortThreads), 0, st, ws.totals_part,
                     pl.passes * radix, ws.totals);
  int cur = 0;
  for (int p = 0; p < pl.passes; ++p) {
    const KeyT* kin = cur ? k1 : k0;
    const uint64_t* vin = cur ? v1 : v0;
    KeyT* kout = cur ? k0 : k1;
    uint64_t* vout = cur ? v0 : v1;
    hipLaunchKernelGGL((radix_hist_kernel<KeyT>), dim3(static_cast<unsigned>(ntiles)), dim3(kSortThreads), 0, st, kin, N,
                       pl.bits * p, pl.bits, ws.hist, ntiles);
    hipLaunchKernelGGL(radix_offsets_kernel, dim3(radix), dim3(kSortThreads), 0, st, ws.hist, ws.totals + radix * p, ntiles);
    hipLaunchKernelGGL((radix_scatter_kernel<KeyT>), dim3(static_cast<unsigned>(ntiles)), dim3(kSortThreads), 0, st, kin,
                       vin, kout, vout, N, pl.bits * p, pl.bits, ws.hist, ntiles);
    cur ^= 1;
  }
  if (hipGetLastError() != hipSuccess) {
    set_error("radix_sort_pairs: launch failed");
    return TBE_ERR_LAUNCH;
  }
  return cur;
}

}  // namespace tbe

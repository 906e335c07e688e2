// Hand-written stable LSD radix sort of (key, 64-bit payload) pairs for gfx950, used by the TBE
// backward to group a batch's contributions by table row (tbe_backward.hip).
//
// 8-bit digits, ceil(key_bits / 8) passes, 3 launches per pass, no inter-workgroup
// synchronisation inside a launch (nothing to dead-lock, nothing stale):
//   0. radix_totals_kernel : once per sort: digit totals of EVERY pass (they do not depend on the order
//                            of the keys), per-workgroup partial counts, no global atomics.
//   1. radix_hist_kernel   : per-tile 256-bin histogram (LDS atomics) -> hist[digit][tile].
//   2. radix_offsets_kernel: one workgroup per digit: digit base = sum of lower digits' totals, then
//                            an exclusive scan of that digit's row over the tiles (in place).
//   3. radix_scatter_kernel: a tile = 4 waves x 16 rounds x 64 keys in input order.  Rank inside a
//                            round comes from a match-any built of 8 wave ballots (one per digit
//                            bit); per-wave digit counters in LDS carry the running offset between
//                            rounds; a cross-wave prefix per digit orders the waves.  Equal keys
//                            therefore keep their input order (stable), which is what makes the
//                            backward's summation order a function of the input only.
#pragma once
#include "common.hpp"

namespace tbe {

constexpr int kSortThreads = 256;
constexpr int kSortRounds = 8;                                   // rounds of 64 keys per wave
constexpr int kSortTile = kSortThreads * kSortRounds;            // 4096 keys per workgroup
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kTotalsBlocks = 256;
constexpr int kMaxPasses = 8;

struct RadixWorkspace {
  uint32_t* hist;    // [256][ntiles]
  uint32_t* totals_part;  // [kTotalsBlocks][passes][256] per-workgroup partial digit totals
  uint32_t* totals;       // [passes][256]
  size_t bytes;
};

inline int radix_passes(int key_bits) { return (key_bits + 7) / 8; }
inline int64_t radix_tiles(int64_t N) { return (N + kSortTile - 1) / kSortTile; }

inline RadixWorkspace radix_carve(void* base, int64_t N, int key_bits) {
  Carver c(base);
  RadixWorkspace w;
  w.hist = c.take<uint32_t>(256 * static_cast<size_t>(radix_tiles(N)));
  w.totals_part = c.take<uint32_t>(static_cast<size_t>(kTotalsBlocks) * 256 * radix_passes(key_bits));
  w.totals = c.take<uint32_t>(static_cast<size_t>(256) * radix_passes(key_bits));
  w.bytes = c.total();
  return w;
}

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const KeyT* __restrict__ keys, int64_t N, int shift,
                                                                 uint32_t* __restrict__ hist, int64_t ntiles) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t i = base + r * kSortThreads + threadIdx.x;
    if (i < N) atomicAdd(&h[(keys[i] >> shift) & 255], 1u);
  }
  __syncthreads();
  hist[static_cast<int64_t>(threadIdx.x) * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Digit totals of all passes, once per sort (independent of key order).  grid = kTotalsBlocks.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void radix_totals_kernel(const KeyT* __restrict__ keys, int64_t N, int passes,
                                                                   uint32_t* __restrict__ totals_part) {
  __shared__ uint32_t h[kMaxPasses * 256];
  for (int i = threadIdx.x; i < passes * 256; i += kSortThreads) h[i] = 0;
  __syncthreads();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSortThreads + threadIdx.x; i < N;
       i += static_cast<int64_t>(gridDim.x) * kSortThreads) {
    const KeyT k = keys[i];
    for (int p = 0; p < passes; ++p) atomicAdd(&h[p * 256 + static_cast<int>((k >> (8 * p)) & 255)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < passes * 256; i += kSortThreads)
    totals_part[static_cast<int64_t>(blockIdx.x) * passes * 256 + i] = h[i];
}

// grid = passes * 256 workgroups: workgroup (p, d) sums the kTotalsBlocks partial counts of digit d.
__global__ __launch_bounds__(kSortThreads) void radix_totals_reduce_kernel(const uint32_t* __restrict__ totals_part,
                                                                          int passes, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_tot[kSortWaves];
  const int pd = blockIdx.x;  // p * 256 + d
  uint32_t s = threadIdx.x < kTotalsBlocks ? totals_part[static_cast<int64_t>(threadIdx.x) * passes * 256 + pd] : 0u;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) totals[pd] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// grid = 256 workgroups (one per digit)
__global__ __launch_bounds__(kSortThreads) void radix_offsets_kernel(uint32_t* __restrict__ hist,
                                                                    const uint32_t* __restrict__ totals, int64_t ntiles) {
  __shared__ uint32_t wave_tot[kSortWaves];
  __shared__ uint32_t digit_base;
  const int d = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // digit base = sum of totals of lower digits
  uint32_t s = static_cast<int>(threadIdx.x) < d ? totals[threadIdx.x] : 0u;
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
                                                                    const uint32_t* __restrict__ hist, int64_t ntiles) {
  __shared__ uint32_t wave_cnt[kSortWaves][256];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t wbase = static_cast<int64_t>(blockIdx.x) * kSortTile + static_cast<int64_t>(wave) * (kSortRounds * kWave);
  for (int d = lane; d < 256; d += kWave) wave_cnt[wave][d] = 0;
  KeyT k[kSortRounds];
  uint64_t v[kSortRounds];
  unsigned long long peers[kSortRounds];
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t i = wbase + r * kWave + lane;
    const bool valid = i < N;
    k[r] = valid ? keys_in[i] : static_cast<KeyT>(0);
    v[r] = valid ? vals_in[i] : 0ull;
    const unsigned digit = static_cast<unsigned>((k[r] >> shift) & 255);
    unsigned long long p = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (digit >> bit) & 1u;
      const unsigned long long b = __ballot(one);
      p &= one ? b : ~b;
    }
    peers[r] = valid ? p : 0ull;
    // leader (lowest lane of the peer group) accumulates the group's size; digits differ between leaders
    if (valid && (p & lt_mask) == 0ull) wave_cnt[wave][digit] += static_cast<uint32_t>(__popcll(p));
  }
  __syncthreads();
  {
    // cross-wave exclusive prefix per digit on top of this tile's global base for the digit
    const int d = threadIdx.x;
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
    if (p != 0ull) {
      const unsigned digit = static_cast<unsigned>((k[r] >> shift) & 255);
      const uint32_t pos = wave_cnt[wave][digit] + static_cast<uint32_t>(__popcll(p & lt_mask));
      keys_out[pos] = k[r];
      vals_out[pos] = v[r];
    }
    // every lane has read wave_cnt for this round before any leader bumps it (same wave: LDS ops of one
    // instruction stream complete in order; the fence keeps the compiler from reordering them)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (p != 0ull && (p & lt_mask) == 0ull) {
      const unsigned digit = static_cast<unsigned>((k[r] >> shift) & 255);
      wave_cnt[wave][digit] += static_cast<uint32_t>(__popcll(p));
    }
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
  const int passes = radix_passes(key_bits);
  const int64_t ntiles = radix_tiles(N);
  if (passes > kMaxPasses) {
    set_error("radix_sort_pairs: key_bits too large");
    return TBE_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((radix_totals_kernel<KeyT>), dim3(kTotalsBlocks), dim3(kSortThreads), 0, st, k0, N, passes, ws.totals_part);
  hipLaunchKernelGGL(radix_totals_reduce_kernel, dim3(passes * 256), dim3(kSortThreads), 0, st, ws.totals_part, passes, ws.totals);
  int cur = 0;
  for (int p = 0; p < passes; ++p) {
    const KeyT* kin = cur ? k1 : k0;
    const uint64_t* vin = cur ? v1 : v0;
    KeyT* kout = cur ? k0 : k1;
    uint64_t* vout = cur ? v0 : v1;
    hipLaunchKernelGGL((radix_hist_kernel<KeyT>), dim3(static_cast<unsigned>(ntiles)), dim3(kSortThreads), 0, st, kin, N,
                       8 * p, ws.hist, ntiles);
    hipLaunchKernelGGL(radix_offsets_kernel, dim3(256), dim3(kSortThreads), 0, st, ws.hist, ws.totals + 256 * p, ntiles);
    hipLaunchKernelGGL((radix_scatter_kernel<KeyT>), dim3(static_cast<unsigned>(ntiles)), dim3(kSortThreads), 0, st, kin,
                       vin, kout, vout, N, 8 * p, ws.hist, ntiles);
    cur ^= 1;
  }
  if (hipGetLastError() != hipSuccess) {
    set_error("radix_sort_pairs: launch failed");
    return TBE_ERR_LAUNCH;
  }
  return cur;
}

}  // namespace tbe

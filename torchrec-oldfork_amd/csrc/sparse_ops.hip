// KeyedJaggedTensor index ops for gfx950: cumsum, permute_2D, block-bucketize, and the
// pooled all-to-all layout copies.  All integer/byte work — results are bit-exact with the
// oracle.  Reference call sites are cited in include/tbe_hip.h.
#include <algorithm>

#include "common.hpp"

namespace tbe {

// ---------------------------------------------------------------------------------------
// Scan.  Three launches: per-tile sums, one-block scan of the tile sums, per-tile scan.
// A tile is kScanRounds rounds of 256 coalesced elements; accumulation is 64-bit and the
// result is truncated to TOut (== two's-complement wrap of the narrower type).
// Optional row gather: element i reads in[perm[i / B] * B + i % B] (permute_2D lengths).
// ---------------------------------------------------------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanRounds = 8;
constexpr int kScanTile = kScanBlock * kScanRounds;

template <typename TIn>
__device__ __forceinline__ int64_t scan_load(const TIn* in, int64_t i, int64_t n, const int32_t* perm, int B) {
  if (i >= n) return 0;
  if (perm != nullptr) {
    const int64_t t = i / B;
    const int64_t b = i - t * B;
    return static_cast<int64_t>(in[static_cast<int64_t>(perm[t]) * B + b]);
  }
  return static_cast<int64_t>(in[i]);
}

// Block-wide inclusive scan of one value per thread (256 threads = 4 waves).
__device__ __forceinline__ int64_t block_inclusive_scan(int64_t x, int64_t* wave_tot, int64_t* total) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int64_t y = shfl64(x, (lane - o) & 63);
    if (lane >= o) x += y;
  }
  if (lane == 63) wave_tot[wave] = x;
  __syncthreads();
  int64_t add = 0;
  int64_t tot = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / kWave; ++w) {
    const int64_t t = wave_tot[w];
    if (w < wave) add += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return x + add;
}

template <typename TIn>
__global__ __launch_bounds__(kScanBlock) void scan_tile_sums_kernel(const TIn* __restrict__ in, int64_t n,
                                                                   const int32_t* __restrict__ perm, int B,
                                                                   int64_t* __restrict__ tile_sums) {
  __shared__ int64_t wave_tot[kScanBlock / kWave];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  int64_t s = 0;
#pragma unroll
  for (int r = 0; r < kScanRounds; ++r) s += scan_load(in, base + r * kScanBlock + threadIdx.x, n, perm, B);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += shfl64(s, (threadIdx.x & 63) ^ o);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t t = 0;
    for (int w = 0; w < kScanBlock / kWave; ++w) t += wave_tot[w];
    tile_sums[blockIdx.x] = t;
  }
}

// Exclusive scan of tile sums, in place, by ONE block (carry across rounds).
__global__ __launch_bounds__(kScanBlock) void scan_tile_offsets_kernel(int64_t* __restrict__ tile_sums, int64_t ntiles) {
  __shared__ int64_t wave_tot[kScanBlock / kWave];
  int64_t carry = 0;
  for (int64_t base = 0; base < ntiles; base += kScanBlock) {
    const int64_t i = base + threadIdx.x;
    const int64_t x = i < ntiles ? tile_sums[i] : 0;
    int64_t total;
    const int64_t inc = block_inclusive_scan(x, wave_tot, &total);
    if (i < ntiles) tile_sums[i] = carry + inc - x;
    carry += total;
  }
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kScanBlock) void scan_final_kernel(const TIn* __restrict__ in, TOut* __restrict__ out,
                                                               int64_t n, const int32_t* __restrict__ perm, int B,
                                                               const int64_t* __restrict__ tile_offsets, int mode,
                                                               TIn* __restrict__ gathered) {
  __shared__ int64_t wave_tot[kScanBlock / kWave];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  int64_t carry = tile_offsets != nullptr ? tile_offsets[blockIdx.x] : 0;
  if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = static_cast<TOut>(0);
#pragma unroll 1
  for (int r = 0; r < kScanRounds; ++r) {
    const int64_t i = base + r * kScanBlock + threadIdx.x;
    if (base + r * kScanBlock >= n) break;  // block-uniform
    const int64_t x = scan_load(in, i, n, perm, B);
    int64_t total;
    const int64_t inc = block_inclusive_scan(x, wave_tot, &total);
    if (i < n) {
      if (mode == 0) out[i + 1] = static_cast<TOut>(carry + inc);
      else if (mode == 1) out[i] = static_cast<TOut>(carry + inc);
      else out[i] = static_cast<TOut>(carry + inc - x);
      if (gathered != nullptr) gathered[i] = static_cast<TIn>(x);
    }
    carry += total;
  }
}

template <typename TIn, typename TOut>
static int run_scan(const TIn* in, TOut* out, int64_t n, const int32_t* perm, int B, int mode, int64_t* tile_ws,
                    TIn* gathered, hipStream_t st) {
  if (n <= 0) {
    if (mode == 0) (void)hipMemsetAsync(out, 0, sizeof(TOut), st);
    return TBE_OK;
  }
  const int64_t ntiles = (n + kScanTile - 1) / kScanTile;
  if (ntiles > 1) {
    hipLaunchKernelGGL((scan_tile_sums_kernel<TIn>), dim3(static_cast<unsigned>(ntiles)), dim3(kScanBlock), 0, st, in, n, perm, B, tile_ws);
    hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(kScanBlock), 0, st, tile_ws, ntiles);
  }
  hipLaunchKernelGGL((scan_final_kernel<TIn, TOut>), dim3(static_cast<unsigned>(ntiles)), dim3(kScanBlock), 0, st, in, out, n, perm,
                     B, ntiles > 1 ? tile_ws : nullptr, mode, gathered);
  TBE_CHECK_LAUNCH("scan");
  return TBE_OK;
}

static size_t scan_ws_bytes(int64_t n) {
  const int64_t ntiles = (std::max<int64_t>(n, 1) + kScanTile - 1) / kScanTile;
  return align_up(static_cast<size_t>(ntiles) * sizeof(int64_t), 256);
}

// ---------------------------------------------------------------------------------------
// permute_2D data copy: one wave per 64 consecutive OUTPUT segments; the wave's output
// range is contiguous, lanes stride over it (coalesced stores) and find their segment by
// a 6-step cross-lane binary search over the 64 segment starts.
// ---------------------------------------------------------------------------------------
template <typename V, typename Wt>
__global__ __launch_bounds__(256) void permute_2d_data_kernel(const int32_t* __restrict__ perm, int T_out, int B,
                                                             const int64_t* __restrict__ in_offsets,
                                                             const int64_t* __restrict__ out_offsets,
                                                             const V* __restrict__ values, V* __restrict__ out_values,
                                                             const Wt* __restrict__ weights, Wt* __restrict__ out_weights) {
  const int lane = threadIdx.x & 63;
  const int64_t nseg = static_cast<int64_t>(T_out) * B;
  const int64_t s0 = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * kWave;
  if (s0 >= nseg) return;
  const int64_t s = s0 + lane;
  int64_t ostart_l = 0, istart_l = 0;
  const int64_t s_end = min(s0 + static_cast<int64_t>(kWave), nseg);
  const int64_t range_begin = out_offsets[s0];
  const int64_t range_end = out_offsets[s_end];
  if (s < nseg) {
    ostart_l = out_offsets[s];
    const int64_t t = s / B;
    const int64_t b = s - t * B;
    istart_l = in_offsets[static_cast<int64_t>(perm[t]) * B + b];
  } else {
    ostart_l = range_end;  // sentinel: never selected by the search below
  }
  for (int64_t eb = range_begin; eb < range_end; eb += kWave) {
    const int64_t e = eb + lane;
    // largest segment k in [0, 64) with ostart[k] <= e
    int k = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const int cand = k + step;
      const int64_t v = shfl64(ostart_l, cand & 63);
      if (cand < kWave && v <= e) k = cand;
    }
    const int64_t os = shfl64(ostart_l, k);
    const int64_t is = shfl64(istart_l, k);
    if (e < range_end) {
      const int64_t src = is + (e - os);
      out_values[e] = values[src];
      if (weights != nullptr) out_weights[e] = weights[src];
    }
  }
}

template <typename LenT>
__global__ __launch_bounds__(256) void permute_lengths_kernel(const int32_t* __restrict__ perm, int T_out, int B,
                                                             const LenT* __restrict__ lengths, LenT* __restrict__ out) {
  const int64_t n = static_cast<int64_t>(T_out) * B;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t t = i / B;
    const int64_t b = i - t * B;
    out[i] = lengths[static_cast<int64_t>(perm[t]) * B + b];
  }
}

// ---------------------------------------------------------------------------------------
// block_bucketize_sparse_features.  Thread per bag for count and scatter (a bag's entries in
// new_lengths / cursor are touched by that thread only => no atomics, stable order).
// ---------------------------------------------------------------------------------------
template <typename LenT, typename IdxT>
__global__ __launch_bounds__(256) void bucketize_count_kernel(const int64_t* __restrict__ offsets, int64_t lengths_size,
                                                             int B, const IdxT* __restrict__ indices,
                                                             const IdxT* __restrict__ block_sizes, int my_size,
                                                             LenT* __restrict__ new_lengths) {
  const int64_t bag = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bag >= lengths_size) return;
  const int f = static_cast<int>(bag / B);
  const uint64_t blk = static_cast<uint64_t>(block_sizes[f]);
  const int64_t s = offsets[bag], e = offsets[bag + 1];
  for (int64_t i = s; i < e; ++i) {
    const uint64_t idx = static_cast<uint64_t>(indices[i]);
    const uint64_t p = idx / blk;
    if (p < static_cast<uint64_t>(my_size)) new_lengths[p * lengths_size + bag] += 1;
  }
}

template <typename LenT, typename IdxT>
__global__ __launch_bounds__(256) void bucketize_scatter_kernel(
    const int64_t* __restrict__ offsets, int64_t lengths_size, int B, const IdxT* __restrict__ indices,
    const IdxT* __restrict__ block_sizes, int my_size, const float* __restrict__ weights,
    const int64_t* __restrict__ new_offsets, int32_t* __restrict__ cursor, IdxT* __restrict__ new_indices,
    float* __restrict__ new_weights, IdxT* __restrict__ new_pos, IdxT* __restrict__ unbucketize_permute) {
  const int64_t bag = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bag >= lengths_size) return;
  const int f = static_cast<int>(bag / B);
  const uint64_t blk = static_cast<uint64_t>(block_sizes[f]);
  const int64_t s = offsets[bag], e = offsets[bag + 1];
  for (int64_t i = s; i < e; ++i) {
    const uint64_t idx = static_cast<uint64_t>(indices[i]);
    const uint64_t p = idx / blk;
    if (p >= static_cast<uint64_t>(my_size)) {
      if (unbucketize_permute != nullptr) unbucketize_permute[i] = static_cast<IdxT>(-1);
      continue;
    }
    const int64_t slot = static_cast<int64_t>(p) * lengths_size + bag;
    const int c = cursor[slot];
    cursor[slot] = c + 1;
    const int64_t dst = new_offsets[slot] + c;
    new_indices[dst] = static_cast<IdxT>(idx - p * blk);
    if (weights != nullptr) new_weights[dst] = weights[i];
    if (new_pos != nullptr) new_pos[dst] = static_cast<IdxT>(i - s);
    if (unbucketize_permute != nullptr) unbucketize_permute[i] = static_cast<IdxT>(dst);
  }
}

// Same two steps with the per-bag bucket counters in LDS (my_size <= 32): layout [bucket][thread]
// puts a thread's counters in one bank column, so the increments never conflict.
constexpr int kBucketizeLdsMaxBuckets = 32;

template <typename LenT, typename IdxT>
__global__ __launch_bounds__(256) void bucketize_count_lds_kernel(const int64_t* __restrict__ offsets,
                                                                 int64_t lengths_size, int B,
                                                                 const IdxT* __restrict__ indices,
                                                                 const IdxT* __restrict__ block_sizes, int my_size,
                                                                 LenT* __restrict__ new_lengths) {
  extern __shared__ int32_t cnt[];  // [my_size][256]
  for (int p = 0; p < my_size; ++p) cnt[p * 256 + threadIdx.x] = 0;
  const int64_t bag = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bag >= lengths_size) return;
  const int f = static_cast<int>(bag / B);
  const uint64_t blk = static_cast<uint64_t>(block_sizes[f]);
  const int64_t s = offsets[bag], e = offsets[bag + 1];
  for (int64_t i = s; i < e; ++i) {
    const uint64_t p = static_cast<uint64_t>(indices[i]) / blk;
    if (p < static_cast<uint64_t>(my_size)) cnt[p * 256 + threadIdx.x] += 1;
  }
  for (int p = 0; p < my_size; ++p)
    new_lengths[static_cast<int64_t>(p) * lengths_size + bag] = static_cast<LenT>(cnt[p * 256 + threadIdx.x]);
}

template <typename LenT, typename IdxT>
__global__ __launch_bounds__(256) void bucketize_scatter_lds_kernel(
    const int64_t* __restrict__ offsets, int64_t lengths_size, int B, const IdxT* __restrict__ indices,
    const IdxT* __restrict__ block_sizes, int my_size, const float* __restrict__ weights,
    const int64_t* __restrict__ new_offsets, IdxT* __restrict__ new_indices, float* __restrict__ new_weights,
    IdxT* __restrict__ new_pos, IdxT* __restrict__ unbucketize_permute) {
  extern __shared__ int32_t cur[];  // [my_size][256] running position inside (bucket, bag)
  for (int p = 0; p < my_size; ++p) cur[p * 256 + threadIdx.x] = 0;
  const int64_t bag = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bag >= lengths_size) return;
  const int f = static_cast<int>(bag / B);
  const uint64_t blk = static_cast<uint64_t>(block_sizes[f]);
  const int64_t s = offsets[bag], e = offsets[bag + 1];
  for (int64_t i = s; i < e; ++i) {
    const uint64_t idx = static_cast<uint64_t>(indices[i]);
    const uint64_t p = idx / blk;
    if (p >= static_cast<uint64_t>(my_size)) {
      if (unbucketize_permute != nullptr) unbucketize_permute[i] = static_cast<IdxT>(-1);
      continue;
    }
    const int c = cur[p * 256 + threadIdx.x];
    cur[p * 256 + threadIdx.x] = c + 1;
    const int64_t dst = new_offsets[static_cast<int64_t>(p) * lengths_size + bag] + c;
    new_indices[dst] = static_cast<IdxT>(idx - p * blk);
    if (weights != nullptr) new_weights[dst] = weights[i];
    if (new_pos != nullptr) new_pos[dst] = static_cast<IdxT>(i - s);
    if (unbucketize_permute != nullptr) unbucketize_permute[i] = static_cast<IdxT>(dst);
  }
}

// ---------------------------------------------------------------------------------------
// Pooled all-to-all layout copies: [src][B_local][D_src] slabs <-> [B_local, sum D_src].
// One thread per 16-B (VEC=4) or 4-B (VEC=1) element of the [B_local, D_total] matrix.
// ---------------------------------------------------------------------------------------
template <int VEC, bool PACK>
__global__ __launch_bounds__(256) void a2a_pooled_layout_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               const int32_t* __restrict__ dim_sum_per_rank, int W,
                                                               int B_local, int D_total, float scale) {
  extern __shared__ int32_t cum[];  // [W+1]
  if (threadIdx.x == 0) {
    int c = 0;
    for (int r = 0; r < W; ++r) {
      cum[r] = c;
      c += dim_sum_per_rank[r];
    }
    cum[W] = c;
  }
  __syncthreads();
  const int cols = D_total / VEC;
  const int64_t total = static_cast<int64_t>(B_local) * cols;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / cols);
    const int d = static_cast<int>(i - static_cast<int64_t>(b) * cols) * VEC;
    int lo = 0, hi = W;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] <= d) lo = mid; else hi = mid;
    }
    const int Dr = cum[lo + 1] - cum[lo];
    const int64_t slab = static_cast<int64_t>(B_local) * cum[lo] + static_cast<int64_t>(b) * Dr + (d - cum[lo]);
    const int64_t mat = static_cast<int64_t>(b) * D_total + d;
    const int64_t si = PACK ? mat : slab;
    const int64_t di = PACK ? slab : mat;
    if (VEC == 4) {
      float4 v = ld4(src + si);
      v.x *= scale;
      v.y *= scale;
      v.z *= scale;
      v.w *= scale;
      st4(dst + di, v);
    } else {
      dst[di] = src[si] * scale;
    }
  }
}

__global__ __launch_bounds__(256) void jagged_2d_to_dense_kernel(const float* __restrict__ values,
                                                                const int64_t* __restrict__ offsets, int B, int D,
                                                                int max_L, float* __restrict__ dense) {
  const int64_t total = static_cast<int64_t>(B) * max_L * D;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int d = static_cast<int>(i % D);
    const int64_t r = i / D;
    const int l = static_cast<int>(r % max_L);
    const int b = static_cast<int>(r / max_L);
    const int64_t s = offsets[b];
    const int64_t len = offsets[b + 1] - s;
    dense[i] = l < len ? values[(s + l) * D + d] : 0.f;
  }
}

// Gradient of jagged_2d_to_dense: values_grad[offsets[b] + l, :] = dense_grad[b, l, :] for l < min(len_b, max_L),
// zero for the truncated tail (l >= max_L).
__global__ __launch_bounds__(256) void dense_to_jagged_2d_kernel(const float* __restrict__ dense,
                                                                const int64_t* __restrict__ offsets, int B, int D,
                                                                int max_L, int64_t N, float* __restrict__ values) {
  const int64_t total = N * D;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int d = static_cast<int>(i % D);
    const int64_t row = i / D;
    int lo = 0, hi = B;  // largest b with offsets[b] <= row
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= row) lo = mid; else hi = mid;
    }
    const int64_t l = row - offsets[lo];
    values[i] = l < max_L ? dense[(static_cast<int64_t>(lo) * max_L + l) * D + d] : 0.f;
  }
}

// Rows `rows[y]` of a [*, row_vecs] matrix of 16-byte vectors into row y of a dense output: the send-order gather of the
// input exchange (torchrec/distributed/dist_data.py:257-263 permutes whole KJTs for this; fixed-length streams need only rows).
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ rows,
                                                       int64_t row_vecs, uint4* __restrict__ dst) {
  const uint4* in = src + static_cast<int64_t>(rows[blockIdx.y]) * row_vecs;
  uint4* out = dst + static_cast<int64_t>(blockIdx.y) * row_vecs;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < row_vecs;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    out[i] = in[i];
}

__global__ __launch_bounds__(256) void offsets_range_kernel(const int64_t* __restrict__ offsets, int64_t n,
                                                           int64_t range_size, int64_t* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < range_size;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t lo = 0, hi = n;  // largest k with offsets[k] <= i
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    out[i] = i - offsets[lo];
  }
}

static unsigned grid_for(int64_t n, int per_block = 256, int64_t cap = 256 * 16) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<unsigned>(g);
}

}  // namespace tbe

using namespace tbe;

extern "C" size_t tbe_cumsum_workspace_bytes(int64_t n) { return scan_ws_bytes(n); }

extern "C" int tbe_cumsum(const void* in, void* out, int64_t n, int32_t elem_size, int32_t mode, void* workspace,
                          size_t workspace_bytes, void* stream) {
  TBE_REQUIRE(n >= 0, "tbe_cumsum: n < 0");
  TBE_REQUIRE(elem_size == 4 || elem_size == 8, "tbe_cumsum: elem_size %d", elem_size);
  TBE_REQUIRE(mode >= 0 && mode <= 2, "tbe_cumsum: mode %d", mode);
  if (n == 0 && mode != 0) return TBE_OK;  // nothing to write
  TBE_REQUIRE(out != nullptr && (n == 0 || in != nullptr), "tbe_cumsum: null pointer");
  if (n > kScanTile) {
    TBE_REQUIRE(workspace != nullptr && workspace_bytes >= scan_ws_bytes(n), "tbe_cumsum: workspace too small");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t* ws = static_cast<int64_t*>(workspace);
  if (elem_size == 4)
    return run_scan<int32_t, int32_t>(static_cast<const int32_t*>(in), static_cast<int32_t*>(out), n, nullptr, 1, mode, ws, nullptr, st);
  return run_scan<int64_t, int64_t>(static_cast<const int64_t*>(in), static_cast<int64_t*>(out), n, nullptr, 1, mode, ws, nullptr, st);
}

extern "C" size_t tbe_permute_2d_workspace_bytes(int32_t T_in, int32_t T_out, int32_t B) {
  const int64_t n = static_cast<int64_t>(std::max(T_in, T_out)) * B;
  return scan_ws_bytes(n);
}

extern "C" int tbe_permute_2d_lengths(const int32_t* permute, int32_t T_in, int32_t T_out, int32_t B,
                                      const void* lengths, int32_t len_elem_size, void* out_lengths,
                                      int64_t* in_offsets, int64_t* out_offsets, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  TBE_REQUIRE(T_in >= 0 && T_out >= 0 && B >= 0, "tbe_permute_2d_lengths: bad sizes");
  TBE_REQUIRE(len_elem_size == 4 || len_elem_size == 8, "tbe_permute_2d_lengths: len_elem_size %d", len_elem_size);
  TBE_REQUIRE(in_offsets && out_offsets, "tbe_permute_2d_lengths: null offsets");
  TBE_REQUIRE(workspace != nullptr && workspace_bytes >= tbe_permute_2d_workspace_bytes(T_in, T_out, B),
              "tbe_permute_2d_lengths: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t* ws = static_cast<int64_t*>(workspace);
  const int64_t n_in = static_cast<int64_t>(T_in) * B;
  const int64_t n_out = static_cast<int64_t>(T_out) * B;
  const int Bs = std::max(B, 1);
  int rc;
  if (len_elem_size == 4) {
    rc = run_scan<int32_t, int64_t>(static_cast<const int32_t*>(lengths), in_offsets, n_in, nullptr, Bs, 0, ws, nullptr, st);
    if (rc != TBE_OK) return rc;
    rc = run_scan<int32_t, int64_t>(static_cast<const int32_t*>(lengths), out_offsets, n_out, permute, Bs, 0, ws,
                                    static_cast<int32_t*>(out_lengths), st);
  } else {
    rc = run_scan<int64_t, int64_t>(static_cast<const int64_t*>(lengths), in_offsets, n_in, nullptr, Bs, 0, ws, nullptr, st);
    if (rc != TBE_OK) return rc;
    rc = run_scan<int64_t, int64_t>(static_cast<const int64_t*>(lengths), out_offsets, n_out, permute, Bs, 0, ws,
                                    static_cast<int64_t*>(out_lengths), st);
  }
  return rc;
}

template <typename V>
static int launch_permute_data(const int32_t* permute, int T_out, int B, const int64_t* in_offsets,
                               const int64_t* out_offsets, const void* values, void* out_values, const void* weights,
                               void* out_weights, int w_elem, hipStream_t st) {
  const int64_t nseg = static_cast<int64_t>(T_out) * B;
  const unsigned grid = static_cast<unsigned>((nseg + 255) / 256);
#define TBE_P(WT)                                                                                              \
  hipLaunchKernelGGL((permute_2d_data_kernel<V, WT>), dim3(grid), dim3(256), 0, st, permute, T_out, B, in_offsets, \
                     out_offsets, static_cast<const V*>(values), static_cast<V*>(out_values),                  \
                     static_cast<const WT*>(weights), static_cast<WT*>(out_weights))
  switch (w_elem) {
    case 1: TBE_P(uint8_t); break;
    case 2: TBE_P(uint16_t); break;
    case 8: TBE_P(uint64_t); break;
    default: TBE_P(uint32_t); break;
  }
#undef TBE_P
  TBE_CHECK_LAUNCH("tbe_permute_2d_data");
  return TBE_OK;
}

extern "C" int tbe_permute_2d_data(const int32_t* permute, int32_t T_out, int32_t B, const int64_t* in_offsets,
                                   const int64_t* out_offsets, const void* values, void* out_values,
                                   int32_t val_elem_size, const void* weights, void* out_weights,
                                   int32_t w_elem_size, void* stream) {
  TBE_REQUIRE(T_out >= 0 && B >= 0, "tbe_permute_2d_data: bad sizes");
  if (static_cast<int64_t>(T_out) * B == 0) return TBE_OK;
  TBE_REQUIRE(permute && in_offsets && out_offsets, "tbe_permute_2d_data: null pointer");
  TBE_REQUIRE(val_elem_size == 1 || val_elem_size == 2 || val_elem_size == 4 || val_elem_size == 8,
              "tbe_permute_2d_data: val_elem_size %d", val_elem_size);
  if (weights != nullptr)
    TBE_REQUIRE(w_elem_size == 1 || w_elem_size == 2 || w_elem_size == 4 || w_elem_size == 8,
                "tbe_permute_2d_data: w_elem_size %d", w_elem_size);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int we = weights != nullptr ? w_elem_size : 4;
  switch (val_elem_size) {
    case 1: return launch_permute_data<uint8_t>(permute, T_out, B, in_offsets, out_offsets, values, out_values, weights, out_weights, we, st);
    case 2: return launch_permute_data<uint16_t>(permute, T_out, B, in_offsets, out_offsets, values, out_values, weights, out_weights, we, st);
    case 4: return launch_permute_data<uint32_t>(permute, T_out, B, in_offsets, out_offsets, values, out_values, weights, out_weights, we, st);
    default: return launch_permute_data<uint64_t>(permute, T_out, B, in_offsets, out_offsets, values, out_values, weights, out_weights, we, st);
  }
}

namespace {
struct BucketizeWs {
  int64_t* offsets;      // [lengths_size + 1]
  int64_t* new_offsets;  // [my_size*lengths_size + 1]
  int32_t* cursor;       // [my_size*lengths_size]
  int64_t* scan_ws;
  size_t total;
};
BucketizeWs carve_bucketize(void* ws, int64_t lengths_size, int32_t my_size) {
  Carver c(ws);
  BucketizeWs w;
  const int64_t nl = lengths_size * my_size;
  w.offsets = c.take<int64_t>(lengths_size + 1);
  w.new_offsets = c.take<int64_t>(nl + 1);
  w.cursor = c.take<int32_t>(nl);
  w.scan_ws = static_cast<int64_t*>(c.take_bytes(scan_ws_bytes(std::max(nl, lengths_size))));
  w.total = c.total();
  return w;
}

template <typename LenT, typename IdxT>
int run_bucketize(const void* lengths, int64_t lengths_size, const void* indices, int64_t N, const void* block_sizes,
                  int F, int my_size, const float* weights, bool bucketize_pos, bool sequence, void* new_lengths,
                  void* new_indices, float* new_weights, void* new_pos, void* unbucketize_permute,
                  const BucketizeWs& w, hipStream_t st) {
  (void)N;
  const int B = static_cast<int>(lengths_size / F);
  const int64_t nl = lengths_size * my_size;
  int rc = run_scan<LenT, int64_t>(static_cast<const LenT*>(lengths), w.offsets, lengths_size, nullptr, 1, 0, w.scan_ws, nullptr, st);
  if (rc != TBE_OK) return rc;
  const unsigned grid = static_cast<unsigned>((lengths_size + 255) / 256);
  if (my_size <= kBucketizeLdsMaxBuckets) {
    // per-thread bucket counters / cursors live in LDS ([bucket][thread], conflict-free): no global
    // read-modify-write, no memsets, new_lengths written coalesced
    const size_t lds = static_cast<size_t>(my_size) * 256 * sizeof(int32_t);
    hipLaunchKernelGGL((bucketize_count_lds_kernel<LenT, IdxT>), dim3(grid), dim3(256), lds, st, w.offsets, lengths_size, B,
                       static_cast<const IdxT*>(indices), static_cast<const IdxT*>(block_sizes), my_size,
                       static_cast<LenT*>(new_lengths));
    TBE_CHECK_LAUNCH("bucketize count");
    rc = run_scan<LenT, int64_t>(static_cast<const LenT*>(new_lengths), w.new_offsets, nl, nullptr, 1, 0, w.scan_ws, nullptr, st);
    if (rc != TBE_OK) return rc;
    hipLaunchKernelGGL((bucketize_scatter_lds_kernel<LenT, IdxT>), dim3(grid), dim3(256), lds, st, w.offsets, lengths_size, B,
                       static_cast<const IdxT*>(indices), static_cast<const IdxT*>(block_sizes), my_size, weights,
                       w.new_offsets, static_cast<IdxT*>(new_indices), new_weights,
                       bucketize_pos ? static_cast<IdxT*>(new_pos) : nullptr,
                       sequence ? static_cast<IdxT*>(unbucketize_permute) : nullptr);
    TBE_CHECK_LAUNCH("bucketize scatter");
    return TBE_OK;
  }
  (void)hipMemsetAsync(new_lengths, 0, nl * sizeof(LenT), st);
  (void)hipMemsetAsync(w.cursor, 0, nl * sizeof(int32_t), st);
  hipLaunchKernelGGL((bucketize_count_kernel<LenT, IdxT>), dim3(grid), dim3(256), 0, st, w.offsets, lengths_size, B,
                     static_cast<const IdxT*>(indices), static_cast<const IdxT*>(block_sizes), my_size,
                     static_cast<LenT*>(new_lengths));
  TBE_CHECK_LAUNCH("bucketize count");
  rc = run_scan<LenT, int64_t>(static_cast<const LenT*>(new_lengths), w.new_offsets, nl, nullptr, 1, 0, w.scan_ws, nullptr, st);
  if (rc != TBE_OK) return rc;
  hipLaunchKernelGGL((bucketize_scatter_kernel<LenT, IdxT>), dim3(grid), dim3(256), 0, st, w.offsets, lengths_size, B,
                     static_cast<const IdxT*>(indices), static_cast<const IdxT*>(block_sizes), my_size, weights,
                     w.new_offsets, w.cursor, static_cast<IdxT*>(new_indices), new_weights,
                     bucketize_pos ? static_cast<IdxT*>(new_pos) : nullptr,
                     sequence ? static_cast<IdxT*>(unbucketize_permute) : nullptr);
  TBE_CHECK_LAUNCH("bucketize scatter");
  return TBE_OK;
}
}  // namespace

extern "C" size_t tbe_bucketize_workspace_bytes(int64_t lengths_size, int32_t my_size) {
  return carve_bucketize(nullptr, std::max<int64_t>(lengths_size, 0), std::max(my_size, 1)).total;
}

extern "C" int tbe_block_bucketize(const void* lengths, int32_t len_elem_size, int64_t lengths_size,
                                   const void* indices, int32_t idx_elem_size, int64_t N, const void* block_sizes,
                                   int32_t F, int32_t my_size, const float* weights, int32_t bucketize_pos,
                                   int32_t sequence, void* new_lengths, void* new_indices, float* new_weights,
                                   void* new_pos, void* unbucketize_permute, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  TBE_REQUIRE(F > 0 && my_size > 0 && lengths_size >= 0 && N >= 0, "tbe_block_bucketize: bad sizes");
  TBE_REQUIRE(lengths_size % F == 0, "tbe_block_bucketize: lengths_size %lld not a multiple of F=%d", (long long)lengths_size, F);
  TBE_REQUIRE(len_elem_size == 4 || len_elem_size == 8, "tbe_block_bucketize: len_elem_size %d", len_elem_size);
  TBE_REQUIRE(idx_elem_size == 4 || idx_elem_size == 8, "tbe_block_bucketize: idx_elem_size %d", idx_elem_size);
  if (lengths_size == 0) return TBE_OK;
  TBE_REQUIRE(lengths && block_sizes && new_lengths && workspace, "tbe_block_bucketize: null pointer");
  TBE_REQUIRE(N == 0 || (indices && new_indices), "tbe_block_bucketize: null indices");
  TBE_REQUIRE(weights == nullptr || new_weights != nullptr, "tbe_block_bucketize: weights without new_weights");
  TBE_REQUIRE(!bucketize_pos || new_pos != nullptr || N == 0, "tbe_block_bucketize: bucketize_pos without new_pos");
  TBE_REQUIRE(!sequence || unbucketize_permute != nullptr || N == 0, "tbe_block_bucketize: sequence without unbucketize_permute");
  TBE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "tbe_block_bucketize: workspace must be 256-B aligned");
  BucketizeWs w = carve_bucketize(workspace, lengths_size, my_size);
  if (w.total > workspace_bytes) {
    set_error("tbe_block_bucketize: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return TBE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
#define TBE_B(LT, IT)                                                                                          \
  return run_bucketize<LT, IT>(lengths, lengths_size, indices, N, block_sizes, F, my_size, weights, bucketize_pos != 0, \
                               sequence != 0, new_lengths, new_indices, new_weights, new_pos, unbucketize_permute, w, st)
  if (len_elem_size == 4 && idx_elem_size == 4) TBE_B(int32_t, int32_t);
  if (len_elem_size == 4 && idx_elem_size == 8) TBE_B(int32_t, int64_t);
  if (len_elem_size == 8 && idx_elem_size == 4) TBE_B(int64_t, int32_t);
  TBE_B(int64_t, int64_t);
#undef TBE_B
}

static int a2a_layout(const float* src, float* dst, const int32_t* dims, int W, int B_local, int D_total, float scale,
                      bool pack, int vec_ok, hipStream_t st) {
  const size_t lds = (static_cast<size_t>(W) + 1) * sizeof(int32_t);
  const int64_t total = static_cast<int64_t>(B_local) * (D_total / (vec_ok ? 4 : 1));
  const unsigned grid = grid_for(total, 256, 256 * 32);
  if (vec_ok) {
    if (pack) hipLaunchKernelGGL((a2a_pooled_layout_kernel<4, true>), dim3(grid), dim3(256), lds, st, src, dst, dims, W, B_local, D_total, scale);
    else hipLaunchKernelGGL((a2a_pooled_layout_kernel<4, false>), dim3(grid), dim3(256), lds, st, src, dst, dims, W, B_local, D_total, scale);
  } else {
    if (pack) hipLaunchKernelGGL((a2a_pooled_layout_kernel<1, true>), dim3(grid), dim3(256), lds, st, src, dst, dims, W, B_local, D_total, scale);
    else hipLaunchKernelGGL((a2a_pooled_layout_kernel<1, false>), dim3(grid), dim3(256), lds, st, src, dst, dims, W, B_local, D_total, scale);
  }
  TBE_CHECK_LAUNCH("a2a pooled layout");
  return TBE_OK;
}

extern "C" int tbe_a2a_pooled_unpack(const float* recv, float* out, const int32_t* dim_sum_per_rank, int32_t W,
                                     int32_t B_local, int32_t D_total, int32_t dims_multiple_of_4, float scale,
                                     void* stream) {
  TBE_REQUIRE(W > 0 && B_local >= 0 && D_total >= 0, "tbe_a2a_pooled_unpack: bad sizes");
  if (static_cast<int64_t>(B_local) * D_total == 0) return TBE_OK;
  TBE_REQUIRE(recv && out && dim_sum_per_rank, "tbe_a2a_pooled_unpack: null pointer");
  const int vec = dims_multiple_of_4 && (D_total % 4 == 0) && ((reinterpret_cast<uintptr_t>(recv) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  return a2a_layout(recv, out, dim_sum_per_rank, W, B_local, D_total, scale, false, vec, static_cast<hipStream_t>(stream));
}

extern "C" int tbe_a2a_pooled_pack(const float* grad, float* send, const int32_t* dim_sum_per_rank, int32_t W,
                                   int32_t B_local, int32_t D_total, int32_t dims_multiple_of_4, float scale,
                                   void* stream) {
  TBE_REQUIRE(W > 0 && B_local >= 0 && D_total >= 0, "tbe_a2a_pooled_pack: bad sizes");
  if (static_cast<int64_t>(B_local) * D_total == 0) return TBE_OK;
  TBE_REQUIRE(grad && send && dim_sum_per_rank, "tbe_a2a_pooled_pack: null pointer");
  const int vec = dims_multiple_of_4 && (D_total % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(send) & 15) == 0);
  return a2a_layout(grad, send, dim_sum_per_rank, W, B_local, D_total, scale, true, vec, static_cast<hipStream_t>(stream));
}

extern "C" int tbe_jagged_2d_to_dense_f32(const float* values, const int64_t* offsets, int32_t B, int32_t D,
                                          int32_t max_L, float* dense, void* stream) {
  TBE_REQUIRE(B >= 0 && D >= 0 && max_L >= 0, "tbe_jagged_2d_to_dense_f32: bad sizes");
  const int64_t total = static_cast<int64_t>(B) * max_L * D;
  if (total == 0) return TBE_OK;
  TBE_REQUIRE(offsets && dense, "tbe_jagged_2d_to_dense_f32: null pointer");
  hipLaunchKernelGGL(jagged_2d_to_dense_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), values,
                     offsets, B, D, max_L, dense);
  TBE_CHECK_LAUNCH("tbe_jagged_2d_to_dense_f32");
  return TBE_OK;
}

extern "C" int tbe_dense_to_jagged_2d_f32(const float* dense, const int64_t* offsets, int32_t B, int32_t D,
                                          int32_t max_L, int64_t N, float* values, void* stream) {
  TBE_REQUIRE(B >= 0 && D >= 0 && max_L >= 0 && N >= 0, "tbe_dense_to_jagged_2d_f32: bad sizes");
  if (N * D == 0) return TBE_OK;
  TBE_REQUIRE(B > 0 && dense && offsets && values, "tbe_dense_to_jagged_2d_f32: null pointer");
  hipLaunchKernelGGL(dense_to_jagged_2d_kernel, dim3(grid_for(N * D)), dim3(256), 0, static_cast<hipStream_t>(stream), dense,
                     offsets, B, D, max_L, N, values);
  TBE_CHECK_LAUNCH("tbe_dense_to_jagged_2d_f32");
  return TBE_OK;
}

extern "C" int tbe_offsets_range(const int64_t* offsets, int64_t n, int64_t range_size, int64_t* out, void* stream) {
  TBE_REQUIRE(n >= 0 && range_size >= 0, "tbe_offsets_range: bad sizes");
  if (range_size == 0) return TBE_OK;
  TBE_REQUIRE(n > 0 && offsets && out, "tbe_offsets_range: null pointer / empty offsets");
  hipLaunchKernelGGL(offsets_range_kernel, dim3(grid_for(range_size)), dim3(256), 0, static_cast<hipStream_t>(stream), offsets, n,
                     range_size, out);
  TBE_CHECK_LAUNCH("tbe_offsets_range");
  return TBE_OK;
}

extern "C" int tbe_copy_rows(const void* src, int64_t src_rows, const int32_t* rows, int32_t n_rows, int64_t row_bytes, void* dst,
                             void* stream) {
  TBE_REQUIRE(n_rows >= 0 && src_rows >= 0 && row_bytes >= 0, "tbe_copy_rows: bad sizes");
  if (n_rows == 0 || row_bytes == 0) return TBE_OK;
  TBE_REQUIRE(src && rows && dst, "tbe_copy_rows: null pointer");
  TBE_REQUIRE(n_rows <= 65535, "tbe_copy_rows: more than 65535 rows");
  TBE_REQUIRE(row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0,
              "tbe_copy_rows: rows must be multiples of 16 bytes at 16-byte aligned addresses");
  const int64_t row_vecs = row_bytes / 16;
  const int64_t per_row = (row_vecs + 255) / 256;
  // ~8 vectors per thread once the launch fills the chip
  const int64_t want = std::max<int64_t>(1, std::min<int64_t>(per_row, std::max<int64_t>(2048 / n_rows, (per_row + 7) / 8)));
  hipLaunchKernelGGL(copy_rows_kernel, dim3(static_cast<unsigned>(want), static_cast<unsigned>(n_rows)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(src), rows, row_vecs, static_cast<uint4*>(dst));
  TBE_CHECK_LAUNCH("tbe_copy_rows");
  return TBE_OK;
}

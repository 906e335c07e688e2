// TBE forward for gfx950: gather + segment-sum over KeyedJaggedTensor bags.
//
// Reference call site: torchrec/distributed/batched_embedding_kernel.py:546-554
// (BaseBatchedEmbeddingBag.forward -> fbgemm SplitTableBatchedEmbeddingBagsCodegen, absent
// from the reference tree).  Semantics pinned by the reference's CPU
// EmbeddingBagCollection (torchrec/modules/embedding_modules.py:165-193).
//
// Design (HBM-bound gather; no MFMA):
//  * one 64-lane wave owns 64 consecutive bags of ONE feature, so the two offsets loads and
//    the first-index load are fully coalesced (512 B per wave-instruction);
//  * a row is read by a group of G lanes, 16 B per lane (G = 16/32/64 for D <= 64/128/256+),
//    so every wave-instruction moves whole 128-B lines of a row;
//  * U = 4 bags are processed concurrently per group (=> 8 independent 512-B row reads in
//    flight per wave at D = 128) which is what hides HBM latency at pooling factor 1;
//  * pooled output rows are written as 16 B per lane, contiguous per bag.
// Long bags use tbe_fwd_long_kernel: the group preloads G indices with one coalesced load
// and walks them with cross-lane broadcasts, 4 rows in flight, partial sums of the wave's
// groups combined through LDS.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace tbe {

struct FwdArgs {
  const uint64_t* feat_weights;
  const int32_t* feat_D;
  const int64_t* feat_out_offset;
  const int64_t* feat_rows;
  const int64_t* feat_window;  // [2F] (first global row, global rows) per feature, or nullptr
  const int32_t* feat_pooling;  // [F] TBE_POOL_SUM / TBE_POOL_MEAN per feature (with pooling_mode MEAN), or nullptr = uniform
  const int64_t* indices;
  const int64_t* offsets;
  const float* psw;
  float* out;
  int32_t* bounds_errors;
  int64_t out_stride;
  int64_t N;  // number of ids: bag ranges outside [0, N] are treated as empty (and counted), never dereferenced
  int32_t F;
  int32_t B;
  int32_t bags_per_wave;  // 64, or 16 for small launches (more waves => more rows in flight)
};

__device__ __forceinline__ void fma4(float4& a, float w, const float4& x) {
  a.x = fmaf(w, x.x, a.x);
  a.y = fmaf(w, x.y, a.y);
  a.z = fmaf(w, x.z, a.z);
  a.w = fmaf(w, x.w, a.w);
}

// Loads columns [d, d+4) of a row; `vec` selects the 16-B path, otherwise 4 guarded scalars.
__device__ __forceinline__ float4 load_cols(const float* row, int d, int D, bool vec) {
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec) {
    x = ld4(row + d);
  } else {
    if (d + 0 < D) x.x = row[d + 0];
    if (d + 1 < D) x.y = row[d + 1];
    if (d + 2 < D) x.z = row[d + 2];
    if (d + 3 < D) x.w = row[d + 3];
  }
  return x;
}
__device__ __forceinline__ void store_cols(float* row, int d, int D, bool vec, float4 x) {
  if (vec) {
    st4(row + d, x);
  } else {
    if (d + 0 < D) row[d + 0] = x.x;
    if (d + 1 < D) row[d + 1] = x.y;
    if (d + 2 < D) row[d + 2] = x.z;
    if (d + 3 < D) row[d + 3] = x.w;
  }
}


template <int G, int NV, bool WEIGHTED, bool MEAN>
__global__ __launch_bounds__(256) void tbe_fwd_short_kernel(FwdArgs a) {
  constexpr int NG = kWave / G;  // row groups per wave
  constexpr int U = 4;           // bags in flight per group
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int f = blockIdx.x % a.F;
  const int tile = blockIdx.x / a.F;
  const int bpw = a.bags_per_wave;
  const int bag0 = (tile * 4 + wave) * bpw;
  if (bag0 >= a.B) return;  // wave-uniform

  const float* __restrict__ W = reinterpret_cast<const float*>(a.feat_weights[f]);
  const int D = a.feat_D[f];
  const int64_t Doff = a.feat_out_offset[f];
  const RowWindow win = load_window(a.feat_rows, a.feat_window, f);
  const bool mean_f = MEAN && (a.feat_pooling == nullptr || a.feat_pooling[f] == TBE_POOL_MEAN);
  const bool vec = ((D & 3) == 0) && ((Doff & 3) == 0) && ((a.out_stride & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(W) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);

  // Coalesced metadata: lane l owns bag (f, bag0 + l).
  const int64_t* __restrict__ offs = a.offsets + static_cast<int64_t>(f) * a.B;
  const int b_l = bag0 + lane;
  int64_t s_l = 0, e_l = 0;
  if (b_l < a.B && lane < bpw) {
    s_l = offs[b_l];
    e_l = offs[b_l + 1];
    if (s_l < 0 || e_l > a.N || s_l > e_l) {  // malformed offsets: empty bag, counted, never dereferenced
      s_l = e_l = 0;
      if (a.bounds_errors != nullptr) atomicAdd(a.bounds_errors, 1);
    }
  }
  const int len_l = static_cast<int>(e_l - s_l);
  int64_t idx0_l = 0;
  float w0_l = 1.f;
  if (len_l > 0) {
    idx0_l = a.indices[s_l];
    if (WEIGHTED) w0_l = a.psw[s_l];
  }

  const int g = lane / G;
  const int gl = lane % G;
  int nbad = 0;

#pragma unroll 1
  for (int p = 0; p < bpw; p += NG * U) {
    if (bag0 + p >= a.B) break;  // wave-uniform tail
    int64_t s[U];
    int len[U];
    int64_t idx[U];
    float w[U];
    float4 acc[U][NV];
    int maxlen = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = p + u * NG + g;
      s[u] = shfl64(s_l, j);
      len[u] = __shfl(len_l, j, kWave);
      idx[u] = shfl64(idx0_l, j);
      w[u] = WEIGHTED ? __shfl(w0_l, j, kWave) : 1.f;
      maxlen = max(maxlen, len[u]);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[u][v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // First element of each bag: indices are already in registers -> U independent row reads.
    {
      float4 x[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t lix;
        const int cls = classify_id(win, idx[u], lix);
        const bool ok = len[u] > 0 && cls == kIdLocal;
        if (len[u] > 0 && cls == kIdBad) ++nbad;
        const float* row = W + lix * D;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          x[u][v] = (ok && d < D) ? load_cols(row, d, D, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) fma4(acc[u][v], w[u], x[u][v]);
    }
    // Remaining elements (pooling factor > 1).
    for (int i = 1; i < maxlen; ++i) {
      float4 x[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = i < len[u];
        int64_t ix = 0;
        if (in) {
          ix = a.indices[s[u] + i];
          if (WEIGHTED) w[u] = a.psw[s[u] + i];
        }
        int64_t lix;
        const int cls = classify_id(win, ix, lix);
        const bool ok = in && cls == kIdLocal;
        if (in && cls == kIdBad) ++nbad;
        const float* row = W + lix * D;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          x[u][v] = (ok && d < D) ? load_cols(row, d, D, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) fma4(acc[u][v], w[u], x[u][v]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = bag0 + p + u * NG + g;
      if (b < a.B && p + u * NG + g < bpw) {
        float scale = 1.f;
        if (mean_f) scale = len[u] > 0 ? 1.f / static_cast<float>(len[u]) : 0.f;
        float* orow = a.out + static_cast<int64_t>(b) * a.out_stride + Doff;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          if (d < D) {
            float4 r = acc[u][v];
            if (mean_f) {
              r.x *= scale;
              r.y *= scale;
              r.z *= scale;
              r.w *= scale;
            }
            store_cols(orow, d, D, vec, r);
          }
        }
      }
    }
  }
  if (a.bounds_errors != nullptr && nbad > 0 && gl == 0) atomicAdd(a.bounds_errors, nbad);
}

// Long-bag variant: one wave per bag.  The wave's NG groups take alternate rows of the bag;
// each group preloads its next G indices with one coalesced load and keeps 4 row reads in
// flight; the NG partial sums are combined through LDS in fixed group order (deterministic).
template <int G, int NV, bool WEIGHTED, bool MEAN>
__global__ __launch_bounds__(256) void tbe_fwd_long_kernel(FwdArgs a) {
  constexpr int NG = kWave / G;
  constexpr int U = 4;
  __shared__ float4 part[4][NG > 1 ? NG : 1][NV][G];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t bag = static_cast<int64_t>(blockIdx.x) * 4 + wave;  // f*B + b
  const int64_t nbags = static_cast<int64_t>(a.F) * a.B;
  if (bag >= nbags) return;
  const int f = static_cast<int>(bag / a.B);
  const int b = static_cast<int>(bag % a.B);
  const float* __restrict__ W = reinterpret_cast<const float*>(a.feat_weights[f]);
  const int D = a.feat_D[f];
  const int64_t Doff = a.feat_out_offset[f];
  const RowWindow win = load_window(a.feat_rows, a.feat_window, f);
  const bool mean_f = MEAN && (a.feat_pooling == nullptr || a.feat_pooling[f] == TBE_POOL_MEAN);
  const bool vec = ((D & 3) == 0) && ((Doff & 3) == 0) && ((a.out_stride & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(W) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  int64_t s = a.offsets[bag];
  int64_t e = a.offsets[bag + 1];
  if (s < 0 || e > a.N || s > e) {  // malformed offsets: empty bag, counted, never dereferenced (wave-uniform)
    s = e = 0;
    if (a.bounds_errors != nullptr && lane == 0) atomicAdd(a.bounds_errors, 1);
  }
  const int len = static_cast<int>(e - s);
  const int g = lane / G;
  const int gl = lane % G;
  int nbad = 0;

  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);

  // The wave walks the bag in chunks of 64 indices; lane l holds index s + c + l.
  for (int c = 0; c < len; c += kWave) {
    const int n = min(kWave, len - c);
    int64_t ix_l = 0;
    float w_l = 1.f;
    if (lane < n) {
      ix_l = a.indices[s + c + lane];
      if (WEIGHTED) w_l = a.psw[s + c + lane];
    }
    // group g takes elements g, g+NG, g+2NG, ... of the chunk (fixed assignment).
    for (int k = 0; k < n; k += NG * U) {
      float4 x[U][NV];
      float w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = k + u * NG + g;
        const int64_t ix = shfl64(ix_l, j & 63);
        w[u] = WEIGHTED ? __shfl(w_l, j & 63, kWave) : 1.f;
        const bool in = j < n;
        int64_t lix;
        const int cls = classify_id(win, ix, lix);
        const bool ok = in && cls == kIdLocal;
        if (in && cls == kIdBad) ++nbad;
        const float* row = W + lix * D;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          x[u][v] = (ok && d < D) ? load_cols(row, d, D, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) fma4(acc[v], w[u], x[u][v]);
    }
  }
  if (NG > 1) {
#pragma unroll
    for (int v = 0; v < NV; ++v) part[wave][g][v][gl] = acc[v];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (g == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        for (int og = 1; og < NG; ++og) {
          const float4 o = part[wave][og][v][gl];
          acc[v].x += o.x;
          acc[v].y += o.y;
          acc[v].z += o.z;
          acc[v].w += o.w;
        }
      }
    }
  }
  if (g == 0) {
    float scale = 1.f;
    if (mean_f) scale = len > 0 ? 1.f / static_cast<float>(len) : 0.f;
    float* orow = a.out + static_cast<int64_t>(b) * a.out_stride + Doff;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) {
        float4 r = acc[v];
        if (mean_f) {
          r.x *= scale;
          r.y *= scale;
          r.z *= scale;
          r.w *= scale;
        }
        store_cols(orow, d, D, vec, r);
      }
    }
  }
  if (a.bounds_errors != nullptr && nbad > 0 && gl == 0) atomicAdd(a.bounds_errors, nbad);
}

// PoolingMode.NONE: out[i, :] = W_f(i)[indices[i], :].  One G-lane group per index,
// 4 indices in flight per group; the feature of position i is found by a search over the
// F+1 feature boundaries offsets[f*B] held in LDS.
template <int G, int NV>
__global__ __launch_bounds__(256) void tbe_fwd_nobag_kernel(const uint64_t* feat_weights,
                                                           const int64_t* feat_rows, int F, int B,
                                                           int D, const int64_t* indices,
                                                           int64_t N, const int64_t* offsets,
                                                           float* out, int32_t* bounds_errors) {
  extern __shared__ int64_t fb[];  // [F+1] feature boundaries
  for (int i = threadIdx.x; i <= F; i += blockDim.x) fb[i] = offsets[static_cast<int64_t>(i) * B];
  __syncthreads();
  constexpr int NG = kWave / G;
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / G;
  const int gl = lane % G;
  const bool vec = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int64_t per_wave = NG * U;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * 4;
  int nbad = 0;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * per_wave; base < N;
       base += nwaves * per_wave) {
    float4 x[U][NV];
    int64_t pos[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pos[u] = base + u * NG + g;
      const bool in = pos[u] < N;
      int64_t ix = 0;
      int f = 0;
      if (in) {
        ix = indices[pos[u]];
        int lo = 0, hi = F;  // largest f with fb[f] <= pos
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (fb[mid] <= pos[u]) lo = mid; else hi = mid;
        }
        f = lo;
      }
      const float* W = reinterpret_cast<const float*>(feat_weights[f]);
      const bool ok = in && static_cast<uint64_t>(ix) < static_cast<uint64_t>(feat_rows[f]);
      if (in && !ok) ++nbad;
      const bool v16 = vec && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
      const float* row = W + ix * D;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int d = (v * G + gl) * 4;
        x[u][v] = (ok && d < D) ? load_cols(row, d, D, v16) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (pos[u] < N) {
        float* orow = out + pos[u] * D;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          if (d < D) store_cols(orow, d, D, vec, x[u][v]);
        }
      }
    }
  }
  if (bounds_errors != nullptr && nbad > 0 && gl == 0) atomicAdd(bounds_errors, nbad);
}

template <int G, int NV>
static int launch_fwd(const FwdArgs& a, bool weighted, bool mean, bool long_bags,
                      hipStream_t st) {
  ProfileSpan span(TBE_PROFILE_FWD_KERNEL, st);
  if (long_bags) {
    const int64_t nbags = static_cast<int64_t>(a.F) * a.B;
    const unsigned grid = static_cast<unsigned>((nbags + 3) / 4);
#define TBE_L(WG, MN) hipLaunchKernelGGL((tbe_fwd_long_kernel<G, NV, WG, MN>), dim3(grid), dim3(256), 0, st, a)
    if (weighted && mean) TBE_L(true, true);
    else if (weighted) TBE_L(true, false);
    else if (mean) TBE_L(false, true);
    else TBE_L(false, false);
#undef TBE_L
  } else {
    const unsigned tiles = (a.B + 4 * a.bags_per_wave - 1) / (4 * a.bags_per_wave);
    const unsigned grid = tiles * a.F;
#define TBE_S(WG, MN) hipLaunchKernelGGL((tbe_fwd_short_kernel<G, NV, WG, MN>), dim3(grid), dim3(256), 0, st, a)
    if (weighted && mean) TBE_S(true, true);
    else if (weighted) TBE_S(true, false);
    else if (mean) TBE_S(false, true);
    else TBE_S(false, false);
#undef TBE_S
  }
  TBE_CHECK_LAUNCH("tbe_forward_pooled_f32");
  return TBE_OK;
}

}  // namespace tbe

using namespace tbe;

extern "C" int tbe_forward_pooled_f32(const uint64_t* feat_weights, const int32_t* feat_D,
                                      const int64_t* feat_out_offset, const int64_t* feat_rows,
                                      int32_t F, int32_t B, int32_t max_D,
                                      const int64_t* indices, int64_t N, const int64_t* offsets,
                                      const float* per_sample_weights, int32_t pooling_mode,
                                      const int32_t* feat_pooling, float* out, int64_t out_row_stride,
                                      int32_t* bounds_errors, const int64_t* feat_window, void* stream) {
  TBE_REQUIRE(F > 0 && B >= 0 && N >= 0, "tbe_forward_pooled_f32: bad sizes F=%d B=%d N=%lld", F, B,
              (long long)N);
  TBE_REQUIRE(pooling_mode == TBE_POOL_SUM || pooling_mode == TBE_POOL_MEAN,
              "tbe_forward_pooled_f32: pooling_mode %d is not pooled", pooling_mode);
  TBE_REQUIRE(max_D > 0 && max_D <= 2048, "tbe_forward_pooled_f32: max_D=%d outside (0, 2048]", max_D);
  TBE_REQUIRE(out_row_stride > 0, "tbe_forward_pooled_f32: out_row_stride <= 0");
  if (B == 0) return TBE_OK;
  TBE_REQUIRE(feat_weights && feat_D && feat_out_offset && feat_rows && offsets && out,
              "tbe_forward_pooled_f32: null pointer");
  TBE_REQUIRE(N == 0 || indices != nullptr, "tbe_forward_pooled_f32: null indices");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FwdArgs a{feat_weights, feat_D, feat_out_offset, feat_rows, feat_window, feat_pooling, indices, offsets, per_sample_weights,
            out, bounds_errors, out_row_stride, N, F, B, 64};
  // small launches: 4x more waves (16 bags each) keep more row reads in flight per CU
  if (static_cast<int64_t>(F) * B < (static_cast<int64_t>(1) << 19)) a.bags_per_wave = 16;
  const bool weighted = per_sample_weights != nullptr;
  const bool mean = pooling_mode == TBE_POOL_MEAN;
  const double avg_len = static_cast<double>(N) / (static_cast<double>(F) * B);
  static const double long_min = [] {
    const char* e = getenv("TBE_FWD_LONG_MIN");  // tuning knob
    return e ? atof(e) : 3.5;  // measured on MI355X: the wave-per-bag kernel wins from ~4 ids per bag
  }();
  const bool long_bags = avg_len >= long_min;
  if (max_D <= 64) return launch_fwd<16, 1>(a, weighted, mean, long_bags, st);
  if (max_D <= 128) return launch_fwd<32, 1>(a, weighted, mean, long_bags, st);
  if (max_D <= 256) return launch_fwd<64, 1>(a, weighted, mean, long_bags, st);
  if (max_D <= 512) return launch_fwd<64, 2>(a, weighted, mean, long_bags, st);
  if (max_D <= 1024) return launch_fwd<64, 4>(a, weighted, mean, long_bags, st);
  return launch_fwd<64, 8>(a, weighted, mean, long_bags, st);
}

extern "C" int tbe_forward_nobag_f32(const uint64_t* feat_weights, const int64_t* feat_rows,
                                     int32_t F, int32_t B, int32_t D, const int64_t* indices,
                                     int64_t N, const int64_t* offsets, float* out,
                                     int32_t* bounds_errors, void* stream) {
  TBE_REQUIRE(F > 0 && B >= 0 && N >= 0 && D > 0 && D <= 2048, "tbe_forward_nobag_f32: bad sizes");
  if (N == 0) return TBE_OK;
  TBE_REQUIRE(feat_weights && feat_rows && indices && offsets && out, "tbe_forward_nobag_f32: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = (static_cast<size_t>(F) + 1) * sizeof(int64_t);
  TBE_REQUIRE(lds <= 60000, "tbe_forward_nobag_f32: too many features (%d)", F);
#define TBE_N(G, NV)                                                                              \
  do {                                                                                            \
    const int64_t per_block = 4 * (kWave / G) * 4;                                                \
    unsigned grid = static_cast<unsigned>(std::min<int64_t>((N + per_block - 1) / per_block, 256 * 16)); \
    hipLaunchKernelGGL((tbe_fwd_nobag_kernel<G, NV>), dim3(grid), dim3(256), lds, st, feat_weights, \
                       feat_rows, F, B, D, indices, N, offsets, out, bounds_errors);              \
  } while (0)
  if (D <= 64) TBE_N(16, 1);
  else if (D <= 128) TBE_N(32, 1);
  else if (D <= 256) TBE_N(64, 1);
  else if (D <= 512) TBE_N(64, 2);
  else if (D <= 1024) TBE_N(64, 4);
  else TBE_N(64, 8);
#undef TBE_N
  TBE_CHECK_LAUNCH("tbe_forward_nobag_f32");
  return TBE_OK;
}

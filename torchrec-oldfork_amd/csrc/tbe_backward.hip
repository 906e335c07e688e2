// TBE backward + fused "exact" optimizer for gfx950.
//
// Reference wiring: torchrec/distributed/batched_embedding_kernel.py:604-665
// (BatchedFusedEmbeddingBag) and :53-257 (EmbeddingFusedOptimizer); the arithmetic itself
// lives in fbgemm_gpu, which is absent from the reference tree.  "Exact" optimizers sum
// every contribution a batch makes to one table row BEFORE applying a single update
// (pinned for EXACT_SGD by torchrec/distributed/test_utils/test_model_parallel_base.py:257-283,
// which compares against nn.EmbeddingBag + torch.optim.SGD).
//
// Pipeline (all on `stream`, no host sync, no atomics on floats => bitwise reproducible):
//  1. linearize : key[p] = feat_row_base[f] + indices[p]  (invalid index -> sentinel),
//                 payload[p] = bag  or  (bag << 32) | p    (thread per bag, coalesced at L = 1)
//  2. sort      : stable LSD radix sort of (key, payload) on the low key_bits bits
//                 (hand-written, radix_sort.hpp: one launch per 10-bit digit pass).  The payload is
//                 the 32-bit bag number for pooled lookups without per-sample weights (the update
//                 needs nothing else), (bag << 32) | position otherwise.
//  3. update    : the sorted contributions are cut into fixed chunks of C; one G-lane group
//                 walks a chunk, 4 gradient rows + 4 weight rows in flight, accumulates
//                 runs of equal keys in registers and applies the optimizer when a run
//                 ends inside the chunk.  Runs that cross a chunk boundary leave a
//                 partial row in the workspace.
//  4. fixup     : the chunk where a crossing run started sums the partials in chunk
//                 order and applies the update.
// Fixed chunks keep the work balanced no matter how skewed the ids are (a 3-row table
// receives B contributions per row).
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "common.hpp"
#include "radix_sort.hpp"

namespace tbe {

struct BwdArgs {
  const uint64_t* feat_weights;
  const int32_t* feat_D;
  const int64_t* feat_out_offset;
  const int64_t* feat_rows;
  const int64_t* feat_row_base;
  const int64_t* feat_window;
  const int32_t* feat_pooling;  // per-feature SUM / MEAN under pooling_mode MEAN, or nullptr = uniform
  const uint64_t* feat_state0;
  const uint64_t* feat_state1;
  const int64_t* indices;
  const int64_t* offsets;
  const float* psw;
  const float* grad_out;
  int64_t grad_stride;
  int64_t N;
  int32_t F;
  int32_t B;
  int32_t pooling_mode;
  int32_t key_bits;
  int32_t C;  // contributions per chunk
  tbe_optimizer_args opt;
  float bias1;  // ADAM: 1 - beta1^t
  float bias2;  // ADAM: 1 - beta2^t
  // workspace
  void* keys_sorted;
  const void* payload_sorted;  // uint32 bag numbers, or uint64 (bag << 32) | position
  float* partial_first;  // [nchunks][max_D_pad]
  float* partial_last;   // [nchunks][max_D_pad]
  int32_t* origin_list;  // [nchunks] chunks whose last run continues (compacted, any order)
  int32_t* origin_count; // [1]
  int32_t max_D_pad;
  int32_t fast_D;  // uniform feature dim when TBE_FLAG_UNIFORM_ALIGNED, else 0
  int32_t* bounds_errors;
  unsigned long long* unique_rows;  // optional profiling counter: table rows updated
};

// One launch instead of two memsets: every key starts as all-ones (= invalid: positions that inconsistent
// offsets leave uncovered must not reach the update kernel as garbage rows) and the sort's state block
// (tickets, digit totals, histogram rows; origin_count lives among the tickets) starts as zero.
__global__ __launch_bounds__(256) void bwd_fill_kernel(uint4* __restrict__ keys16, int64_t n_keys16,
                                                        uint4* __restrict__ state16, int64_t n_state16) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t i = t; i < n_keys16; i += stride) keys16[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (int64_t i = t; i < n_state16; i += stride) state16[i] = make_uint4(0u, 0u, 0u, 0u);
}

template <typename PayT>
__device__ __forceinline__ PayT make_payload(uint32_t bag, uint32_t pos) {
  if constexpr (sizeof(PayT) == 4) return bag;
  else return (static_cast<uint64_t>(bag) << 32) | pos;
}
template <typename PayT>
__device__ __forceinline__ uint32_t payload_bag(PayT p) {
  if constexpr (sizeof(PayT) == 4) return p;
  else return static_cast<uint32_t>(p >> 32);
}
template <typename PayT>
__device__ __forceinline__ uint32_t payload_pos(PayT p) {
  if constexpr (sizeof(PayT) == 4) return 0u;  // never used: no per-sample weights, pooled
  else return static_cast<uint32_t>(p);
}

template <typename KeyT, typename PayT>
__global__ __launch_bounds__(256) void bwd_linearize_pooled_kernel(
    const int64_t* __restrict__ indices, const int64_t* __restrict__ offsets,
    const int64_t* __restrict__ feat_rows, const int64_t* __restrict__ feat_row_base,
    const int64_t* __restrict__ feat_window, int F, int B,
    int64_t N, int key_bits, KeyT* __restrict__ keys, PayT* __restrict__ payload,
    int32_t* bounds_errors) {
  const int64_t bag = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bag >= static_cast<int64_t>(F) * B) return;
  const int f = static_cast<int>(bag / B);
  const int64_t s = offsets[bag];
  const int64_t e = offsets[bag + 1];
  if (s < 0 || e > N || s > e) {
    // malformed offsets: never touch memory through them.  The positions such a bag fails to cover keep the
    // all-ones key the caller pre-filled, which the update kernel treats as "no row".
    if (bounds_errors != nullptr) atomicAdd(bounds_errors, 1);
    return;
  }
  const RowWindow win = load_window(feat_rows, feat_window, f);
  const int64_t base = feat_row_base[f];
  const KeyT sentinel = static_cast<KeyT>((key_bits >= 64) ? ~0ull : ((1ull << key_bits) - 1ull));
  int nbad = 0;
  for (int64_t p = s; p < e; ++p) {
    int64_t lidx;
    const int cls = classify_id(win, indices[p], lidx);  // rows of other shards: no key, not an error
    if (cls == kIdBad) ++nbad;
    keys[p] = cls == kIdLocal ? static_cast<KeyT>(base + lidx) : sentinel;
    payload[p] = make_payload<PayT>(static_cast<uint32_t>(bag), static_cast<uint32_t>(p));
  }
  if (nbad > 0 && bounds_errors != nullptr) atomicAdd(bounds_errors, nbad);
}

// PoolingMode.NONE: position p belongs to feature f(p); "bag" is encoded as f*B so that the
// update kernel recovers f with the same division.
template <typename KeyT>
__global__ __launch_bounds__(256) void bwd_linearize_nobag_kernel(
    const int64_t* __restrict__ indices, const int64_t* __restrict__ offsets,
    const int64_t* __restrict__ feat_rows, const int64_t* __restrict__ feat_row_base, int F, int B,
    int64_t N, int key_bits, KeyT* __restrict__ keys, uint64_t* __restrict__ payload,
    int32_t* bounds_errors) {
  extern __shared__ int64_t fb[];
  for (int i = threadIdx.x; i <= F; i += blockDim.x) fb[i] = offsets[static_cast<int64_t>(i) * B];
  __syncthreads();
  const KeyT sentinel = static_cast<KeyT>((key_bits >= 64) ? ~0ull : ((1ull << key_bits) - 1ull));
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < N;
       p += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int lo = 0, hi = F;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (fb[mid] <= p) lo = mid; else hi = mid;
    }
    const int f = lo;
    const int64_t idx = indices[p];
    const bool ok = static_cast<uint64_t>(idx) < static_cast<uint64_t>(feat_rows[f]);
    if (!ok && bounds_errors != nullptr) atomicAdd(bounds_errors, 1);
    keys[p] = ok ? static_cast<KeyT>(feat_row_base[f] + idx) : sentinel;
    payload[p] = (static_cast<uint64_t>(static_cast<int64_t>(f) * B) << 32) | static_cast<uint32_t>(p);
  }
}

__device__ __forceinline__ float4 ldc(const float* row, int d, int D, bool vec) {
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
__device__ __forceinline__ void stc(float* row, int d, int D, bool vec, float4 x) {
  if (vec) {
    st4(row + d, x);
  } else {
    if (d + 0 < D) row[d + 0] = x.x;
    if (d + 1 < D) row[d + 1] = x.y;
    if (d + 2 < D) row[d + 2] = x.z;
    if (d + 3 < D) row[d + 3] = x.w;
  }
}

// Sum over the G lanes of a group (G consecutive lanes), fixed butterfly order.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// Applies the optimizer to one table row.  `g` = coalesced gradient columns held by this lane,
// `w` = current weight columns (pre-loaded).  Group-uniform control flow.
// OPTC >= 0 fixes the optimizer at compile time (smaller live state => more waves per SIMD).
template <int G, int NV, int OPTC = -1, int ABL = 0>
__device__ __forceinline__ void apply_row(const BwdArgs& a, int f, int64_t local_row, int D,
                                          bool vec, int gl, float* wrow, float4 (&w)[NV],
                                          float4 (&g)[NV]) {
  const int optimizer = OPTC >= 0 ? OPTC : a.opt.optimizer;
  const float lr = a.opt.learning_rate;
  if (optimizer == TBE_OPT_EXACT_SGD) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) {
        float4 r;
        r.x = fmaf(-lr, g[v].x, w[v].x);
        r.y = fmaf(-lr, g[v].y, w[v].y);
        r.z = fmaf(-lr, g[v].z, w[v].z);
        r.w = fmaf(-lr, g[v].w, w[v].w);
        if (ABL == 0 || r.x == 1.2345e-31f) stc(wrow, d, D, vec, r);  // ABL: tuning runs without the row write
      }
    }
  } else if (optimizer == TBE_OPT_EXACT_ROWWISE_ADAGRAD) {
    const float wd = a.opt.weight_decay;
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) {
        if (wd != 0.f) {
          g[v].x = fmaf(wd, w[v].x, g[v].x);
          g[v].y = fmaf(wd, w[v].y, g[v].y);
          g[v].z = fmaf(wd, w[v].z, g[v].z);
          g[v].w = fmaf(wd, w[v].w, g[v].w);
        }
        // columns beyond D are zero in g (never loaded), so the vector form is safe
        ss = fmaf(g[v].x, g[v].x, ss);
        ss = fmaf(g[v].y, g[v].y, ss);
        ss = fmaf(g[v].z, g[v].z, ss);
        ss = fmaf(g[v].w, g[v].w, ss);
      }
    }
    ss = group_sum<G>(ss);
    float* m = reinterpret_cast<float*>(a.feat_state0[f]) + local_row;
    const float m_new = *m + ss / static_cast<float>(D);
    const float mult = lr / (sqrtf(m_new) + a.opt.eps);
    if (gl == 0) *m = m_new;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) {
        float4 r;
        r.x = fmaf(-mult, g[v].x, w[v].x);
        r.y = fmaf(-mult, g[v].y, w[v].y);
        r.z = fmaf(-mult, g[v].z, w[v].z);
        r.w = fmaf(-mult, g[v].w, w[v].w);
        stc(wrow, d, D, vec, r);
      }
    }
  } else if (optimizer == TBE_OPT_DENSE_GRAD) {
    float* grow = reinterpret_cast<float*>(a.feat_state0[f]) + local_row * D;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) stc(grow, d, D, vec && ((reinterpret_cast<uintptr_t>(grow) & 15) == 0), g[v]);
    }
  } else if (optimizer == TBE_OPT_EXACT_ADAGRAD) {
    float* mrow = reinterpret_cast<float*>(a.feat_state0[f]) + local_row * D;
    const bool mvec = vec && ((reinterpret_cast<uintptr_t>(mrow) & 15) == 0);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) {
        float4 m = ldc(mrow, d, D, mvec);
        m.x = fmaf(g[v].x, g[v].x, m.x);
        m.y = fmaf(g[v].y, g[v].y, m.y);
        m.z = fmaf(g[v].z, g[v].z, m.z);
        m.w = fmaf(g[v].w, g[v].w, m.w);
        stc(mrow, d, D, mvec, m);
        float4 r;
        r.x = w[v].x - lr * g[v].x / (sqrtf(m.x) + a.opt.eps);
        r.y = w[v].y - lr * g[v].y / (sqrtf(m.y) + a.opt.eps);
        r.z = w[v].z - lr * g[v].z / (sqrtf(m.z) + a.opt.eps);
        r.w = w[v].w - lr * g[v].w / (sqrtf(m.w) + a.opt.eps);
        stc(wrow, d, D, vec, r);
      }
    }
  } else if (optimizer == TBE_OPT_ADAM) {
    float* m1row = reinterpret_cast<float*>(a.feat_state0[f]) + local_row * D;
    float* m2row = reinterpret_cast<float*>(a.feat_state1[f]) + local_row * D;
    const bool mvec = vec && ((reinterpret_cast<uintptr_t>(m1row) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(m2row) & 15) == 0);
    const float b1 = a.opt.beta1, b2 = a.opt.beta2, eps = a.opt.eps, wd = a.opt.weight_decay;
#define TBE_ADAM1(c)                                                     \
  m1.c = fmaf(b1, m1.c, (1.f - b1) * g[v].c);                            \
  m2.c = fmaf(b2, m2.c, (1.f - b2) * g[v].c * g[v].c);                   \
  r.c = w[v].c - lr * ((m1.c / a.bias1) / (sqrtf(m2.c / a.bias2) + eps) + wd * w[v].c);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int d = (v * G + gl) * 4;
      if (d < D) {
        float4 m1 = ldc(m1row, d, D, mvec);
        float4 m2 = ldc(m2row, d, D, mvec);
        float4 r;
        TBE_ADAM1(x) TBE_ADAM1(y) TBE_ADAM1(z) TBE_ADAM1(w)
        stc(m1row, d, D, mvec, m1);
        stc(m2row, d, D, mvec, m2);
        stc(wrow, d, D, vec, r);
      }
    }
#undef TBE_ADAM1
  }
}

template <int NV>
struct BwdUnroll {
  static constexpr int U = NV == 1 ? 4 : (NV == 2 ? 2 : 1);
};

// FAST: every feature has dim a.fast_D (multiple of 4), SUM pooling, no per-sample weights, every
// row base 16-B aligned (TBE_FLAG_UNIFORM_ALIGNED from the host) — the Criteo configuration.
template <typename KeyT, typename PayT, int G, int NV, int OPTC, bool FAST, int U, int MINW, int ABL = 0>
__global__ __launch_bounds__(256, MINW) void bwd_update_kernel(BwdArgs a) {
  constexpr int NG = kWave / G;
  const int lane = threadIdx.x & 63;
  const int g = lane / G;
  const int gl = lane % G;
  const int gbase = g * G;  // first lane of this group
  const int64_t nchunks = (a.N + a.C - 1) / a.C;
  const int64_t chunk = (static_cast<int64_t>(blockIdx.x) * (blockDim.x / kWave) + (threadIdx.x >> 6)) * NG + g;
  // Whole wave out of range -> exit; a partially filled wave keeps its idle groups alive
  // (they execute the shuffles with in-range flags false).
  const int64_t chunk_w0 = chunk - g;
  if (chunk_w0 >= nchunks) return;
  const bool active = chunk < nchunks;

  const KeyT* __restrict__ skey = static_cast<const KeyT*>(a.keys_sorted);
  const PayT* __restrict__ spay = static_cast<const PayT*>(a.payload_sorted);
  const KeyT sentinel = static_cast<KeyT>((a.key_bits >= 64) ? ~0ull : ((1ull << a.key_bits) - 1ull));
  const bool nobag = a.pooling_mode == TBE_POOL_NONE;
  const bool mean = a.pooling_mode == TBE_POOL_MEAN;

  const int64_t i0 = active ? chunk * a.C : 0;
  const int64_t i1 = active ? min(a.N, i0 + static_cast<int64_t>(a.C)) : 0;
  bool started_here = true;
  if (active && i0 > 0) started_here = skey[i0 - 1] != skey[i0];
  bool tail_open = false;  // last processed contribution did not end its run
  int nrows = 0;           // table rows this group finished (profiling counter)

  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);

  // The loop trip count must be wave-uniform for the shuffles: use the max span over groups.
  const int span = static_cast<int>(i1 - i0);
  int max_span = span;
#pragma unroll
  for (int o = 32; o >= G; o >>= 1) max_span = max(max_span, __shfl_xor(max_span, o, kWave));

  for (int sb = 0; sb < max_span; sb += G) {
    const int64_t kk = i0 + sb + gl;
    const bool in = active && kk < i1;
    KeyT key_k = sentinel;
    KeyT keyn_k = sentinel;
    PayT pay_k = 0;
    if (in) {
      key_k = skey[kk];
      pay_k = spay[kk];
      if (kk + 1 < a.N) keyn_k = skey[kk + 1];
    }
    const bool valid_k = in && key_k < sentinel;  // sentinel = invalid id; anything above = never written
    const bool last_k = valid_k && (key_k != keyn_k || kk + 1 >= a.N);
    const uint32_t bag_k = payload_bag<PayT>(pay_k);
    const uint32_t pos_k = payload_pos<PayT>(pay_k);
    const int f_k = valid_k ? static_cast<int>(bag_k / static_cast<uint32_t>(a.B)) : 0;
    const int b_k = static_cast<int>(bag_k - static_cast<uint32_t>(f_k) * static_cast<uint32_t>(a.B));
    const int D_k = FAST ? a.fast_D : a.feat_D[f_k];
    float w_k = 1.f;
    const float* gptr_k = a.grad_out;
    const float* wptr_k = nullptr;
    int64_t lrow_k = 0;
    if (valid_k) {
      if (!FAST) {
        if (a.psw != nullptr) w_k = a.psw[pos_k];
        if (mean && (a.feat_pooling == nullptr || a.feat_pooling[f_k] == TBE_POOL_MEAN)) {
          const int64_t len = a.offsets[bag_k + 1] - a.offsets[bag_k];
          w_k = w_k / static_cast<float>(len);
        }
      }
      gptr_k = (!FAST && nobag) ? a.grad_out + static_cast<int64_t>(pos_k) * a.grad_stride
                     : a.grad_out + static_cast<int64_t>(b_k) * a.grad_stride + a.feat_out_offset[f_k];
      lrow_k = static_cast<int64_t>(key_k) - a.feat_row_base[f_k];
      wptr_k = reinterpret_cast<const float*>(a.feat_weights[f_k]) + lrow_k * D_k;
    }
    const int n = active ? static_cast<int>(min<int64_t>(G, i1 - (i0 + sb))) : 0;
    int max_n = n;
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) max_n = max(max_n, __shfl_xor(max_n, o, kWave));

    for (int j = 0; j < max_n; j += U) {
      float4 x[U][NV];
      float4 wr[U][NV];
      float wt[U];
      bool val[U], lst[U];
      const float* wp[U];
      int Du[U], fu[U];
      int64_t lrow[U];
      bool vecu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = gbase + ((j + u) & (G - 1));
        const bool inb = (j + u) < n;
        val[u] = inb && (__shfl(static_cast<int>(valid_k), src, kWave) != 0);
        lst[u] = inb && (__shfl(static_cast<int>(last_k), src, kWave) != 0);
        wt[u] = FAST ? 1.f : __shfl(w_k, src, kWave);
        const float* gp = reinterpret_cast<const float*>(shflu64(reinterpret_cast<uint64_t>(gptr_k), src));
        wp[u] = reinterpret_cast<const float*>(shflu64(reinterpret_cast<uint64_t>(wptr_k), src));
        Du[u] = FAST ? a.fast_D : __shfl(D_k, src, kWave);
        fu[u] = __shfl(f_k, src, kWave);
        lrow[u] = shfl64(lrow_k, src);
        const bool gvec = FAST || (((Du[u] & 3) == 0) && ((reinterpret_cast<uintptr_t>(gp) & 15) == 0));
        vecu[u] = FAST || (((Du[u] & 3) == 0) && ((reinterpret_cast<uintptr_t>(wp[u]) & 15) == 0));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          x[u][v] = (val[u] && d < Du[u]) ? ldc(gp, d, Du[u], gvec) : make_float4(0.f, 0.f, 0.f, 0.f);
          // the current weight row (not needed when the coalesced gradient is only written out: DENSE_GRAD)
          wr[u][v] = (ABL != 2 && OPTC != TBE_OPT_DENSE_GRAD && lst[u] && d < Du[u]) ? ldc(wp[u], d, Du[u], vecu[u])
                                                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (val[u]) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc[v].x = fmaf(wt[u], x[u][v].x, acc[v].x);
            acc[v].y = fmaf(wt[u], x[u][v].y, acc[v].y);
            acc[v].z = fmaf(wt[u], x[u][v].z, acc[v].z);
            acc[v].w = fmaf(wt[u], x[u][v].w, acc[v].w);
          }
          tail_open = !lst[u];
          if (lst[u]) {
            if (started_here) {
              ++nrows;
              apply_row<G, NV, OPTC, ABL>(a, fu[u], lrow[u], Du[u], vecu[u], gl, const_cast<float*>(wp[u]), wr[u], acc);
            } else {
              float* pf = a.partial_first + chunk * a.max_D_pad;
#pragma unroll
              for (int v = 0; v < NV; ++v) {
                const int d = (v * G + gl) * 4;
                if (d < Du[u]) st4(pf + d, acc[v]);
              }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            started_here = true;
          }
        }
      }
    }
  }
  if (active) {
    int is_origin = 0;
    if (tail_open) {
      float* dst = started_here ? a.partial_last + chunk * a.max_D_pad : a.partial_first + chunk * a.max_D_pad;
      is_origin = started_here ? 1 : 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int d = (v * G + gl) * 4;
        if (d < a.max_D_pad) st4(dst + d, acc[v]);
      }
    }
    if (gl == 0 && is_origin) a.origin_list[atomicAdd(a.origin_count, 1)] = static_cast<int32_t>(chunk);
    nrows += is_origin;
  }
  if (a.unique_rows != nullptr) {  // profiling only: one add per WAVE, spread over kProfileRowSlots cache lines
    int wave_rows = (active && gl == 0) ? nrows : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wave_rows += __shfl_xor(wave_rows, o, kWave);
    if (lane == 0 && wave_rows > 0)
      atomicAdd(a.unique_rows + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % kProfileRowSlots) * 16,
                static_cast<unsigned long long>(wave_rows));
  }
}

// Fix-up: every "origin" chunk owns a run that continues into the following chunks.  One wave
// per origin: the wave gallops over the following chunk heads (64 per step) to find the chain
// length, its NG groups sum contiguous halves of the chain's partial rows (4 loads in flight)
// and the halves are combined through LDS in fixed order.  Chains longer than kLongChain
// (rows of tiny tables that receive thousands of contributions) are summed by the whole
// workgroup: 4*NG groups, LDS-staged partials, fixed combine order => bitwise reproducible.
constexpr int kLongChain = 24;

template <typename KeyT, int G, int NV>
struct FixupRow {
  int f;
  int D;
  int64_t lrow;
  float* wrow;
  bool vec;
};

template <int G, int NV>
__device__ __forceinline__ void sum_partials(const float* __restrict__ base, int max_D_pad, int64_t first,
                                             int begin, int end, int D, int gl, float4 (&acc)[NV]) {
  constexpr int U = 4;
  for (int j = begin; j < end; j += U) {
    float4 x[U][NV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* pf = base + (first + j + u) * max_D_pad;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int d = (v * G + gl) * 4;
        x[u][v] = (j + u < end && d < D) ? ld4(pf + d) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        acc[v].x += x[u][v].x;
        acc[v].y += x[u][v].y;
        acc[v].z += x[u][v].z;
        acc[v].w += x[u][v].w;
      }
  }
}

template <typename KeyT, typename PayT, int G, int NV>
__global__ __launch_bounds__(256) void bwd_fixup_kernel(BwdArgs a) {
  constexpr int NG = kWave / G;
  constexpr int NGB = 4 * NG;
  __shared__ float4 part[NGB][NV][G];
  __shared__ int64_t long_chunk[4];
  __shared__ int long_len[4];
  __shared__ int n_long;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / G;
  const int gl = lane % G;
  const int q = wave * NG + g;  // group index inside the block
  const int64_t nchunks = (a.N + a.C - 1) / a.C;
  const KeyT* __restrict__ skey = static_cast<const KeyT*>(a.keys_sorted);
  const PayT* __restrict__ spay = static_cast<const PayT*>(a.payload_sorted);
  const int count = *a.origin_count;

  for (int base = blockIdx.x * 4; base < count; base += gridDim.x * 4) {  // block-uniform
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    const int oi = base + wave;
    if (oi < count) {  // wave-uniform
      const int64_t chunk = a.origin_list[oi];
      const KeyT key_run = skey[(chunk + 1) * a.C - 1];
      int len = 0;
      bool more = true;
      while (more) {
        const int64_t cc = chunk + 1 + len + lane;
        const bool ok = cc < nchunks && skey[cc * a.C] == key_run;
        const unsigned long long m = __ballot(ok);
        const int lead = (m == ~0ull) ? 64 : __builtin_ctzll(~m);
        len += lead;
        more = lead == 64;
      }
      if (len > kLongChain) {
        if (lane == 0) {
          const int s = atomicAdd(&n_long, 1);
          long_chunk[s] = chunk;
          long_len[s] = len;
        }
      } else {
        const PayT pay = spay[(chunk + 1) * a.C - 1];
        const int f = static_cast<int>(payload_bag<PayT>(pay) / static_cast<uint32_t>(a.B));
        const int D = a.feat_D[f];
        const int64_t lrow = static_cast<int64_t>(key_run) - a.feat_row_base[f];
        float* wrow = reinterpret_cast<float*>(a.feat_weights[f]) + lrow * D;
        const bool vec = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(wrow) & 15) == 0);
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int per = (len + NG - 1) / NG;
        sum_partials<G, NV>(a.partial_first, a.max_D_pad, chunk + 1, min(len, g * per), min(len, (g + 1) * per), D, gl, acc);
        if (NG > 1) {
#pragma unroll
          for (int v = 0; v < NV; ++v) part[q][v][gl] = acc[v];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (g == 0) {
          float4 tot[NV], w[NV];
          const float* pl = a.partial_last + chunk * a.max_D_pad;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const int d = (v * G + gl) * 4;
            tot[v] = (d < D) ? ld4(pl + d) : make_float4(0.f, 0.f, 0.f, 0.f);
            w[v] = (d < D) ? ldc(wrow, d, D, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            tot[v].x += acc[v].x;
            tot[v].y += acc[v].y;
            tot[v].z += acc[v].z;
            tot[v].w += acc[v].w;
            for (int og = 1; og < NG; ++og) {
              const float4 o = part[wave * NG + og][v][gl];
              tot[v].x += o.x;
              tot[v].y += o.y;
              tot[v].z += o.z;
              tot[v].w += o.w;
            }
          }
          apply_row<G, NV>(a, f, lrow, D, vec, gl, wrow, w, tot);
        }
      }
    }
    __syncthreads();
    const int nl = n_long;
    for (int s = 0; s < nl; ++s) {  // block-uniform
      const int64_t chunk = long_chunk[s];
      const int len = long_len[s];
      const KeyT key_run = skey[(chunk + 1) * a.C - 1];
      const PayT pay = spay[(chunk + 1) * a.C - 1];
      const int f = static_cast<int>(payload_bag<PayT>(pay) / static_cast<uint32_t>(a.B));
      const int D = a.feat_D[f];
      const int64_t lrow = static_cast<int64_t>(key_run) - a.feat_row_base[f];
      float* wrow = reinterpret_cast<float*>(a.feat_weights[f]) + lrow * D;
      const bool vec = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(wrow) & 15) == 0);
      float4 acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int per = (len + NGB - 1) / NGB;
      sum_partials<G, NV>(a.partial_first, a.max_D_pad, chunk + 1, min(len, q * per), min(len, (q + 1) * per), D, gl, acc);
      __syncthreads();  // previous iteration's readers are done with `part`
#pragma unroll
      for (int v = 0; v < NV; ++v) part[q][v][gl] = acc[v];
      __syncthreads();
      if (q == 0) {
        float4 tot[NV], w[NV];
        const float* pl = a.partial_last + chunk * a.max_D_pad;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int d = (v * G + gl) * 4;
          tot[v] = (d < D) ? ld4(pl + d) : make_float4(0.f, 0.f, 0.f, 0.f);
          w[v] = (d < D) ? ldc(wrow, d, D, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
          for (int oq = 0; oq < NGB; ++oq) {
            const float4 o = part[oq][v][gl];
            tot[v].x += o.x;
            tot[v].y += o.y;
            tot[v].z += o.z;
            tot[v].w += o.w;
          }
        }
        apply_row<G, NV>(a, f, lrow, D, vec, gl, wrow, w, tot);
      }
    }
    __syncthreads();
  }
}

static int pick_chunk(int64_t N) {
  static const int forced = [] {
    const char* e = getenv("TBE_BWD_CHUNK");  // tuning knob (multiple of 8 in [8, 1024])
    const int v = e ? atoi(e) : 0;
    return (v >= 8 && v <= 1024 && v % 8 == 0) ? v : 0;
  }();
  if (forced) return forced;
  int64_t c = (N + 16383) / 16384;
  c = (c + 31) / 32 * 32;
  if (c < 32) c = 32;
  if (c > 256) c = 256;
  return static_cast<int>(c);
}

struct BwdWorkspace {
  void* keys_in;
  void* keys_out;
  void* pay_in;   // uint32 or uint64 payloads (sized for uint64)
  void* pay_out;
  float* partial_first;
  float* partial_last;
  int32_t* origin_list;
  int32_t* origin_count;
  RadixWorkspace sort;
  size_t total;
};

static int carve(void* ws, int64_t N, int32_t max_D, int32_t key_bits, BwdWorkspace* out) {
  const bool k64 = key_bits > 32;
  const size_t ksz = k64 ? 8 : 4;
  const int C = pick_chunk(N);
  const int64_t nchunks = (N + C - 1) / C;
  const int max_D_pad = (max_D + 3) / 4 * 4;
  const size_t sort_bytes = radix_carve(nullptr, N, key_bits).bytes;
  Carver c(ws);
  out->keys_in = c.take_bytes(N * ksz);
  out->keys_out = c.take_bytes(N * ksz);
  out->pay_in = c.take_bytes(N * sizeof(uint64_t));
  out->pay_out = c.take_bytes(N * sizeof(uint64_t));
  out->partial_first = c.take<float>(nchunks * max_D_pad);
  out->partial_last = c.take<float>(nchunks * max_D_pad);
  out->origin_list = c.take<int32_t>(nchunks);
  out->sort = radix_carve(c.take_bytes(sort_bytes), N, key_bits);
  // zeroed together with the sort's tickets by the one memset radix_sort_pairs issues
  out->origin_count = out->sort.tickets ? reinterpret_cast<int32_t*>(out->sort.tickets + 8) : nullptr;
  out->total = c.total();
  return TBE_OK;
}

template <typename KeyT, typename PayT, int G, int NV>
static int launch_update(const BwdArgs& a, hipStream_t st) {
  constexpr int NG = kWave / G;
  const int64_t nchunks = (a.N + a.C - 1) / a.C;
  const int64_t groups_per_block = 4 * NG;
  const unsigned grid = static_cast<unsigned>((nchunks + groups_per_block - 1) / groups_per_block);
  {
    ProfileSpan span(TBE_PROFILE_BWD_UPDATE_KERNEL, st);
    constexpr int UG = BwdUnroll<NV>::U;
    static const int variant = [] {
      const char* e = getenv("TBE_BWD_VARIANT");
      return e ? atoi(e) : 0;
    }();
    const bool fast = a.fast_D > 0 && a.pooling_mode == TBE_POOL_SUM && a.psw == nullptr;
    const int oc = a.opt.optimizer;
#define TBE_UPD(OPTC, FAST_, UU, MW) \
  hipLaunchKernelGGL((bwd_update_kernel<KeyT, PayT, G, NV, OPTC, FAST_, UU, MW>), dim3(grid), dim3(256), 0, st, a)
    if (fast && G == 32 && NV == 1 && oc == TBE_OPT_EXACT_SGD) {
      switch (variant) {
        case 1: TBE_UPD(TBE_OPT_EXACT_SGD, true, 4, 8); break;
        case 2: TBE_UPD(TBE_OPT_EXACT_SGD, true, 2, 8); break;
        case 3: TBE_UPD(TBE_OPT_EXACT_SGD, true, 8, 4); break;
        case 4: TBE_UPD(TBE_OPT_EXACT_SGD, true, 2, 4); break;
        case 5:  // tuning: no row write
          hipLaunchKernelGGL((bwd_update_kernel<KeyT, PayT, G, NV, TBE_OPT_EXACT_SGD, true, 4, 4, 1>), dim3(grid), dim3(256), 0, st, a);
          break;
        case 6:  // tuning: no row read, no row write (gradient streaming only)
          hipLaunchKernelGGL((bwd_update_kernel<KeyT, PayT, G, NV, TBE_OPT_EXACT_SGD, true, 4, 4, 2>), dim3(grid), dim3(256), 0, st, a);
          break;
        default: TBE_UPD(TBE_OPT_EXACT_SGD, true, 4, 4); break;
      }
    } else if (fast && G == 32 && NV == 1 && oc == TBE_OPT_EXACT_ROWWISE_ADAGRAD) {
      TBE_UPD(TBE_OPT_EXACT_ROWWISE_ADAGRAD, true, 4, 4);
    } else if (fast && G == 32 && NV == 1 && oc == TBE_OPT_DENSE_GRAD) {
      TBE_UPD(TBE_OPT_DENSE_GRAD, true, 4, 4);  // the replicated tiny tables of a sharded collection (dense gradient)
    } else {
      TBE_UPD(-1, false, UG, 1);
    }
#undef TBE_UPD
  }
  TBE_CHECK_LAUNCH("tbe_backward update");
  const unsigned fgrid = static_cast<unsigned>(std::min<int64_t>((nchunks + 3) / 4, 1024));
  hipLaunchKernelGGL((bwd_fixup_kernel<KeyT, PayT, G, NV>), dim3(fgrid), dim3(256), 0, st, a);
  TBE_CHECK_LAUNCH("tbe_backward fixup");
  return TBE_OK;
}

constexpr int kPhasePrepare = 1;  // gradient-independent: linearize + sort
constexpr int kPhaseApply = 2;    // update + fix-up

template <typename KeyT, typename PayT>
static int run_backward(BwdArgs a, const BwdWorkspace& w, int32_t max_D, hipStream_t st, int phase) {
  ProfileSpan total_span(phase == kPhasePrepare ? -1 : TBE_PROFILE_BWD_TOTAL, st);
  KeyT* kin = static_cast<KeyT*>(w.keys_in);
  KeyT* kout = static_cast<KeyT*>(w.keys_out);
  PayT* pin = static_cast<PayT*>(w.pay_in);
  PayT* pout = static_cast<PayT*>(w.pay_out);
  // the sort ping-pongs between the two buffer pairs: an odd number of passes ends in the second
  const bool in_second = (radix_passes(a.key_bits) & 1) != 0;
  a.keys_sorted = in_second ? static_cast<void*>(kout) : static_cast<void*>(kin);
  a.payload_sorted = in_second ? static_cast<void*>(pout) : static_cast<void*>(pin);
  if (phase & kPhasePrepare) {
    ProfileSpan prep_span(TBE_PROFILE_BWD_PREPARE, st);
    {
      // 16-B units; both buffers start 256-B aligned and the carver pads each to the next 256-B boundary
      const int64_t n_keys16 = a.pooling_mode == TBE_POOL_NONE ? 0 : (a.N * static_cast<int64_t>(sizeof(KeyT)) + 15) / 16;
      const int64_t n_state16 = (static_cast<int64_t>(radix_state_words(a.N, a.key_bits, sizeof(KeyT) + sizeof(PayT))) + 3) / 4;
      const unsigned grid = static_cast<unsigned>(std::min<int64_t>((std::max(n_keys16, n_state16) + 255) / 256, 2048));
      hipLaunchKernelGGL(bwd_fill_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint4*>(kin), n_keys16,
                         reinterpret_cast<uint4*>(w.sort.state), n_state16);
    }
    if (a.pooling_mode == TBE_POOL_NONE) {
      if constexpr (sizeof(PayT) == 8) {
        const size_t lds = (static_cast<size_t>(a.F) + 1) * sizeof(int64_t);
        const unsigned grid = static_cast<unsigned>(std::min<int64_t>((a.N + 255) / 256, 256 * 16));
        hipLaunchKernelGGL((bwd_linearize_nobag_kernel<KeyT>), dim3(grid), dim3(256), lds, st, a.indices, a.offsets,
                           a.feat_rows, a.feat_row_base, a.F, a.B, a.N, a.key_bits, kin, pin, a.bounds_errors);
      }
    } else {
      const int64_t nbags = static_cast<int64_t>(a.F) * a.B;
      const unsigned grid = static_cast<unsigned>((nbags + 255) / 256);
      hipLaunchKernelGGL((bwd_linearize_pooled_kernel<KeyT, PayT>), dim3(grid), dim3(256), 0, st, a.indices, a.offsets,
                         a.feat_rows, a.feat_row_base, a.feat_window, a.F, a.B, a.N, a.key_bits, kin, pin, a.bounds_errors);
    }
    TBE_CHECK_LAUNCH("tbe_backward linearize");
    const int where = radix_sort_pairs<KeyT, PayT>(kin, kout, pin, pout, a.N, a.key_bits, w.sort, st, kSortStateZeroed);
    if (where < 0) return where;
  }  // prepare
  if (!(phase & kPhaseApply)) return TBE_OK;
  if (max_D <= 64) return launch_update<KeyT, PayT, 16, 1>(a, st);
  if (max_D <= 128) return launch_update<KeyT, PayT, 32, 1>(a, st);
  if (max_D <= 256) return launch_update<KeyT, PayT, 64, 1>(a, st);
  if (max_D <= 512) return launch_update<KeyT, PayT, 64, 2>(a, st);
  if (max_D <= 1024) return launch_update<KeyT, PayT, 64, 4>(a, st);
  return launch_update<KeyT, PayT, 64, 8>(a, st);
}

}  // namespace tbe

using namespace tbe;

extern "C" size_t tbe_backward_workspace_bytes(int64_t N, int32_t F, int32_t B, int32_t max_D,
                                               int32_t key_bits) {
  (void)F;
  (void)B;
  if (N <= 0) return 256;
  if (N >= kSortMaxPairs) return 0;  // not sortable in one call (the backward entry points say why)
  BwdWorkspace w;
  if (carve(nullptr, N, max_D, key_bits, &w) != TBE_OK) return 0;
  return w.total;
}

static int backward_entry(
    const uint64_t* feat_weights, const int32_t* feat_D, const int64_t* feat_out_offset,
    const int64_t* feat_rows, const int64_t* feat_row_base, const uint64_t* feat_state0,
    const uint64_t* feat_state1, int32_t F, int32_t B, int32_t max_D,
    int32_t key_bits, const int64_t* indices, int64_t N, const int64_t* offsets,
    const float* per_sample_weights, int32_t pooling_mode, const int32_t* feat_pooling, const float* grad_out,
    int64_t grad_row_stride, tbe_optimizer_args opt, int32_t flags, void* workspace,
    size_t workspace_bytes, int32_t* bounds_errors, const int64_t* feat_window, void* stream, int phase) {
  TBE_REQUIRE(F > 0 && B >= 0 && N >= 0, "tbe_backward_fused_f32: bad sizes");
  if (phase == kPhasePrepare) {  // gradient / optimizer arguments are not used by this phase
    grad_row_stride = 1;
    opt.optimizer = TBE_OPT_EXACT_SGD;
  }
  TBE_REQUIRE(max_D > 0 && max_D <= 2048, "tbe_backward_fused_f32: max_D=%d outside (0, 2048]", max_D);
  TBE_REQUIRE(key_bits >= 1 && key_bits <= 64, "tbe_backward_fused_f32: key_bits=%d", key_bits);
  TBE_REQUIRE(pooling_mode == TBE_POOL_SUM || pooling_mode == TBE_POOL_MEAN || pooling_mode == TBE_POOL_NONE,
              "tbe_backward_fused_f32: pooling_mode %d", pooling_mode);
  TBE_REQUIRE(static_cast<int64_t>(F) * B < (1ll << 32), "tbe_backward_fused_f32: F*B must be < 2^32");
  // the pair sort's histogram words hold {pass tag | count} with a 29-bit count (radix_sort.hpp)
  TBE_REQUIRE(N < kSortMaxPairs, "tbe_backward_fused_f32: N = %lld ids in one call; the limit is 2^29 - 1 (split the batch)",
              static_cast<long long>(N));
  TBE_REQUIRE(grad_row_stride > 0, "tbe_backward_fused_f32: grad_row_stride <= 0");
  if (phase == (kPhasePrepare | kPhaseApply) && per_sample_weights != nullptr) flags |= TBE_FLAG_WEIGHTED;
  TBE_REQUIRE(per_sample_weights == nullptr || (flags & TBE_FLAG_WEIGHTED) != 0,
              "tbe_backward_apply_f32: per_sample_weights given but TBE_FLAG_WEIGHTED not set (it must be set in "
              "both tbe_backward_prepare and tbe_backward_apply_f32)");
  switch (opt.optimizer) {
    case TBE_OPT_EXACT_SGD:
      break;
    case TBE_OPT_EXACT_ROWWISE_ADAGRAD:
    case TBE_OPT_EXACT_ADAGRAD:
    case TBE_OPT_DENSE_GRAD:
      TBE_REQUIRE(feat_state0 != nullptr, "tbe_backward_fused_f32: optimizer %d needs feat_state0", opt.optimizer);
      break;
    case TBE_OPT_ADAM:
      TBE_REQUIRE(feat_state0 != nullptr && feat_state1 != nullptr, "tbe_backward_fused_f32: ADAM needs two states");
      TBE_REQUIRE(opt.iteration >= 1, "tbe_backward_fused_f32: ADAM iteration must be >= 1");
      break;
    default:
      set_error("tbe_backward_fused_f32: unknown optimizer %d", opt.optimizer);
      return TBE_ERR_UNSUPPORTED;
  }
  if (N == 0 || B == 0) return TBE_OK;
  TBE_REQUIRE(feat_rows && feat_row_base && indices && offsets && workspace, "tbe_backward: null pointer");
  if (phase & kPhaseApply)
    TBE_REQUIRE(feat_weights && feat_D && feat_out_offset && grad_out, "tbe_backward_fused_f32: null pointer");
  TBE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "tbe_backward_fused_f32: workspace must be 256-B aligned");
  BwdWorkspace w;
  int rc = carve(workspace, N, max_D, key_bits, &w);
  if (rc != TBE_OK) return rc;
  if (w.total > workspace_bytes) {
    set_error("tbe_backward_fused_f32: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    return TBE_ERR_WORKSPACE;
  }
  BwdArgs a{};
  a.feat_weights = feat_weights;
  a.feat_D = feat_D;
  a.feat_out_offset = feat_out_offset;
  a.feat_rows = feat_rows;
  a.feat_row_base = feat_row_base;
  a.feat_window = feat_window;
  a.feat_pooling = feat_pooling;
  a.feat_state0 = feat_state0;
  a.feat_state1 = feat_state1;
  a.indices = indices;
  a.offsets = offsets;
  a.psw = per_sample_weights;
  a.grad_out = grad_out;
  a.grad_stride = grad_row_stride;
  a.N = N;
  a.F = F;
  a.B = B;
  a.pooling_mode = pooling_mode;
  a.key_bits = key_bits;
  a.C = pick_chunk(N);
  a.opt = opt;
  a.bias1 = 1.f;
  a.bias2 = 1.f;
  if (opt.optimizer == TBE_OPT_ADAM) {
    a.bias1 = 1.f - powf(opt.beta1, static_cast<float>(opt.iteration));
    a.bias2 = 1.f - powf(opt.beta2, static_cast<float>(opt.iteration));
  }
  a.partial_first = w.partial_first;
  a.partial_last = w.partial_last;
  a.origin_list = w.origin_list;
  a.origin_count = w.origin_count;
  a.max_D_pad = (max_D + 3) / 4 * 4;
  a.fast_D = ((flags & TBE_FLAG_UNIFORM_ALIGNED) && max_D % 4 == 0 && grad_row_stride % 4 == 0 &&
              (reinterpret_cast<uintptr_t>(grad_out) & 15) == 0) ? max_D : 0;
  a.bounds_errors = bounds_errors;
  a.unique_rows = (phase & kPhaseApply) ? profile_unique_rows_counter() : nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // payload width: the bag number alone unless positions are needed (per-sample weights, unpooled rows)
  static const bool force_wide = getenv("TBE_BWD_WIDE_PAYLOAD") != nullptr;  // development A/B switch
  const bool wide = force_wide || pooling_mode == TBE_POOL_NONE || (flags & TBE_FLAG_WEIGHTED) != 0;
  if (key_bits > 32)
    return wide ? run_backward<uint64_t, uint64_t>(a, w, max_D, st, phase) : run_backward<uint64_t, uint32_t>(a, w, max_D, st, phase);
  return wide ? run_backward<uint32_t, uint64_t>(a, w, max_D, st, phase) : run_backward<uint32_t, uint32_t>(a, w, max_D, st, phase);
}

extern "C" int tbe_backward_fused_f32(
    const uint64_t* feat_weights, const int32_t* feat_D, const int64_t* feat_out_offset,
    const int64_t* feat_rows, const int64_t* feat_row_base, const uint64_t* feat_state0,
    const uint64_t* feat_state1, int32_t F, int32_t B, int32_t max_D,
    int32_t key_bits, const int64_t* indices, int64_t N, const int64_t* offsets,
    const float* per_sample_weights, int32_t pooling_mode, const int32_t* feat_pooling, const float* grad_out,
    int64_t grad_row_stride, tbe_optimizer_args opt, int32_t flags, void* workspace,
    size_t workspace_bytes, int32_t* bounds_errors, const int64_t* feat_window, void* stream) {
  return backward_entry(feat_weights, feat_D, feat_out_offset, feat_rows, feat_row_base, feat_state0, feat_state1, F,
                        B, max_D, key_bits, indices, N, offsets, per_sample_weights, pooling_mode, feat_pooling, grad_out,
                        grad_row_stride, opt, flags, workspace, workspace_bytes, bounds_errors, feat_window, stream,
                        kPhasePrepare | kPhaseApply);
}

extern "C" int tbe_backward_prepare(const int64_t* feat_rows, const int64_t* feat_row_base, int32_t F, int32_t B,
                                    int32_t max_D, int32_t key_bits, const int64_t* indices, int64_t N,
                                    const int64_t* offsets, int32_t pooling_mode, int32_t flags, void* workspace,
                                    size_t workspace_bytes, int32_t* bounds_errors, const int64_t* feat_window,
                                    void* stream) {
  tbe_optimizer_args opt{};
  return backward_entry(nullptr, nullptr, nullptr, feat_rows, feat_row_base, nullptr, nullptr, F, B, max_D, key_bits,
                        indices, N, offsets, nullptr, pooling_mode, nullptr, nullptr, 1, opt, flags & TBE_FLAG_WEIGHTED, workspace,
                        workspace_bytes, bounds_errors, feat_window, stream, kPhasePrepare);
}

extern "C" int tbe_backward_apply_f32(
    const uint64_t* feat_weights, const int32_t* feat_D, const int64_t* feat_out_offset,
    const int64_t* feat_rows, const int64_t* feat_row_base, const uint64_t* feat_state0,
    const uint64_t* feat_state1, int32_t F, int32_t B, int32_t max_D,
    int32_t key_bits, const int64_t* indices, int64_t N, const int64_t* offsets,
    const float* per_sample_weights, int32_t pooling_mode, const int32_t* feat_pooling, const float* grad_out,
    int64_t grad_row_stride, tbe_optimizer_args opt, int32_t flags, void* workspace,
    size_t workspace_bytes, void* stream) {
  return backward_entry(feat_weights, feat_D, feat_out_offset, feat_rows, feat_row_base, feat_state0, feat_state1, F,
                        B, max_D, key_bits, indices, N, offsets, per_sample_weights, pooling_mode, feat_pooling, grad_out,
                        grad_row_stride, opt, flags, workspace, workspace_bytes, nullptr, nullptr, stream, kPhaseApply);
}

// ---- the pair sort as a public entry (tests, micro-benchmarks) ---------------------------------------
extern "C" size_t tbe_sort_pairs_workspace_bytes(int64_t n, int32_t key_bits) {
  if (key_bits < 1 || key_bits > 64) return 0;
  return radix_carve(nullptr, n, key_bits).bytes;
}

extern "C" int tbe_sort_pairs(void* keys, void* keys_tmp, void* payload, void* payload_tmp, int64_t n, int32_t key_bits,
                              int32_t key_bytes, int32_t payload_bytes, void* workspace, size_t workspace_bytes,
                              void* stream) {
  TBE_REQUIRE(n >= 0, "tbe_sort_pairs: n < 0");
  TBE_REQUIRE(key_bytes == 4 || key_bytes == 8, "tbe_sort_pairs: key_bytes must be 4 or 8");
  TBE_REQUIRE(payload_bytes == 4 || payload_bytes == 8, "tbe_sort_pairs: payload_bytes must be 4 or 8");
  TBE_REQUIRE(key_bits >= 1 && key_bits <= 8 * key_bytes, "tbe_sort_pairs: key_bits=%d", key_bits);
  if (n == 0) return TBE_OK;
  TBE_REQUIRE(keys && keys_tmp && payload && payload_tmp && workspace, "tbe_sort_pairs: null pointer");
  TBE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "tbe_sort_pairs: workspace must be 256-B aligned");
  const RadixWorkspace ws = radix_carve(workspace, n, key_bits);
  if (ws.bytes > workspace_bytes) {
    set_error("tbe_sort_pairs: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
    return TBE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  int where;
  if (key_bytes == 4 && payload_bytes == 4)
    where = radix_sort_pairs<uint32_t, uint32_t>(static_cast<uint32_t*>(keys), static_cast<uint32_t*>(keys_tmp),
                                                 static_cast<uint32_t*>(payload), static_cast<uint32_t*>(payload_tmp), n,
                                                 key_bits, ws, st);
  else if (key_bytes == 4)
    where = radix_sort_pairs<uint32_t, uint64_t>(static_cast<uint32_t*>(keys), static_cast<uint32_t*>(keys_tmp),
                                                 static_cast<uint64_t*>(payload), static_cast<uint64_t*>(payload_tmp), n,
                                                 key_bits, ws, st);
  else if (payload_bytes == 4)
    where = radix_sort_pairs<uint64_t, uint32_t>(static_cast<uint64_t*>(keys), static_cast<uint64_t*>(keys_tmp),
                                                 static_cast<uint32_t*>(payload), static_cast<uint32_t*>(payload_tmp), n,
                                                 key_bits, ws, st);
  else
    where = radix_sort_pairs<uint64_t, uint64_t>(static_cast<uint64_t*>(keys), static_cast<uint64_t*>(keys_tmp),
                                                 static_cast<uint64_t*>(payload), static_cast<uint64_t*>(payload_tmp), n,
                                                 key_bits, ws, st);
  if (where < 0) return where;
  if (where == 1) {
    if (hipMemcpyAsync(keys, keys_tmp, static_cast<size_t>(n) * key_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(payload, payload_tmp, static_cast<size_t>(n) * payload_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      set_error("tbe_sort_pairs: copy back failed");
      return TBE_ERR_LAUNCH;
    }
  }
  return TBE_OK;
}

extern "C" int tbe_debug_sort_timeouts(int64_t* count) {
  TBE_REQUIRE(count != nullptr, "tbe_debug_sort_timeouts: null pointer");
  if (hipDeviceSynchronize() != hipSuccess) {
    set_error("tbe_debug_sort_timeouts: hipDeviceSynchronize failed");
    return TBE_ERR_LAUNCH;
  }
  return tbe_fault_status(count);
}

// the device side of a give-up without a sort that hangs: what radix_pass_kernel does when a wait outlives kSpinLimit
__global__ void inject_sort_giveup_kernel(uint32_t* fault) { report_sort_giveup(fault); }
extern "C" int tbe_debug_inject_sort_giveup(void* stream) {
  uint32_t* const fault = fault_word_device();
  TBE_REQUIRE(fault != nullptr, "tbe_debug_inject_sort_giveup: no fault word (no HIP device?)");
  hipLaunchKernelGGL(inject_sort_giveup_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), fault);
  TBE_CHECK_LAUNCH("tbe_debug_inject_sort_giveup");
  return TBE_OK;
}

extern "C" int tbe_debug_set_sort_stamps(void* device_buffer) {
  g_sort_stamps = static_cast<uint64_t*>(device_buffer);
  return TBE_OK;
}

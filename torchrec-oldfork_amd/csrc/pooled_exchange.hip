// Pooled-embedding exchange for mixed table-wise + row-wise sharding on xGMI.
//
// The reference runs one all-to-all per table-wise group (torchrec/distributed/comm_ops.py:462-605,
// sharding/tw_sharding.py:272-309), one ring reduce-scatter per row-wise group
// (comm_ops.py:848-930, sharding/rw_sharding.py:314-341) and then torch.cat's the results
// (embeddingbag.py:212-223).  On MI355X xGMI is point-to-point (7 links), so an all-to-all uses
// every link at once while a ring reduce-scatter is bound by one link.  This build therefore
// sends BOTH kinds through ONE all-to-all of the a2a-ready TBE output [dst][B_local][D_local]
// and finishes on the receiver:
//   table-wise feature : copy the owner's columns                      (all-to-all semantics)
//   row-wise feature   : sum the partial pools of all W source ranks    (reduce-scatter
//                        semantics, fixed rank order 0..W-1 => bitwise reproducible)
// `unpack` does that and writes [B_local, sum D] in the collection's feature order; `pack` is its
// transpose for the gradient (table-wise: route to the owner; row-wise: broadcast to every rank),
// with the 1/W gradient division of comm_ops.py:527-528, :883-885 fused in as `scale`.
#include <algorithm>

#include "common.hpp"

namespace tbe {

struct ExchangeArgs {
  const int32_t* feat_out_col;   // [Fg+1] column of each global feature in the [B_local, D_total] matrix
  const int32_t* feat_src;       // [Fg] owning rank, or -1 for a row-wise feature (all ranks hold partials)
  const int32_t* feat_slab_col;  // [Fg] column of the feature inside the slab(s)
  const int64_t* slab_offset;    // [W] element offset of rank r's slab in the exchange buffer
  const int32_t* slab_stride;    // [W] row stride (= D_local of rank r)
  int32_t Fg;
  int32_t W;
  int32_t B_local;
  int32_t D_total;
  float scale;
};

template <int VEC, bool PACK>
__global__ __launch_bounds__(256) void pooled_exchange_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             ExchangeArgs a) {
  extern __shared__ int32_t lds[];
  int32_t* s_col = lds;                       // [Fg+1]
  int32_t* s_src = s_col + a.Fg + 1;          // [Fg]
  int32_t* s_scol = s_src + a.Fg;             // [Fg]
  int32_t* s_stride = s_scol + a.Fg;          // [W]
  int64_t* s_off = reinterpret_cast<int64_t*>(s_stride + a.W + ((a.Fg * 3 + 1 + a.W) & 1));  // [W], 8-B aligned
  for (int i = threadIdx.x; i <= a.Fg; i += blockDim.x) s_col[i] = a.feat_out_col[i];
  for (int i = threadIdx.x; i < a.Fg; i += blockDim.x) {
    s_src[i] = a.feat_src[i];
    s_scol[i] = a.feat_slab_col[i];
  }
  for (int i = threadIdx.x; i < a.W; i += blockDim.x) {
    s_stride[i] = a.slab_stride[i];
    s_off[i] = a.slab_offset[i];
  }
  __syncthreads();
  const int cols = a.D_total / VEC;
  const int64_t total = static_cast<int64_t>(a.B_local) * cols;
  // PACK: in = grad matrix, out = send slabs.  UNPACK: in = recv slabs, out = matrix.
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / cols);
    const int d = static_cast<int>(i - static_cast<int64_t>(b) * cols) * VEC;
    int lo = 0, hi = a.Fg;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_col[mid] <= d) lo = mid; else hi = mid;
    }
    const int src = s_src[lo];
    if (src < -1) continue;  // data-parallel (replicated) feature: not part of the exchange
    const int within = s_scol[lo] + (d - s_col[lo]);
    const int64_t mat = static_cast<int64_t>(b) * a.D_total + d;
    if (PACK) {
      if (VEC == 4) {
        float4 v = ld4(in + mat);
        v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
        if (src >= 0) {
          st4(out + s_off[src] + static_cast<int64_t>(b) * s_stride[src] + within, v);
        } else {
          for (int r = 0; r < a.W; ++r) st4(out + s_off[r] + static_cast<int64_t>(b) * s_stride[r] + within, v);
        }
      } else {
        const float v = in[mat] * a.scale;
        if (src >= 0) {
          out[s_off[src] + static_cast<int64_t>(b) * s_stride[src] + within] = v;
        } else {
          for (int r = 0; r < a.W; ++r) out[s_off[r] + static_cast<int64_t>(b) * s_stride[r] + within] = v;
        }
      }
    } else {
      if (VEC == 4) {
        float4 v;
        if (src >= 0) {
          v = ld4(in + s_off[src] + static_cast<int64_t>(b) * s_stride[src] + within);
        } else {
          v = ld4(in + s_off[0] + static_cast<int64_t>(b) * s_stride[0] + within);
          for (int r = 1; r < a.W; ++r) {
            const float4 o = ld4(in + s_off[r] + static_cast<int64_t>(b) * s_stride[r] + within);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
        }
        v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
        st4(out + mat, v);
      } else {
        float v;
        if (src >= 0) {
          v = in[s_off[src] + static_cast<int64_t>(b) * s_stride[src] + within];
        } else {
          v = in[s_off[0] + static_cast<int64_t>(b) * s_stride[0] + within];
          for (int r = 1; r < a.W; ++r) v += in[s_off[r] + static_cast<int64_t>(b) * s_stride[r] + within];
        }
        out[mat] = v * a.scale;
      }
    }
  }
}

static int launch_exchange(const float* in, float* out, const ExchangeArgs& a, bool pack, bool vec, hipStream_t st) {
  const size_t lds = (static_cast<size_t>(a.Fg) * 3 + 1 + a.W + 1) * sizeof(int32_t) + static_cast<size_t>(a.W) * sizeof(int64_t);
  if (lds > 60000) {
    set_error("pooled exchange: too many features (%d) / ranks (%d)", a.Fg, a.W);
    return TBE_ERR_UNSUPPORTED;
  }
  const int64_t total = static_cast<int64_t>(a.B_local) * (a.D_total / (vec ? 4 : 1));
  int64_t g = (total + 255) / 256;
  g = std::max<int64_t>(1, std::min<int64_t>(g, 256 * 32));
  const dim3 grid(static_cast<unsigned>(g));
  if (vec) {
    if (pack) hipLaunchKernelGGL((pooled_exchange_kernel<4, true>), grid, dim3(256), lds, st, in, out, a);
    else hipLaunchKernelGGL((pooled_exchange_kernel<4, false>), grid, dim3(256), lds, st, in, out, a);
  } else {
    if (pack) hipLaunchKernelGGL((pooled_exchange_kernel<1, true>), grid, dim3(256), lds, st, in, out, a);
    else hipLaunchKernelGGL((pooled_exchange_kernel<1, false>), grid, dim3(256), lds, st, in, out, a);
  }
  TBE_CHECK_LAUNCH("pooled exchange");
  return TBE_OK;
}

}  // namespace tbe

using namespace tbe;

static int exchange_entry(const float* in, float* out, const int32_t* feat_out_col, const int32_t* feat_src,
                          const int32_t* feat_slab_col, const int64_t* slab_offset, const int32_t* slab_stride,
                          int32_t Fg, int32_t W, int32_t B_local, int32_t D_total, int32_t all_multiple_of_4,
                          float scale, bool pack, void* stream, const char* name) {
  TBE_REQUIRE(Fg > 0 && W > 0 && B_local >= 0 && D_total >= 0, "%s: bad sizes", name);
  if (static_cast<int64_t>(B_local) * D_total == 0) return TBE_OK;
  TBE_REQUIRE(in && out && feat_out_col && feat_src && feat_slab_col && slab_offset && slab_stride, "%s: null pointer", name);
  ExchangeArgs a{feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, Fg, W, B_local, D_total, scale};
  const bool vec = all_multiple_of_4 && (D_total % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  return launch_exchange(in, out, a, pack, vec, static_cast<hipStream_t>(stream));
}

extern "C" int tbe_pooled_exchange_unpack(const float* recv, float* out, const int32_t* feat_out_col,
                                          const int32_t* feat_src, const int32_t* feat_slab_col,
                                          const int64_t* slab_offset, const int32_t* slab_stride, int32_t Fg,
                                          int32_t W, int32_t B_local, int32_t D_total, int32_t all_multiple_of_4,
                                          float scale, void* stream) {
  return exchange_entry(recv, out, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, Fg, W, B_local,
                        D_total, all_multiple_of_4, scale, false, stream, "tbe_pooled_exchange_unpack");
}

extern "C" int tbe_pooled_exchange_pack(const float* grad, float* send, const int32_t* feat_out_col,
                                        const int32_t* feat_src, const int32_t* feat_slab_col,
                                        const int64_t* slab_offset, const int32_t* slab_stride, int32_t Fg,
                                        int32_t W, int32_t B_local, int32_t D_total, int32_t all_multiple_of_4,
                                        float scale, void* stream) {
  return exchange_entry(grad, send, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, Fg, W, B_local,
                        D_total, all_multiple_of_4, scale, true, stream, "tbe_pooled_exchange_pack");
}

// ReLU backward + bias gradient of one MLP layer in ONE pass (gfx950).
//
// Reference: torchrec/modules/mlp.py:14-170 (Perceptron = Linear + relu) trained through autograd,
// which runs per layer  g' = threshold_backward(g, act)  and  bias.grad = g'.sum(0)  as two kernels
// (4 passes over a [B, N] matrix: read g, read act, write g', read g').  Here one kernel reads g and
// act once, writes g' and accumulates the column sums (3 passes); the column sums of the row blocks
// are combined by a second small kernel in fixed order (deterministic, no float atomics).
// HBM-bound elementwise work between the library GEMMs of the dense MLPs — the only place, besides
// the dot interaction, where this build touches the dense side.
#include <algorithm>
#include <cstring>

#include "common.hpp"

namespace tbe {

// rows of [B, N] handled by one workgroup of the first stage; the second stage sums B / rows partial rows.
// Wide layers have enough column tiles to fill the chip with 256-row blocks (4x fewer partials to sum).
__host__ __device__ inline int rows_per_block(int N) { return N >= 512 ? 256 : 64; }

// block = 256 threads = TY rows x TX float4-columns; grid = (column tiles, row blocks)
template <int TX>
__global__ __launch_bounds__(256) void drelu_bgrad_kernel(const float* __restrict__ gy, const float* __restrict__ act,
                                                          float* __restrict__ gx, float* __restrict__ partial,
                                                          int64_t B, int N) {
  constexpr int TY = 256 / TX;
  __shared__ float4 red[TY][TX];
  const int tx = threadIdx.x % TX;
  const int ty = threadIdx.x / TX;
  const int col = (blockIdx.x * TX + tx) * 4;
  const int rpb = rows_per_block(N);
  const int64_t row0 = static_cast<int64_t>(blockIdx.y) * rpb;
  const int64_t row1 = min(B, row0 + rpb);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N) {
#pragma unroll 4
    for (int64_t r = row0 + ty; r < row1; r += TY) {
      float4 g = ld4(gy + r * N + col);
      const float4 a = ld4(act + r * N + col);
      g.x = a.x > 0.f ? g.x : 0.f;
      g.y = a.y > 0.f ? g.y : 0.f;
      g.z = a.z > 0.f ? g.z : 0.f;
      g.w = a.w > 0.f ? g.w : 0.f;
      st4(gx + r * N + col, g);
      acc.x += g.x;
      acc.y += g.y;
      acc.z += g.z;
      acc.w += g.w;
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < N) {
    float4 s = red[0][tx];
#pragma unroll
    for (int y = 1; y < TY; ++y) {
      const float4 o = red[y][tx];
      s.x += o.x;
      s.y += o.y;
      s.z += o.z;
      s.w += o.w;
    }
    st4(partial + static_cast<int64_t>(blockIdx.y) * N + col, s);
  }
}

// partial[rb][c] = sum over the block's rows of w[b] * x[b, c]  (weight gradient of a Linear with ONE output:
// dW[0, c] = sum_b dy[b] * x[b, c] — a GEMM with N = 1 that the BLAS libraries run at ~1 % of HBM speed)
template <int TX>
__global__ __launch_bounds__(256) void wcolsum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ partial, int64_t B, int N) {
  constexpr int TY = 256 / TX;
  __shared__ float4 red[TY][TX];
  const int tx = threadIdx.x % TX;
  const int ty = threadIdx.x / TX;
  const int col = (blockIdx.x * TX + tx) * 4;
  const int rpb = rows_per_block(N);
  const int64_t row0 = static_cast<int64_t>(blockIdx.y) * rpb;
  const int64_t row1 = min(B, row0 + rpb);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N) {
#pragma unroll 4
    for (int64_t r = row0 + ty; r < row1; r += TY) {
      const float4 v = ld4(x + r * N + col);
      const float s = w[r];
      acc.x = fmaf(s, v.x, acc.x);
      acc.y = fmaf(s, v.y, acc.y);
      acc.z = fmaf(s, v.z, acc.z);
      acc.w = fmaf(s, v.w, acc.w);
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < N) {
    float4 s = red[0][tx];
#pragma unroll
    for (int y = 1; y < TY; ++y) {
      const float4 o = red[y][tx];
      s.x += o.x;
      s.y += o.y;
      s.z += o.z;
      s.w += o.w;
    }
    st4(partial + static_cast<int64_t>(blockIdx.y) * N + col, s);
  }
}

// bias_grad[c] = sum over row blocks of partial[rb][c], fixed order: wave w takes blocks w, w+4, ...
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, int64_t nblocks, int N,
                                                              float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (c < N) {
#pragma unroll 8
    for (int64_t rb = wave; rb < nblocks; rb += 4) acc += partial[rb * N + c];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < N) out[c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// dst[off_s + i] = scale * sum_{c < chunks_s} src_s[c * numel_s + i], chunk order fixed (deterministic), for every segment s
// of a table {src address, chunks, numel, dst element offset} held in device memory: ONE launch that finishes every
// split-K weight gradient (chunks = batch slices of the batched wgrad GEMM), every bias gradient (chunks = row blocks of
// drelu_bgrad_kernel / wcolsum_kernel) and the gradients that are already complete (chunks = 1) of a captured backward,
// scaled by 1 / world size, straight into the flat gradient buffer — instead of one reduce kernel per layer, one
// colsum_partials launch per layer, a cat and a mul (≈ 18 launches of ≈ 5 us inside the per-rank step's graphs).
// chunks = 0 writes zeros (a parameter that received no gradient).  grid = (tiles of 256 elements, segments).
struct ChunkSeg {
  const float* src;
  int64_t chunks;
  int64_t numel;
  int64_t dst_off;
};
constexpr int kChunkSegsByValue = 32;
struct ChunkSegs32 {
  ChunkSeg s[kChunkSegsByValue];
};
__device__ __forceinline__ void multi_chunk_sum_body(const ChunkSeg sg, float* __restrict__ dst, float scale);

// segment table in device memory (captured backward graphs: a static table per segment)
__global__ __launch_bounds__(256) void multi_chunk_sum_kernel(const ChunkSeg* __restrict__ segs, float* __restrict__ dst,
                                                              float scale) {
  multi_chunk_sum_body(segs[blockIdx.y], dst, scale);
}
// segment table passed BY VALUE in the kernel arguments (eager steps: sources and destinations change every step, and a
// host buffer the GPU reads later could be overwritten by a host that runs many steps ahead)
__global__ __launch_bounds__(256) void multi_chunk_sum_args_kernel(const ChunkSegs32 segs, float* __restrict__ dst, float scale) {
  multi_chunk_sum_body(segs.s[blockIdx.y], dst, scale);
}

__device__ __forceinline__ void multi_chunk_sum_body(const ChunkSeg sg, float* __restrict__ dst, float scale) {
  // many chunks (row-block sums of a bias: 32 ... 1024): block = 64 float4 columns x 4 waves; wave w adds chunks w, w + 4,
  // w + 8, ... (8 loads in flight), the four wave sums are combined through LDS in wave order.  Few chunks (split-K
  // slices of a weight: <= 16): block = 256 float4 columns, every thread adds all chunks of its column (all loads in
  // flight).  Either way the summation order is a function of (chunks) only.
  __shared__ float4 red[4][64];
  if (sg.chunks <= 16) {  // segment-uniform, hence block-uniform
    const int64_t j0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (j0 >= sg.numel) return;
    float* o = dst + sg.dst_off + j0;
    const bool v16 = ((sg.numel & 3) == 0) && ((reinterpret_cast<uintptr_t>(sg.src) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dst + sg.dst_off) & 15) == 0);
    if (v16) {
      float4 a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c)
        a[c] = c < sg.chunks ? ld4(sg.src + c * sg.numel + j0) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 acc = a[0];
#pragma unroll
      for (int c = 1; c < 16; ++c) {
        if (c < sg.chunks) {
          acc.x += a[c].x;
          acc.y += a[c].y;
          acc.z += a[c].z;
          acc.w += a[c].w;
        }
      }
      st4(o, make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale));
    } else {
      for (int k = 0; k < 4 && j0 + k < sg.numel; ++k) {
        float acc = 0.f;
        for (int64_t c = 0; c < sg.chunks; ++c) acc += sg.src[c * sg.numel + j0 + k];
        o[k] = acc * scale;
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * 64 + lane) * 4;
  if (static_cast<int64_t>(blockIdx.x) * 256 >= sg.numel) return;  // block-uniform
  const bool in = i0 < sg.numel;
  float* out = dst + sg.dst_off + i0;
  const bool vec = ((sg.numel & 3) == 0) && ((reinterpret_cast<uintptr_t>(sg.src) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(dst + sg.dst_off) & 15) == 0);
  const int nk = in ? static_cast<int>(min<int64_t>(4, sg.numel - i0)) : 0;
  auto load = [&](int64_t c) -> float4 {
    const float* p = sg.src + c * sg.numel + i0;
    if (vec) return ld4(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nk > 0) v.x = p[0];
    if (nk > 1) v.y = p[1];
    if (nk > 2) v.z = p[2];
    if (nk > 3) v.w = p[3];
    return v;
  };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    int64_t c = wave;
    for (; c + 28 < sg.chunks; c += 32) {
      float4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = load(c + 4 * u);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x += a[u].x;
        acc.y += a[u].y;
        acc.z += a[u].z;
        acc.w += a[u].w;
      }
    }
    for (; c < sg.chunks; c += 4) {
      const float4 a = load(c);
      acc.x += a.x;
      acc.y += a.y;
      acc.z += a.z;
      acc.w += a.w;
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && in) {
    float4 r;
    r.x = (((red[0][lane].x + red[1][lane].x) + red[2][lane].x) + red[3][lane].x) * scale;
    r.y = (((red[0][lane].y + red[1][lane].y) + red[2][lane].y) + red[3][lane].y) * scale;
    r.z = (((red[0][lane].z + red[1][lane].z) + red[2][lane].z) + red[3][lane].z) * scale;
    r.w = (((red[0][lane].w + red[1][lane].w) + red[2][lane].w) + red[3][lane].w) * scale;
    if (vec) {
      st4(out, r);
    } else {
      if (nk > 0) out[0] = r.x;
      if (nk > 1) out[1] = r.y;
      if (nk > 2) out[2] = r.z;
      if (nk > 3) out[3] = r.w;
    }
  }
}

// Binary cross entropy with logits, mean over the batch, forward AND gradient in one launch (the reference's train wrapper
// applies nn.BCEWithLogitsLoss: examples/dlrm/modules/dlrm_train.py; through torch that is 8 element-wise / reduce
// kernels forward and 5 backward, ~5 us each inside a graph):
//   l_i = max(x_i, 0) - x_i * y_i + log1p(exp(-|x_i|));  loss = (1 / B) sum_i l_i;  dlogits_i = (sigmoid(x_i) - y_i) / B
// Blocks own contiguous ranges and reduce in a fixed order; the last block to finish (ticket) adds the block partials in
// index order and resets the ticket: deterministic, one launch.  labels: float32 or int64.
constexpr int kBceMaxBlocks = 64;
template <typename LabelT>
__global__ __launch_bounds__(256) void bce_with_logits_kernel(const float* __restrict__ x, const LabelT* __restrict__ y, int64_t B,
                                                              float* __restrict__ loss, float* __restrict__ dx,
                                                              float* __restrict__ partial, unsigned int* __restrict__ ticket) {
  __shared__ float red[256];
  __shared__ bool is_last;
  const int nblk = gridDim.x;
  const int64_t per = (B + nblk - 1) / nblk;
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * per;
  const int64_t hi = min(B, lo + per);
  const float invB = 1.f / static_cast<float>(B);
  float acc = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float xi = x[i];
    const float yi = static_cast<float>(y[i]);
    const float e = __expf(-fabsf(xi));
    acc += fmaxf(xi, 0.f) - xi * yi + log1pf(e);
    if (dx != nullptr) {
      const float sig = xi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      dx[i] = (sig - yi) * invB;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    is_last = atomicAdd(ticket, 1u) == static_cast<unsigned>(nblk - 1);
  }
  __syncthreads();
  if (is_last && threadIdx.x == 0) {
    __threadfence();
    float tot = 0.f;
    for (int b = 0; b < nblk; ++b) tot += __hip_atomic_load(&partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *loss = tot * invB;
    *ticket = 0u;  // ready for the next launch on this workspace
  }
}

}  // namespace tbe

using namespace tbe;

extern "C" size_t tbe_relu_backward_bias_grad_workspace_bytes(int64_t B, int32_t N) {
  if (B <= 0 || N <= 0) return 256;
  const int rpb = rows_per_block(N);
  return align_up(static_cast<size_t>((B + rpb - 1) / rpb) * N * sizeof(float), 256);
}

extern "C" int tbe_relu_backward_bias_grad_f32(const float* grad_out, const float* act, int64_t B, int32_t N,
                                               float* grad_in, float* bias_grad, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  TBE_REQUIRE(B >= 0 && N > 0 && (N & 3) == 0, "tbe_relu_backward_bias_grad_f32: N=%d must be a positive multiple of 4", N);
  TBE_REQUIRE(bias_grad != nullptr, "tbe_relu_backward_bias_grad_f32: null bias_grad");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B == 0) {
    if (hipMemsetAsync(bias_grad, 0, sizeof(float) * N, st) != hipSuccess) return TBE_ERR_LAUNCH;
    return TBE_OK;
  }
  TBE_REQUIRE(grad_out && act && grad_in && workspace, "tbe_relu_backward_bias_grad_f32: null pointer");
  TBE_REQUIRE(((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(act) |
                reinterpret_cast<uintptr_t>(grad_in) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
              "tbe_relu_backward_bias_grad_f32: tensors must be 16-B aligned");
  TBE_REQUIRE(workspace_bytes >= tbe_relu_backward_bias_grad_workspace_bytes(B, N),
              "tbe_relu_backward_bias_grad_f32: workspace too small");
  const int rpb = rows_per_block(N);
  const int64_t nrb = (B + rpb - 1) / rpb;
  TBE_REQUIRE(nrb <= 65535, "tbe_relu_backward_bias_grad_f32: B=%lld too large", (long long)B);
  float* partial = static_cast<float*>(workspace);
  const int vecs = N / 4;
  if (vecs >= 64) {
    const dim3 grid((vecs + 63) / 64, static_cast<unsigned>(nrb));
    hipLaunchKernelGGL(drelu_bgrad_kernel<64>, grid, dim3(256), 0, st, grad_out, act, grad_in, partial, B, N);
  } else if (vecs >= 32) {
    const dim3 grid((vecs + 31) / 32, static_cast<unsigned>(nrb));
    hipLaunchKernelGGL(drelu_bgrad_kernel<32>, grid, dim3(256), 0, st, grad_out, act, grad_in, partial, B, N);
  } else {
    const dim3 grid((vecs + 15) / 16, static_cast<unsigned>(nrb));
    hipLaunchKernelGGL(drelu_bgrad_kernel<16>, grid, dim3(256), 0, st, grad_out, act, grad_in, partial, B, N);
  }
  TBE_CHECK_LAUNCH("tbe_relu_backward_bias_grad_f32");
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((N + 63) / 64), dim3(256), 0, st, partial, nrb, N, bias_grad);
  TBE_CHECK_LAUNCH("tbe_relu_backward_bias_grad_f32 colsum");
  return TBE_OK;
}

extern "C" size_t tbe_weighted_colsum_workspace_bytes(int64_t B, int32_t N) {
  return tbe_relu_backward_bias_grad_workspace_bytes(B, N);
}

extern "C" int tbe_weighted_colsum_f32(const float* x, const float* w, int64_t B, int32_t N, float* out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  TBE_REQUIRE(B >= 0 && N > 0 && (N & 3) == 0, "tbe_weighted_colsum_f32: N=%d must be a positive multiple of 4", N);
  TBE_REQUIRE(out != nullptr, "tbe_weighted_colsum_f32: null out");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B == 0) {
    if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) return TBE_ERR_LAUNCH;
    return TBE_OK;
  }
  TBE_REQUIRE(x && w && workspace, "tbe_weighted_colsum_f32: null pointer");
  TBE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
              "tbe_weighted_colsum_f32: tensors must be 16-B aligned");
  TBE_REQUIRE(workspace_bytes >= tbe_weighted_colsum_workspace_bytes(B, N), "tbe_weighted_colsum_f32: workspace too small");
  const int rpb = rows_per_block(N);
  const int64_t nrb = (B + rpb - 1) / rpb;
  TBE_REQUIRE(nrb <= 65535, "tbe_weighted_colsum_f32: B=%lld too large", (long long)B);
  float* partial = static_cast<float*>(workspace);
  const int vecs = N / 4;
  if (vecs >= 64) {
    hipLaunchKernelGGL(wcolsum_kernel<64>, dim3((vecs + 63) / 64, static_cast<unsigned>(nrb)), dim3(256), 0, st, x, w, partial, B, N);
  } else if (vecs >= 32) {
    hipLaunchKernelGGL(wcolsum_kernel<32>, dim3((vecs + 31) / 32, static_cast<unsigned>(nrb)), dim3(256), 0, st, x, w, partial, B, N);
  } else {
    hipLaunchKernelGGL(wcolsum_kernel<16>, dim3((vecs + 15) / 16, static_cast<unsigned>(nrb)), dim3(256), 0, st, x, w, partial, B, N);
  }
  TBE_CHECK_LAUNCH("tbe_weighted_colsum_f32");
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((N + 63) / 64), dim3(256), 0, st, partial, nrb, N, out);
  TBE_CHECK_LAUNCH("tbe_weighted_colsum_f32 colsum");
  return TBE_OK;
}


// ---- first stages alone: the column sums of the row blocks are left in `partial` [row blocks][N] for a later
//      tbe_multi_chunk_sum_f32 (one launch for every layer of a captured backward) --------------------------------------
extern "C" int64_t tbe_colsum_row_blocks(int64_t B, int32_t N) {
  if (B <= 0 || N <= 0) return 0;
  const int rpb = rows_per_block(N);
  return (B + rpb - 1) / rpb;
}

extern "C" int tbe_relu_backward_bias_partials_f32(const float* grad_out, const float* act, int64_t B, int32_t N,
                                                   float* grad_in, float* partial, size_t partial_bytes, void* stream) {
  TBE_REQUIRE(B > 0 && N > 0 && (N & 3) == 0, "tbe_relu_backward_bias_partials_f32: B > 0 and N a positive multiple of 4 required");
  TBE_REQUIRE(grad_out && act && grad_in && partial, "tbe_relu_backward_bias_partials_f32: null pointer");
  TBE_REQUIRE(((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(act) |
                reinterpret_cast<uintptr_t>(grad_in) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0,
              "tbe_relu_backward_bias_partials_f32: tensors must be 16-B aligned");
  const int64_t nrb = tbe_colsum_row_blocks(B, N);
  TBE_REQUIRE(nrb <= 65535, "tbe_relu_backward_bias_partials_f32: B=%lld too large", (long long)B);
  TBE_REQUIRE(partial_bytes >= static_cast<size_t>(nrb) * N * sizeof(float), "tbe_relu_backward_bias_partials_f32: partial too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vecs = N / 4;
  if (vecs >= 64) {
    hipLaunchKernelGGL(drelu_bgrad_kernel<64>, dim3((vecs + 63) / 64, static_cast<unsigned>(nrb)), dim3(256), 0, st, grad_out, act, grad_in, partial, B, N);
  } else if (vecs >= 32) {
    hipLaunchKernelGGL(drelu_bgrad_kernel<32>, dim3((vecs + 31) / 32, static_cast<unsigned>(nrb)), dim3(256), 0, st, grad_out, act, grad_in, partial, B, N);
  } else {
    hipLaunchKernelGGL(drelu_bgrad_kernel<16>, dim3((vecs + 15) / 16, static_cast<unsigned>(nrb)), dim3(256), 0, st, grad_out, act, grad_in, partial, B, N);
  }
  TBE_CHECK_LAUNCH("tbe_relu_backward_bias_partials_f32");
  return TBE_OK;
}

extern "C" int tbe_weighted_colsum_partials_f32(const float* x, const float* w, int64_t B, int32_t N, float* partial,
                                                size_t partial_bytes, void* stream) {
  TBE_REQUIRE(B > 0 && N > 0 && (N & 3) == 0, "tbe_weighted_colsum_partials_f32: B > 0 and N a positive multiple of 4 required");
  TBE_REQUIRE(x && w && partial, "tbe_weighted_colsum_partials_f32: null pointer");
  TBE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0,
              "tbe_weighted_colsum_partials_f32: tensors must be 16-B aligned");
  const int64_t nrb = tbe_colsum_row_blocks(B, N);
  TBE_REQUIRE(nrb <= 65535, "tbe_weighted_colsum_partials_f32: B=%lld too large", (long long)B);
  TBE_REQUIRE(partial_bytes >= static_cast<size_t>(nrb) * N * sizeof(float), "tbe_weighted_colsum_partials_f32: partial too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vecs = N / 4;
  if (vecs >= 64) {
    hipLaunchKernelGGL(wcolsum_kernel<64>, dim3((vecs + 63) / 64, static_cast<unsigned>(nrb)), dim3(256), 0, st, x, w, partial, B, N);
  } else if (vecs >= 32) {
    hipLaunchKernelGGL(wcolsum_kernel<32>, dim3((vecs + 31) / 32, static_cast<unsigned>(nrb)), dim3(256), 0, st, x, w, partial, B, N);
  } else {
    hipLaunchKernelGGL(wcolsum_kernel<16>, dim3((vecs + 15) / 16, static_cast<unsigned>(nrb)), dim3(256), 0, st, x, w, partial, B, N);
  }
  TBE_CHECK_LAUNCH("tbe_weighted_colsum_partials_f32");
  return TBE_OK;
}

extern "C" int tbe_multi_chunk_sum_f32(const int64_t* seg_table, int32_t nseg, int64_t max_numel, float* dst, float scale,
                                       void* stream) {
  TBE_REQUIRE(nseg >= 0 && max_numel >= 0, "tbe_multi_chunk_sum_f32: bad sizes");
  if (nseg == 0 || max_numel == 0) return TBE_OK;
  TBE_REQUIRE(seg_table && dst, "tbe_multi_chunk_sum_f32: null pointer");
  TBE_REQUIRE(nseg <= 65535, "tbe_multi_chunk_sum_f32: more than 65535 segments");
  static_assert(sizeof(ChunkSeg) == 4 * sizeof(int64_t), "segment table layout");
  const int64_t tiles = (max_numel + 255) / 256;
  TBE_REQUIRE(tiles < (1ll << 31), "tbe_multi_chunk_sum_f32: segment too long");
  hipLaunchKernelGGL(multi_chunk_sum_kernel, dim3(static_cast<unsigned>(tiles), static_cast<unsigned>(nseg)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const ChunkSeg*>(seg_table), dst, scale);
  TBE_CHECK_LAUNCH("tbe_multi_chunk_sum_f32");
  return TBE_OK;
}

extern "C" int tbe_multi_chunk_sum_host_table_f32(const int64_t* host_seg_table, int32_t nseg, int64_t max_numel, float* dst,
                                                  float scale, void* stream) {
  TBE_REQUIRE(nseg >= 0 && nseg <= kChunkSegsByValue && max_numel >= 0,
              "tbe_multi_chunk_sum_host_table_f32: 0 <= nseg <= %d required", kChunkSegsByValue);
  if (nseg == 0 || max_numel == 0) return TBE_OK;
  TBE_REQUIRE(host_seg_table != nullptr, "tbe_multi_chunk_sum_host_table_f32: null table");
  ChunkSegs32 segs{};
  memcpy(segs.s, host_seg_table, static_cast<size_t>(nseg) * sizeof(ChunkSeg));
  const int64_t tiles = (max_numel + 255) / 256;
  TBE_REQUIRE(tiles < (1ll << 31), "tbe_multi_chunk_sum_host_table_f32: segment too long");
  hipLaunchKernelGGL(multi_chunk_sum_args_kernel, dim3(static_cast<unsigned>(tiles), static_cast<unsigned>(nseg)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), segs, dst, scale);
  TBE_CHECK_LAUNCH("tbe_multi_chunk_sum_host_table_f32");
  return TBE_OK;
}

extern "C" size_t tbe_bce_with_logits_workspace_bytes(void) { return 512; }  // [64 block partials | ticket], zeroed ONCE

extern "C" int tbe_bce_with_logits_f32(const float* logits, const void* labels, int32_t label_elem_size, int64_t B, float* loss,
                                       float* dlogits, void* workspace, size_t workspace_bytes, void* stream) {
  TBE_REQUIRE(B > 0, "tbe_bce_with_logits_f32: empty batch");
  TBE_REQUIRE(label_elem_size == 4 || label_elem_size == 8, "tbe_bce_with_logits_f32: labels must be float32 (4) or int64 (8)");
  TBE_REQUIRE(logits && labels && loss && workspace, "tbe_bce_with_logits_f32: null pointer");
  TBE_REQUIRE(workspace_bytes >= tbe_bce_with_logits_workspace_bytes() && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "tbe_bce_with_logits_f32: workspace too small or misaligned");
  float* partial = static_cast<float*>(workspace);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(partial + kBceMaxBlocks);
  const unsigned nblk = static_cast<unsigned>(std::min<int64_t>(kBceMaxBlocks, (B + 2047) / 2048));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (label_elem_size == 4)
    hipLaunchKernelGGL(bce_with_logits_kernel<float>, dim3(nblk), dim3(256), 0, st, logits, static_cast<const float*>(labels), B,
                       loss, dlogits, partial, ticket);
  else
    hipLaunchKernelGGL(bce_with_logits_kernel<int64_t>, dim3(nblk), dim3(256), 0, st, logits, static_cast<const int64_t*>(labels),
                       B, loss, dlogits, partial, ticket);
  TBE_CHECK_LAUNCH("tbe_bce_with_logits_f32");
  return TBE_OK;
}

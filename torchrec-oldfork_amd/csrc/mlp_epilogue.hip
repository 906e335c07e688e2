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

// Fused DLRM dot interaction for gfx950 — the one MFMA user of the embedding hot path.
//
// Reference: InteractionArch.forward, torchrec/models/dlrm.py:193-219
//   combined = cat(dense[B,1,D], sparse[B,F,D])           (copy)
//   inter    = bmm(combined, combined^T)                    (batched 27x128x27 GEMM)
//   flat     = inter[:, triu_indices(F+1, F+1, offset=1)]   (gather)
//   out      = cat(dense, flat)                             (copy)            -> [B, D + (F+1)F/2]
// and its autograd backward (index_put, 2 bmm, cat backward).  Here each is ONE kernel that reads
// every input byte once and writes every output byte once (HBM-bound: 15.7 KB/sample forward,
// 29.5 KB/sample backward at F = 26, D = 128).
//
// A wave owns one sample at a time: X = [dense; sparse] (R = F+1 <= 32 rows) is staged in an LDS
// tile private to the wave, products run on v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf
// chain, so the oracle reproduces it bit for bit), results are re-staged in LDS and stored
// coalesced.  No inter-wave synchronisation.
//   forward : Z = X X^T, 3 of the 4 16x16 tiles (upper triangle), K = D
//   backward: dX = (G + G^T) X with G the strict upper-triangular matrix of d(flat),
//             2 x D/16 tiles, K = R; d(dense) additionally receives d(out)[:, :D].
#include <algorithm>

#include "common.hpp"

namespace tbe {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// index of pair (i, j), i < j < R, in torch.triu_indices(R, R, offset=1) row-major order
__device__ __forceinline__ int triu_index(int i, int j, int R) { return i * (2 * R - i - 1) / 2 + (j - i - 1); }

__global__ __launch_bounds__(256, 2) void interaction_fwd_kernel(const float* __restrict__ dense,
                                                                 const float* __restrict__ sparse,
                                                                 float* __restrict__ out, int B, int F, int D) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int R = F + 1;
  const int P = R * (R - 1) / 2;
  const int OUT = D + P;
  const int XS = D + 2;  // row stride: (2*row + k) % 32 distinct over a half-wave => no bank conflict
  const int per_wave = 32 * XS + ((P + 3) & ~3);
  float* xs = smem + wave * per_wave;
  float* zs = xs + 32 * XS;
  const int r16 = lane & 15;
  const int kq = lane >> 4;
  const int nvec = R * D / 4;

  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    // stage X: row 0 = dense[b], rows 1..F = sparse[b]
    for (int v = lane; v < nvec; v += kWave) {
      const int e = v * 4;
      const int r = e / D;
      const int c = e - r * D;
      const float* src = (r == 0) ? dense + static_cast<int64_t>(b) * D + c
                                  : sparse + (static_cast<int64_t>(b) * F + (r - 1)) * D + c;
      const float4 x = ld4(src);
      float2* dst = reinterpret_cast<float2*>(xs + r * XS + c);
      dst[0] = make_float2(x.x, x.y);
      dst[1] = make_float2(x.z, x.w);
    }
    wave_lds_fence();
    f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc11 = acc00;
    const float* pa0 = xs + r16 * XS + kq;
    const float* pa1 = xs + (16 + r16) * XS + kq;
#pragma unroll 8
    for (int k0 = 0; k0 < D; k0 += 4) {
      const float a0 = pa0[k0];
      const float a1 = pa1[k0];  // rows >= R hold stale data: they only reach Z rows/cols >= R, never stored
      acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, acc01, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc11, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = kq * 4 + q;
      const int j = r16;
      if (i < j && j < R) zs[triu_index(i, j, R)] = acc00[q];
      if (16 + j < R && i < R) zs[triu_index(i, 16 + j, R)] = acc01[q];
      if (i < j && 16 + j < R) zs[triu_index(16 + i, 16 + j, R)] = acc11[q];
    }
    wave_lds_fence();
    float* orow = out + static_cast<int64_t>(b) * OUT;
    for (int c = lane; c < D; c += kWave) orow[c] = xs[c];
    for (int p = lane; p < P; p += kWave) orow[D + p] = zs[p];
    wave_lds_fence();  // xs/zs are rewritten by the next sample
  }
}

template <int NT>  // NT = D / 16 column tiles
__global__ __launch_bounds__(256, 2) void interaction_bwd_kernel(const float* __restrict__ dense,
                                                                 const float* __restrict__ sparse,
                                                                 const float* __restrict__ grad_out,
                                                                 float* __restrict__ grad_dense,
                                                                 float* __restrict__ grad_sparse, int B, int F) {
  extern __shared__ float smem[];
  constexpr int D = NT * 16;
  constexpr int XS = D + 16;  // B-operand reads (16k + 16n + c) % 32: conflict-free over a half-wave
  constexpr int GS = 34;      // A-operand reads (2*row + k) % 32: conflict-free
  constexpr int XROWS = 28;   // K steps cover rows 0..27
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int R = F + 1;
  const int P = R * (R - 1) / 2;
  const int OUT = D + P;
  float* xs = smem + wave * (XROWS * XS + 32 * GS);
  float* gs = xs + XROWS * XS;
  const int r16 = lane & 15;
  const int kq = lane >> 4;
  const int nvec = R * D / 4;
  const int ksteps = (R + 3) / 4;
  // rows R..27 of X and the whole of G start as zeros; per sample only defined entries are rewritten
  for (int e = lane; e < XROWS * XS; e += kWave) xs[e] = 0.f;
  for (int e = lane; e < 32 * GS; e += kWave) gs[e] = 0.f;
  wave_lds_fence();

  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    for (int v = lane; v < nvec; v += kWave) {
      const int e = v * 4;
      const int r = e / D;
      const int c = e - r * D;
      const float* src = (r == 0) ? dense + static_cast<int64_t>(b) * D + c
                                  : sparse + (static_cast<int64_t>(b) * F + (r - 1)) * D + c;
      st4(xs + r * XS + c, ld4(src));
    }
    const float* grow = grad_out + static_cast<int64_t>(b) * OUT;
    for (int p = lane; p < P; p += kWave) {
      const float g = grow[D + p];
      int i = 0, rem = p;
      while (rem >= R - 1 - i) {  // row of pair p in the strict upper triangle
        rem -= R - 1 - i;
        ++i;
      }
      const int j = i + 1 + rem;
      gs[i * GS + j] = g;
      gs[j * GS + i] = g;
    }
    wave_lds_fence();
    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < ksteps; ++ks) {
      const int k = ks * 4 + kq;
      const float a0 = gs[r16 * GS + k];
      const float a1 = gs[(16 + r16) * GS + k];
      const float* xrow = xs + k * XS + r16;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float bn = xrow[16 * n];
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bn, acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bn, acc[1][n], 0, 0, 0);
      }
    }
    wave_lds_fence();  // every lane is done reading X before it is overwritten with dX
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * m + 4 * kq + q;
          if (row < R) xs[row * XS + 16 * n + r16] = acc[m][n][q];
        }
    wave_lds_fence();
    for (int v = lane; v < nvec; v += kWave) {
      const int e = v * 4;
      const int r = e / D;
      const int c = e - r * D;
      float4 x = ld4(xs + r * XS + c);
      if (r == 0) {
        x.x += grow[c + 0];
        x.y += grow[c + 1];
        x.z += grow[c + 2];
        x.w += grow[c + 3];
        st4(grad_dense + static_cast<int64_t>(b) * D + c, x);
      } else {
        st4(grad_sparse + (static_cast<int64_t>(b) * F + (r - 1)) * D + c, x);
      }
    }
    wave_lds_fence();
  }
}

}  // namespace tbe

using namespace tbe;

static unsigned interaction_grid(int B) {
  // persistent-style: 2 workgroups per CU, each wave strides over samples
  const int64_t want = (static_cast<int64_t>(B) + 3) / 4;
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(want, 256 * 2)));
}

extern "C" int tbe_dlrm_interaction_forward_f32(const float* dense, const float* sparse, int32_t B, int32_t F,
                                                int32_t D, float* out, void* stream) {
  TBE_REQUIRE(B >= 0 && F >= 1 && F <= 31, "tbe_dlrm_interaction_forward_f32: F=%d outside [1, 31]", F);
  TBE_REQUIRE(D >= 4 && D <= 256 && D % 4 == 0, "tbe_dlrm_interaction_forward_f32: D=%d must be a multiple of 4 in [4, 256]", D);
  if (B == 0) return TBE_OK;
  TBE_REQUIRE(dense && sparse && out, "tbe_dlrm_interaction_forward_f32: null pointer");
  TBE_REQUIRE(((reinterpret_cast<uintptr_t>(dense) | reinterpret_cast<uintptr_t>(sparse)) & 15) == 0,
              "tbe_dlrm_interaction_forward_f32: inputs must be 16-B aligned");
  const int R = F + 1, P = R * (R - 1) / 2;
  const size_t lds = 4 * (static_cast<size_t>(32) * (D + 2) + ((P + 3) & ~3)) * sizeof(float);
  static size_t fwd_lds_set = 0;
  if (lds > fwd_lds_set) {  // dynamic LDS above 64 KB must be opted into
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(interaction_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess) {
      set_error("tbe_dlrm_interaction_forward_f32: cannot reserve %zu B of LDS", lds);
      return TBE_ERR_LAUNCH;
    }
    fwd_lds_set = lds;
  }
  hipLaunchKernelGGL(interaction_fwd_kernel, dim3(interaction_grid(B)), dim3(256), lds, static_cast<hipStream_t>(stream),
                     dense, sparse, out, B, F, D);
  TBE_CHECK_LAUNCH("tbe_dlrm_interaction_forward_f32");
  return TBE_OK;
}

extern "C" int tbe_dlrm_interaction_backward_f32(const float* dense, const float* sparse, const float* grad_out,
                                                 int32_t B, int32_t F, int32_t D, float* grad_dense,
                                                 float* grad_sparse, void* stream) {
  TBE_REQUIRE(B >= 0 && F >= 1 && F <= 27, "tbe_dlrm_interaction_backward_f32: F=%d outside [1, 27]", F);
  TBE_REQUIRE(D == 16 || D == 32 || D == 64 || D == 128, "tbe_dlrm_interaction_backward_f32: D=%d not in {16,32,64,128}", D);
  if (B == 0) return TBE_OK;
  TBE_REQUIRE(dense && sparse && grad_out && grad_dense && grad_sparse, "tbe_dlrm_interaction_backward_f32: null pointer");
  TBE_REQUIRE(((reinterpret_cast<uintptr_t>(dense) | reinterpret_cast<uintptr_t>(sparse) |
                reinterpret_cast<uintptr_t>(grad_dense) | reinterpret_cast<uintptr_t>(grad_sparse)) & 15) == 0,
              "tbe_dlrm_interaction_backward_f32: tensors must be 16-B aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = 4 * (static_cast<size_t>(28) * (D + 16) + 32 * 34) * sizeof(float);
  const dim3 grid(interaction_grid(B));
  static bool bwd_lds_set = false;
  if (!bwd_lds_set) {
    const int big = static_cast<int>(4 * (28 * (128 + 16) + 32 * 34) * sizeof(float));
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(interaction_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, big) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(interaction_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, big) == hipSuccess;
    if (!ok) {
      set_error("tbe_dlrm_interaction_backward_f32: cannot reserve LDS");
      return TBE_ERR_LAUNCH;
    }
    bwd_lds_set = true;
  }
  switch (D) {
    case 16: hipLaunchKernelGGL(interaction_bwd_kernel<1>, grid, dim3(256), lds, st, dense, sparse, grad_out, grad_dense, grad_sparse, B, F); break;
    case 32: hipLaunchKernelGGL(interaction_bwd_kernel<2>, grid, dim3(256), lds, st, dense, sparse, grad_out, grad_dense, grad_sparse, B, F); break;
    case 64: hipLaunchKernelGGL(interaction_bwd_kernel<4>, grid, dim3(256), lds, st, dense, sparse, grad_out, grad_dense, grad_sparse, B, F); break;
    default: hipLaunchKernelGGL(interaction_bwd_kernel<8>, grid, dim3(256), lds, st, dense, sparse, grad_out, grad_dense, grad_sparse, B, F); break;
  }
  TBE_CHECK_LAUNCH("tbe_dlrm_interaction_backward_f32");
  return TBE_OK;
}

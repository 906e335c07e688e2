/*
 * dlrm_oracle.c — CPU restatement of the DLRM dot interaction (TEST INFRASTRUCTURE ONLY).
 * Follows InteractionArch.forward, torchrec/models/dlrm.py:193-219:
 *   combined = cat(dense.unsqueeze(1), sparse); inter = bmm(combined, combined^T);
 *   out = cat(dense, inter[:, triu_indices(F+1, F+1, offset=1)]).
 * Pinned against the reference module itself by tests/golden/dlrm_small.npz (make_golden.py).
 * Sums are fmaf chains starting from 0 — the arithmetic v_mfma_f32_16x16x4_f32 performs — walked
 * in the order the HIP kernels feed the matrix core, so they can be compared bit for bit: the
 * backward in ascending k; the forward over columns 16 s + 4 q + e in (s, e, q) order (its MFMA
 * operands are 16-B global loads: lane quarter q holds columns 16 s + 4 q .. + 3 of segment s;
 * csrc/dlrm_interaction.hip).  D must be a multiple of 16 for that order; other D fall back to
 * ascending k (no HIP kernel exists for them).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static const float* row_of(const float* dense, const float* sparse, int64_t b, int32_t F, int32_t D, int32_t r) {
  return r == 0 ? dense + b * D : sparse + (b * F + (r - 1)) * D;
}

void oracle_interaction_forward(const float* dense, const float* sparse, int32_t B, int32_t F, int32_t D, float* out) {
  const int32_t R = F + 1, P = R * (R - 1) / 2, OUT = D + P;
  for (int64_t b = 0; b < B; ++b) {
    float* o = out + b * OUT;
    for (int32_t d = 0; d < D; ++d) o[d] = dense[b * D + d];
    int32_t p = 0;
    for (int32_t i = 0; i < R; ++i)
      for (int32_t j = i + 1; j < R; ++j) {
        const float* xi = row_of(dense, sparse, b, F, D, i);
        const float* xj = row_of(dense, sparse, b, F, D, j);
        float acc = 0.f;
        if (D % 16 == 0) {
          for (int32_t s = 0; s < D / 16; ++s)
            for (int32_t e = 0; e < 4; ++e)
              for (int32_t q = 0; q < 4; ++q) {
                const int32_t k = 16 * s + 4 * q + e;
                acc = fmaf(xi[k], xj[k], acc);
              }
        } else {
          for (int32_t k = 0; k < D; ++k) acc = fmaf(xi[k], xj[k], acc);
        }
        o[D + p++] = acc;
      }
  }
}

/* dX = (G + G^T) X, G strict upper triangular from grad_out[:, D:]; grad_dense += grad_out[:, :D]. */
void oracle_interaction_backward(const float* dense, const float* sparse, const float* grad_out, int32_t B, int32_t F,
                                 int32_t D, float* grad_dense, float* grad_sparse) {
  const int32_t R = F + 1, P = R * (R - 1) / 2, OUT = D + P;
  float* G = (float*)malloc(sizeof(float) * R * R);
  for (int64_t b = 0; b < B; ++b) {
    const float* go = grad_out + b * OUT;
    for (int32_t i = 0; i < R * R; ++i) G[i] = 0.f;
    int32_t p = 0;
    for (int32_t i = 0; i < R; ++i)
      for (int32_t j = i + 1; j < R; ++j) {
        G[i * R + j] = go[D + p];
        G[j * R + i] = go[D + p];
        ++p;
      }
    for (int32_t i = 0; i < R; ++i) {
      float* dst = i == 0 ? grad_dense + b * D : grad_sparse + (b * F + (i - 1)) * D;
      for (int32_t c = 0; c < D; ++c) {
        float acc = 0.f;
        for (int32_t j = 0; j < R; ++j) acc = fmaf(G[i * R + j], row_of(dense, sparse, b, F, D, j)[c], acc);
        dst[c] = i == 0 ? acc + go[c] : acc;
      }
    }
  }
  free(G);
}

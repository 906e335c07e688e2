/*
 * tbe_oracle.c — CPU restatement (plain C, scalar) of the hot path's algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (torchrec-oldfork_amd/) may import, link
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use
 * it, and only as the checker.
 *
 * Where the algorithm comes from.  The arithmetic of this path lives in fbgemm_gpu, an
 * un-vendored, un-pinned submodule of the reference (.gitmodules:1-4, third_party/fbgemm is
 * empty), so each function below restates the PUBLISHED fbgemm semantics and is anchored on the
 * reference's own call sites and tests:
 *   - pooled forward / exact-SGD backward: pinned against the reference's CPU
 *     EmbeddingBagCollection (torchrec/modules/embedding_modules.py:127-193, nn.EmbeddingBag
 *     include_last_offset=True) + torch.optim.SGD, the ground truth of the reference's own
 *     sharded-vs-unsharded test (torchrec/distributed/test_utils/test_model_parallel_base.py:257-283);
 *     golden vectors in tests/golden/ were produced by importing that module (make_golden.py).
 *   - permute_2D: known answers of torchrec/sparse/tests/test_jagged_tensor.py:632-755.
 *   - block_bucketize: python reference torchrec/distributed/tests/test_utils.py:83-236.
 *   - cumsum: torchrec/sparse/jagged_tensor.py:27-36 (_cumsum / _to_offsets).
 *   - row-wise Adagrad / Adam / Adagrad arithmetic: public fbgemm formula only — PARITY UNPINNED
 *     (the reference's test_fused_optim.py:201-306 compares fused-vs-fused across shardings).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).  fmaf() is used wherever the
 * HIP kernels use an explicit fused multiply-add so duplicate-free cases agree bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define POOL_SUM 0
#define POOL_MEAN 1
#define POOL_NONE 2

#define OPT_EXACT_SGD 0
#define OPT_EXACT_ROWWISE_ADAGRAD 1
#define OPT_ADAM 2
#define OPT_EXACT_ADAGRAD 3
#define OPT_DENSE_GRAD 100

/* out[b, Doff[f] + d] = sum_i w_i * W_f[idx_i, d], accumulated in position order.
 * feat_weights[f] = pointer to the table of feature f ([rows, D] row-major).
 * Out-of-range indices contribute nothing and are counted (returned). */
int64_t oracle_tbe_forward_pooled(const float* const* feat_weights, const int32_t* feat_D,
                                  const int32_t* feat_D_offset, const int64_t* feat_rows, int32_t F,
                                  int32_t B, const int64_t* indices, const int64_t* offsets,
                                  const float* psw, int32_t pooling_mode, float* out,
                                  int64_t out_stride) {
  int64_t bad = 0;
  for (int32_t f = 0; f < F; ++f) {
    const int32_t D = feat_D[f];
    const float* W = feat_weights[f];
    for (int32_t b = 0; b < B; ++b) {
      const int64_t s = offsets[(int64_t)f * B + b], e = offsets[(int64_t)f * B + b + 1];
      float* o = out + (int64_t)b * out_stride + feat_D_offset[f];
      for (int32_t d = 0; d < D; ++d) o[d] = 0.f;
      for (int64_t i = s; i < e; ++i) {
        const int64_t idx = indices[i];
        if (idx < 0 || idx >= feat_rows[f]) {
          ++bad;
          continue;
        }
        const float w = psw ? psw[i] : 1.f;
        const float* row = W + idx * D;
        for (int32_t d = 0; d < D; ++d) o[d] = fmaf(w, row[d], o[d]);
      }
      if (pooling_mode == POOL_MEAN) {
        const float scale = e > s ? 1.f / (float)(e - s) : 0.f;
        for (int32_t d = 0; d < D; ++d) o[d] *= scale;
      }
    }
  }
  return bad;
}

/* PoolingMode.NONE: out[i, :] = W_f(i)[indices[i], :] */
int64_t oracle_tbe_forward_nobag(const float* const* feat_weights, const int64_t* feat_rows, int32_t F,
                                 int32_t B, int32_t D, const int64_t* indices,
                                 const int64_t* offsets, float* out) {
  int64_t bad = 0;
  for (int32_t f = 0; f < F; ++f) {
    const int64_t s = offsets[(int64_t)f * B], e = offsets[(int64_t)(f + 1) * B];
    for (int64_t i = s; i < e; ++i) {
      const int64_t idx = indices[i];
      float* o = out + i * D;
      if (idx < 0 || idx >= feat_rows[f]) {
        ++bad;
        for (int32_t d = 0; d < D; ++d) o[d] = 0.f;
        continue;
      }
      memcpy(o, feat_weights[f] + idx * D, sizeof(float) * D);
    }
  }
  return bad;
}

typedef struct {
  int64_t key; /* global row = feat_row_base[f] + idx */
  int64_t pos; /* position in `indices` */
  int32_t f;
  int32_t b;
} contrib_t;

static int cmp_contrib(const void* a, const void* b) {
  const contrib_t* x = (const contrib_t*)a;
  const contrib_t* y = (const contrib_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

/* Backward + fused exact optimizer.  For every table row touched by the batch: g = sum of its
 * contributions in position order (w_i * grad_out[b, cols of f]), then ONE update.
 * hyper = {lr, eps, weight_decay, beta1, beta2}; iteration is the 1-based step for ADAM.
 * state0/state1: per-feature base pointers (same aliasing rules as feat_weights). */
int64_t oracle_tbe_backward(float* const* feat_weights, const int32_t* feat_D,
                            const int32_t* feat_D_offset, const int64_t* feat_rows,
                            const int64_t* feat_row_base, float* const* feat_state0,
                            float* const* feat_state1, int32_t F, int32_t B,
                            const int64_t* indices, int64_t N, const int64_t* offsets,
                            const float* psw, int32_t pooling_mode, const float* grad_out,
                            int64_t grad_stride, int32_t optimizer, const float* hyper,
                            int64_t iteration) {
  const float lr = hyper[0], eps = hyper[1], wd = hyper[2], beta1 = hyper[3], beta2 = hyper[4];
  contrib_t* c = (contrib_t*)malloc(sizeof(contrib_t) * (size_t)(N > 0 ? N : 1));
  int64_t n = 0, bad = 0;
  for (int32_t f = 0; f < F; ++f) {
    for (int32_t b = 0; b < B; ++b) {
      const int64_t s = offsets[(int64_t)f * B + b], e = offsets[(int64_t)f * B + b + 1];
      for (int64_t i = s; i < e; ++i) {
        const int64_t idx = indices[i];
        if (idx < 0 || idx >= feat_rows[f]) {
          ++bad;
          continue;
        }
        c[n].key = feat_row_base[f] + idx;
        c[n].pos = i;
        c[n].f = f;
        c[n].b = b;
        ++n;
      }
    }
  }
  qsort(c, (size_t)n, sizeof(contrib_t), cmp_contrib);
  float* g = (float*)malloc(sizeof(float) * 4096);
  const float bias1 = 1.f - powf(beta1, (float)iteration);
  const float bias2 = 1.f - powf(beta2, (float)iteration);
  int64_t i = 0;
  while (i < n) {
    int64_t j = i;
    const int32_t f0 = c[i].f;
    const int32_t D = feat_D[f0];
    for (int32_t d = 0; d < D; ++d) g[d] = 0.f;
    while (j < n && c[j].key == c[i].key) {
      const int32_t f = c[j].f;
      float w = psw ? psw[c[j].pos] : 1.f;
      if (pooling_mode == POOL_MEAN) {
        const int64_t bag = (int64_t)f * B + c[j].b;
        w = w / (float)(offsets[bag + 1] - offsets[bag]);
      }
      const float* go = pooling_mode == POOL_NONE ? grad_out + c[j].pos * grad_stride
                                                  : grad_out + (int64_t)c[j].b * grad_stride + feat_D_offset[f];
      for (int32_t d = 0; d < D; ++d) g[d] = fmaf(w, go[d], g[d]);
      ++j;
    }
    const int64_t lrow = c[i].key - feat_row_base[f0];
    float* w = feat_weights[f0] + lrow * D;
    if (optimizer == OPT_EXACT_SGD) {
      for (int32_t d = 0; d < D; ++d) w[d] = fmaf(-lr, g[d], w[d]);
    } else if (optimizer == OPT_EXACT_ROWWISE_ADAGRAD) {
      /* public fbgemm: m += mean_d(g^2); w -= lr / (sqrt(m) + eps) * g  (L2 decay folded into g) */
      double ss = 0.0;
      for (int32_t d = 0; d < D; ++d) {
        if (wd != 0.f) g[d] = fmaf(wd, w[d], g[d]);
        ss += (double)g[d] * (double)g[d];
      }
      float* m = feat_state0[f0] + lrow;
      const float m_new = *m + (float)ss / (float)D;
      *m = m_new;
      const float mult = lr / (sqrtf(m_new) + eps);
      for (int32_t d = 0; d < D; ++d) w[d] = fmaf(-mult, g[d], w[d]);
    } else if (optimizer == OPT_EXACT_ADAGRAD) {
      float* m = feat_state0[f0] + lrow * D;
      for (int32_t d = 0; d < D; ++d) {
        m[d] = fmaf(g[d], g[d], m[d]);
        w[d] = w[d] - lr * g[d] / (sqrtf(m[d]) + eps);
      }
    } else if (optimizer == OPT_ADAM) {
      float* m1 = feat_state0[f0] + lrow * D;
      float* m2 = feat_state1[f0] + lrow * D;
      for (int32_t d = 0; d < D; ++d) {
        m1[d] = fmaf(beta1, m1[d], (1.f - beta1) * g[d]);
        m2[d] = fmaf(beta2, m2[d], (1.f - beta2) * g[d] * g[d]);
        w[d] = w[d] - lr * ((m1[d] / bias1) / (sqrtf(m2[d] / bias2) + eps) + wd * w[d]);
      }
    } else if (optimizer == OPT_DENSE_GRAD) {
      float* gw = feat_state0[f0] + lrow * D;
      for (int32_t d = 0; d < D; ++d) gw[d] = g[d];
    }
    i = j;
  }
  free(g);
  free(c);
  return bad;
}

/* torch.ops.fbgemm.asynchronous_complete_cumsum: out[0] = 0, out[i+1] = out[i] + in[i]
 * (torchrec/sparse/jagged_tensor.py:27-36). mode 0 complete, 1 inclusive, 2 exclusive. */
void oracle_cumsum_i32(const int32_t* in, int32_t* out, int64_t n, int32_t mode) {
  uint32_t acc = 0; /* wrap-around like the device's int32 arithmetic */
  if (mode == 0) out[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t x = (uint32_t)in[i];
    if (mode == 2) out[i] = (int32_t)acc;
    acc += x;
    if (mode == 0) out[i + 1] = (int32_t)acc;
    if (mode == 1) out[i] = (int32_t)acc;
  }
}
void oracle_cumsum_i64(const int64_t* in, int64_t* out, int64_t n, int32_t mode) {
  uint64_t acc = 0;
  if (mode == 0) out[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t x = (uint64_t)in[i];
    if (mode == 2) out[i] = (int64_t)acc;
    acc += x;
    if (mode == 0) out[i + 1] = (int64_t)acc;
    if (mode == 1) out[i] = (int64_t)acc;
  }
}

/* torch.ops.fbgemm.permute_2D_sparse_data on int64 lengths; values/weights are opaque
 * elements of val_size / w_size bytes.  Returns the permuted total; if out_values is NULL only
 * lengths are produced (size query). */
int64_t oracle_permute_2d(const int32_t* permute, int32_t T_in, int32_t T_out, int32_t B,
                          const int64_t* lengths, int64_t* out_lengths, const char* values,
                          char* out_values, int32_t val_size, const char* weights,
                          char* out_weights, int32_t w_size) {
  int64_t* in_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)T_in * B + 1));
  in_off[0] = 0;
  for (int64_t i = 0; i < (int64_t)T_in * B; ++i) in_off[i + 1] = in_off[i] + lengths[i];
  int64_t o = 0;
  for (int32_t t = 0; t < T_out; ++t) {
    for (int32_t b = 0; b < B; ++b) {
      const int64_t src_seg = (int64_t)permute[t] * B + b;
      const int64_t len = lengths[src_seg];
      out_lengths[(int64_t)t * B + b] = len;
      if (out_values) {
        memcpy(out_values + o * val_size, values + in_off[src_seg] * val_size, (size_t)(len * val_size));
        if (weights) memcpy(out_weights + o * w_size, weights + in_off[src_seg] * w_size, (size_t)(len * w_size));
      }
      o += len;
    }
  }
  free(in_off);
  return o;
}

/* torch.ops.fbgemm.block_bucketize_sparse_features (int64 everywhere).
 * Follows torchrec/distributed/tests/test_utils.py:83-236: bucket = idx / block_sizes[f],
 * new idx = idx % block_sizes[f], output ordered (bucket, f, b), stable inside a bag; indices
 * whose bucket >= my_size are dropped.  new_pos (optional) = position inside the source bag;
 * unbucketize_permute (optional) = destination of each source element (-1 if dropped). */
void oracle_block_bucketize(const int64_t* lengths, int64_t lengths_size, const int64_t* indices,
                            const int64_t* block_sizes, int32_t F, int32_t my_size,
                            const float* weights, int64_t* new_lengths, int64_t* new_indices,
                            float* new_weights, int64_t* new_pos, int64_t* unbucketize_permute) {
  const int64_t B = lengths_size / F;
  const int64_t nl = lengths_size * my_size;
  int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(lengths_size + 1));
  off[0] = 0;
  for (int64_t i = 0; i < lengths_size; ++i) off[i + 1] = off[i] + lengths[i];
  for (int64_t i = 0; i < nl; ++i) new_lengths[i] = 0;
  for (int64_t bag = 0; bag < lengths_size; ++bag) {
    const int64_t blk = block_sizes[bag / B];
    for (int64_t i = off[bag]; i < off[bag + 1]; ++i) {
      const int64_t p = (int64_t)((uint64_t)indices[i] / (uint64_t)blk);
      if (p < my_size) new_lengths[p * lengths_size + bag] += 1;
    }
  }
  int64_t* noff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nl + 1));
  noff[0] = 0;
  for (int64_t i = 0; i < nl; ++i) noff[i + 1] = noff[i] + new_lengths[i];
  for (int64_t bag = 0; bag < lengths_size; ++bag) {
    const int64_t blk = block_sizes[bag / B];
    for (int64_t i = off[bag]; i < off[bag + 1]; ++i) {
      const uint64_t idx = (uint64_t)indices[i];
      const int64_t p = (int64_t)(idx / (uint64_t)blk);
      if (p >= my_size) {
        if (unbucketize_permute) unbucketize_permute[i] = -1;
        continue;
      }
      const int64_t dst = noff[p * lengths_size + bag]++;
      new_indices[dst] = (int64_t)(idx % (uint64_t)blk);
      if (weights) new_weights[dst] = weights[i];
      if (new_pos) new_pos[dst] = i - off[bag];
      if (unbucketize_permute) unbucketize_permute[i] = dst;
    }
  }
  free(noff);
  free(off);
}

/* Pooled all-to-all layout: recv[src][B_local][D_src] -> out[B_local, sum D_src] (* scale)
 * (torchrec/distributed/comm_ops.py:555-561); pack is the inverse (:418-428). */
void oracle_a2a_pooled_unpack(const float* recv, float* out, const int32_t* dims, int32_t W,
                              int32_t B_local, int32_t D_total, float scale) {
  int64_t slab = 0;
  int32_t col = 0;
  for (int32_t r = 0; r < W; ++r) {
    for (int32_t b = 0; b < B_local; ++b)
      for (int32_t d = 0; d < dims[r]; ++d)
        out[(int64_t)b * D_total + col + d] = recv[slab + (int64_t)b * dims[r] + d] * scale;
    slab += (int64_t)B_local * dims[r];
    col += dims[r];
  }
}
void oracle_a2a_pooled_pack(const float* grad, float* send, const int32_t* dims, int32_t W,
                            int32_t B_local, int32_t D_total, float scale) {
  int64_t slab = 0;
  int32_t col = 0;
  for (int32_t r = 0; r < W; ++r) {
    for (int32_t b = 0; b < B_local; ++b)
      for (int32_t d = 0; d < dims[r]; ++d)
        send[slab + (int64_t)b * dims[r] + d] = grad[(int64_t)b * D_total + col + d] * scale;
    slab += (int64_t)B_local * dims[r];
    col += dims[r];
  }
}

void oracle_jagged_2d_to_dense(const float* values, const int64_t* offsets, int32_t B, int32_t D,
                               int32_t max_L, float* dense) {
  for (int32_t b = 0; b < B; ++b)
    for (int32_t l = 0; l < max_L; ++l)
      for (int32_t d = 0; d < D; ++d) {
        const int64_t len = offsets[b + 1] - offsets[b];
        dense[((int64_t)b * max_L + l) * D + d] = l < len ? values[(offsets[b] + l) * D + d] : 0.f;
      }
}

void oracle_offsets_range(const int64_t* offsets, int64_t n, int64_t range_size, int64_t* out) {
  int64_t k = 0;
  for (int64_t i = 0; i < range_size; ++i) {
    while (k + 1 < n && offsets[k + 1] <= i) ++k;
    out[i] = i - offsets[k];
  }
}

/* Mixed table-wise + row-wise exchange (see include/tbe_hip.h tbe_pooled_exchange_*):
 * restates All2All_Pooled_Wait split+cat (comm_ops.py:555-561) for table-wise features and the
 * reduce_scatter sum (comm_ops.py:848-930) for row-wise features, summed in rank order. */
void oracle_pooled_exchange_unpack(const float* recv, float* out, const int32_t* feat_out_col,
                                   const int32_t* feat_src, const int32_t* feat_slab_col,
                                   const int64_t* slab_offset, const int32_t* slab_stride, int32_t Fg,
                                   int32_t W, int32_t B_local, int32_t D_total, float scale) {
  for (int32_t b = 0; b < B_local; ++b)
    for (int32_t g = 0; g < Fg; ++g)
      for (int32_t c = feat_out_col[g]; c < feat_out_col[g + 1]; ++c) {
        const int32_t within = feat_slab_col[g] + (c - feat_out_col[g]);
        float v;
        if (feat_src[g] < -1) continue; /* replicated feature: columns left untouched */
        if (feat_src[g] >= 0) {
          const int32_t r = feat_src[g];
          v = recv[slab_offset[r] + (int64_t)b * slab_stride[r] + within];
        } else {
          v = recv[slab_offset[0] + (int64_t)b * slab_stride[0] + within];
          for (int32_t r = 1; r < W; ++r) v += recv[slab_offset[r] + (int64_t)b * slab_stride[r] + within];
        }
        out[(int64_t)b * D_total + c] = v * scale;
      }
}
void oracle_pooled_exchange_pack(const float* grad, float* send, const int32_t* feat_out_col,
                                 const int32_t* feat_src, const int32_t* feat_slab_col,
                                 const int64_t* slab_offset, const int32_t* slab_stride, int32_t Fg,
                                 int32_t W, int32_t B_local, int32_t D_total, float scale) {
  for (int32_t b = 0; b < B_local; ++b)
    for (int32_t g = 0; g < Fg; ++g)
      for (int32_t c = feat_out_col[g]; c < feat_out_col[g + 1]; ++c) {
        const int32_t within = feat_slab_col[g] + (c - feat_out_col[g]);
        const float v = grad[(int64_t)b * D_total + c] * scale;
        if (feat_src[g] < -1) continue;
        if (feat_src[g] >= 0) {
          const int32_t r = feat_src[g];
          send[slab_offset[r] + (int64_t)b * slab_stride[r] + within] = v;
        } else {
          for (int32_t r = 0; r < W; ++r) send[slab_offset[r] + (int64_t)b * slab_stride[r] + within] = v;
        }
      }
}

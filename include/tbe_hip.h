/*
 * tbe_hip.h — C ABI of the MI355X (gfx950) sharded-embedding hot path.
 *
 * This is the drop-in boundary.  The reference (samiwilf/torchrec-oldfork) has no C
 * ABI of its own for this path: it calls the un-vendored `fbgemm_gpu` Python module /
 * `torch.ops.fbgemm.*` dispatcher ops (third_party/fbgemm is an empty submodule,
 * .gitmodules:1-4).  Every entry point below replaces one of those call sites; the
 * reference file:line it serves is cited on each declaration.  The Python host side
 * (`torchrec-oldfork_amd/fbgemm_gpu/`) binds this header with ctypes and re-exposes
 * the reference's names (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - Every pointer is a DEVICE-visible address (hipMalloc, or pinned-host memory
 *    mapped into the GPU for EmbeddingLocation.MANAGED/HOST tables) unless marked
 *    `host`.  No torch types; no allocation; no host synchronisation: every call
 *    only enqueues kernels on `stream` (a hipStream_t passed as void*), so callers
 *    may capture them into a hipGraph.
 *  - Return value: 0 on success, negative TBE_ERR_* otherwise;
 *    `tbe_last_error()` gives a thread-local message.
 *  - "feature metadata" arrays (feat_*) have one entry per FEATURE (not per table):
 *    feature f of the KeyedJaggedTensor looks up table feature_table_map[f]; the
 *    host resolves that indirection once at module construction.
 *  - Jagged layout is the reference's (torchrec/sparse/jagged_tensor.py:614-1081):
 *    `offsets[f*B + b] .. offsets[f*B + b + 1]` delimits bag (feature f, sample b)
 *    inside `indices`; offsets has F*B+1 entries.
 */
#ifndef TBE_HIP_H_
#define TBE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBE_OK 0
#define TBE_ERR_INVALID_ARGUMENT (-1)
#define TBE_ERR_LAUNCH (-2)
#define TBE_ERR_WORKSPACE (-3)
#define TBE_ERR_UNSUPPORTED (-4)

/* fbgemm_gpu.split_table_batched_embeddings_ops.PoolingMode
 * (torchrec/distributed/batched_embedding_kernel.py:18-25, embedding_configs.py:59-73) */
#define TBE_POOL_SUM 0
#define TBE_POOL_MEAN 1
#define TBE_POOL_NONE 2

/* fbgemm_gpu.split_embedding_configs.EmbOptimType as used through fused_params
 * (torchrec/distributed/tests/test_fused_optim.py:52-58, examples/bert4rec/bert4rec_main.py:488-491).
 * TBE_OPT_DENSE_GRAD is the DenseTableBatchedEmbeddingBagsCodegen backward
 * (batched_embedding_kernel.py:677-704): the coalesced row gradient is written to a
 * dense gradient table instead of being applied. */
#define TBE_OPT_EXACT_SGD 0
#define TBE_OPT_EXACT_ROWWISE_ADAGRAD 1
#define TBE_OPT_ADAM 2
#define TBE_OPT_EXACT_ADAGRAD 3
#define TBE_OPT_DENSE_GRAD 100

/* An id with this value is skipped silently by every lookup (no row, no bounds error): how the row cache hands
 * "this row belongs to another rank's shard" to the lookup kernels. */
#define TBE_ID_SKIP INT64_MIN

#define TBE_FLAG_UNIFORM_ALIGNED 1
#define TBE_FLAG_WEIGHTED 2 /* per_sample_weights will be / are given (backward: selects the sort payload layout) */

/* Hyper-parameters of the fused optimizer, passed by value. */
typedef struct tbe_optimizer_args {
  int32_t optimizer;     /* TBE_OPT_* */
  float learning_rate;   /* set_learning_rate(), batched_embedding_kernel.py:250-257 */
  float eps;
  float weight_decay;
  float beta1;
  float beta2;
  int64_t iteration;     /* 1-based step count, used by ADAM bias correction */
} tbe_optimizer_args;

const char* tbe_last_error(void);
int32_t tbe_abi_version(void); /* 3 */

/* Faults that make RESULTS wrong without failing a call (everything here only enqueues kernels, so a kernel that
 * gives up cannot fail the call that launched it).  Today there is one: a spin-wait of the pair sort (backward, row
 * cache prefetch) that outlived its bound because a predecessor workgroup never published — the sorted order, and
 * with it the row updates, are garbage then.  Kernels report into ONE line of GPU-mapped pinned host memory, so this
 * call neither synchronises nor touches a stream: it returns what has arrived so far (everything from work whose
 * completion the caller has observed).  Must stay 0; the Python host side raises on any increase at its check points
 * (optimizer step, flush(), bounds_check_errors(), split_embedding_weights()) — the reference's counterpart of those
 * points: torchrec/distributed/batched_embedding_kernel.py:250-257, 563.  On a box without a HIP device the word is
 * plain host memory (nothing can write it but tbe_debug_inject_fault_host). */
int tbe_fault_status(int64_t* sort_giveups);
/* test hooks: the host-side / device-side write of one give-up without a sort that hangs */
int tbe_debug_inject_fault_host(void);
int tbe_debug_inject_sort_giveup(void* stream);

/* Optional kernel timing for the measurement harness (bench.py `roofline`): when enabled the
 * library brackets its dominant kernels with HIP events recorded on the launch stream.
 * tbe_profile_read(slot) synchronises on the recorded events, returns the summed duration and
 * launch count since the last read, and clears the slot.  Never enabled by the product path. */
#define TBE_PROFILE_FWD_KERNEL 0        /* tbe_fwd_*_kernel, one launch per forward call */
#define TBE_PROFILE_BWD_UPDATE_KERNEL 1 /* bwd_update_kernel, one launch per backward call */
#define TBE_PROFILE_BWD_TOTAL 2         /* fused call: linearize + sort + update + fix-up; apply call: update + fix-up */
#define TBE_PROFILE_BWD_PREPARE 3       /* linearize + sort (gradient independent) */
#define TBE_PROFILE_NUM_SLOTS 4
int tbe_profile_enable(int32_t on);
int tbe_profile_read(int32_t slot, double* total_ms, int64_t* count);
/* table rows updated by backward launches since the last read (distinct rows per batch = the U of
 * SURVEY.md §8d's byte formula); synchronises the device. */
int tbe_profile_read_rows(int64_t* rows_updated);

/* ------------------------------------------------------------------------------------
 * TBE forward (pooled): replaces SplitTableBatchedEmbeddingBagsCodegen.__call__ /
 * DenseTableBatchedEmbeddingBagsCodegen.__call__ as called at
 * torchrec/distributed/batched_embedding_kernel.py:546-554.
 *
 *   out[b * out_row_stride + feat_out_offset[f] + d] = sum_{i in bag(f,b)} w_i * W_f[indices[i], d]
 *   (w_i = per_sample_weights[i] or 1; MEAN divides by the bag length)
 *
 * feat_weights   [F] device array of table base addresses (const float*), one per feature
 * feat_D         [F] embedding dim of the feature's table
 * feat_out_offset [F] int64 element offset of the feature's block for sample 0.  The reference layout
 *                [B, sum D] is offset = prefix sum of feat_D, stride = sum D; an all-to-all-ready
 *                layout [dst rank][B_local][D_local] (removes the split+cat copies of
 *                torchrec/distributed/comm_ops.py:555-561 / :418-428) is expressed by making each
 *                (src rank, feature) pair a pseudo-feature with offset = rank*B_local*D_local + col.
 * feat_rows      [F] number of rows of the feature's table (bounds check)
 * indices [N] int64, offsets [F*B+1] int64, per_sample_weights [N] float or NULL
 * out: float buffer addressed as above; out_row_stride in elements
 * bounds_errors  optional device int32 counter: incremented for every index outside
 *                [0, rows); such an index contributes a zero row (never dereferenced).  Also incremented for
 *                every bag whose offsets are malformed (start < 0, end > N, start > end): such a bag is
 *                empty in forward and contributes nothing in backward — no memory is touched through it.
 * feat_pooling   optional [F] int32 (TBE_POOL_SUM / TBE_POOL_MEAN per feature), honoured when pooling_mode is
 *                TBE_POOL_MEAN: tables of both pooling types in ONE lookup (the reference builds one TBE per
 *                pooling type and concatenates: embedding_sharding.py:393-490, embedding_lookup.py:219-253).
 *                NULL = every feature pools as pooling_mode says.  Same parameter in tbe_backward_*.
 * feat_window    optional [2F] int64: (first global row held, global rows of the table) per feature, for
 *                row-wise shards whose ids arrive un-bucketized (the reference bucketizes instead:
 *                torchrec/distributed/embedding_sharding.py:121-184, sharding/rw_sharding.py:229-236).  Ids are
 *                then GLOBAL rows: those in [first, first + feat_rows) are looked up at id - first, other
 *                ids in [0, global rows) belong to another rank's shard and are skipped SILENTLY, ids outside
 *                [0, global rows) count as bounds errors.  NULL = every feature holds its whole table.
 *                The same parameter of tbe_backward_* and tbe_cache_prefetch means the same.
 * ---------------------------------------------------------------------------------- */
int tbe_forward_pooled_f32(const uint64_t* feat_weights, const int32_t* feat_D,
                           const int64_t* feat_out_offset, const int64_t* feat_rows, int32_t F,
                           int32_t B, int32_t max_D, const int64_t* indices,
                           int64_t N, const int64_t* offsets, const float* per_sample_weights,
                           int32_t pooling_mode, const int32_t* feat_pooling, float* out, int64_t out_row_stride,
                           int32_t* bounds_errors, const int64_t* feat_window, void* stream);

/* TBE forward, PoolingMode.NONE (sequence / unpooled): out[i, :] = W_f(i)[indices[i], :]
 * with f(i) the feature whose offsets range contains i.  All features must share one
 * dim D.  Replaces batched_embedding_kernel.py:332-335 (BatchedFusedEmbedding.forward). */
int tbe_forward_nobag_f32(const uint64_t* feat_weights, const int64_t* feat_rows, int32_t F,
                          int32_t B, int32_t D, const int64_t* indices, int64_t N,
                          const int64_t* offsets, float* out, int32_t* bounds_errors,
                          void* stream);

/* ------------------------------------------------------------------------------------
 * TBE backward + fused "exact" optimizer: the autograd backward of the call above
 * plus the in-backward optimizer fbgemm fuses into it (wired by
 * batched_embedding_kernel.py:604-665 BatchedFusedEmbeddingBag and :53-257
 * EmbeddingFusedOptimizer).  "Exact" = all contributions of one batch to the same
 * table row are summed first (deterministic, position order), then ONE update is
 * applied to the row.
 *
 * feat_row_base [F]  global row number of row 0 of the feature's table (features that
 *                    share a table share the base); key_bits = bits needed to hold the
 *                    largest global row number.
 * feat_state0/1 [F]  base addresses of optimizer state of the feature's table
 *                    (rowwise Adagrad: float[rows] momentum1; ADAM: float[rows*D] m, v;
 *                    DENSE_GRAD: state0 = float[rows*D] dense gradient table). May be NULL
 *                    for SGD.
 * grad_out: same addressing as the forward output (feat_out_offset, grad_row_stride) when pooled,
 *           or [N, D] rows (grad_row_stride = D) for pooling_mode NONE.
 * flags: TBE_FLAG_UNIFORM_ALIGNED = the host asserts that every feature has dim == max_D
 *        (a multiple of 4) and that every table / state base, out offset and the gradient rows
 *        are 16-B aligned; enables the specialised kernels.  0 is always correct.
 * workspace: at least tbe_backward_workspace_bytes(N, F, B, max_D, key_bits) bytes.
 * Limits: F*B < 2^32 and N < 2^29 ids per call (the pair sort counts in 29 bits); tbe_backward_workspace_bytes
 *         returns 0 and the entry points TBE_ERR_INVALID_ARGUMENT beyond them, before anything is launched.
 *         The same N < 2^29 holds for tbe_cache_prefetch and tbe_sort_pairs.
 * ---------------------------------------------------------------------------------- */
size_t tbe_backward_workspace_bytes(int64_t N, int32_t F, int32_t B, int32_t max_D,
                                    int32_t key_bits);

int tbe_backward_fused_f32(const uint64_t* feat_weights, const int32_t* feat_D,
                           const int64_t* feat_out_offset, const int64_t* feat_rows,
                           const int64_t* feat_row_base, const uint64_t* feat_state0,
                           const uint64_t* feat_state1, int32_t F, int32_t B,
                           int32_t max_D, int32_t key_bits, const int64_t* indices, int64_t N,
                           const int64_t* offsets, const float* per_sample_weights,
                           int32_t pooling_mode, const int32_t* feat_pooling, const float* grad_out,
                           int64_t grad_row_stride, tbe_optimizer_args opt, int32_t flags,
                           void* workspace, size_t workspace_bytes, int32_t* bounds_errors, const int64_t* feat_window,
                           void* stream);

/* The backward in two phases, so that the gradient-independent half (linearize + stable sort of
 * the batch's row keys) can be enqueued on a side stream right after the forward call and overlap
 * the dense MLPs; `apply` then needs only update + fix-up once grad_out exists.
 * tbe_backward_fused_f32 == prepare followed by apply on one stream.  `apply` must receive the
 * workspace a `prepare` call filled for the same (indices, offsets, N, F, B, max_D, key_bits).
 * flags (prepare): TBE_FLAG_WEIGHTED when `apply` will be given per_sample_weights (the sort then
 * carries (bag, position) payloads instead of bag numbers); `apply` must get the same bit. */
int tbe_backward_prepare(const int64_t* feat_rows, const int64_t* feat_row_base, int32_t F,
                         int32_t B, int32_t max_D, int32_t key_bits, const int64_t* indices,
                         int64_t N, const int64_t* offsets, int32_t pooling_mode, int32_t flags,
                         void* workspace, size_t workspace_bytes, int32_t* bounds_errors, const int64_t* feat_window,
                         void* stream);

/* The stable pair sort the backward and the row cache use, exposed for tests and micro-benchmarks
 * (no reference counterpart: fbgemm's backward sorts with cub inside the absent submodule).
 * Sorts n (key, payload) pairs on the low key_bits bits of the keys; equal keys keep input order.
 * key_bytes / payload_bytes: 4 or 8.  keys / payload are overwritten with the sorted result; keys_tmp /
 * payload_tmp are scratch of the same size.  workspace >= tbe_sort_pairs_workspace_bytes(n, key_bits).
 * tbe_debug_sort_timeouts: tbe_fault_status after a device synchronisation (tests). */
size_t tbe_sort_pairs_workspace_bytes(int64_t n, int32_t key_bits);
int tbe_sort_pairs(void* keys, void* keys_tmp, void* payload, void* payload_tmp, int64_t n,
                   int32_t key_bits, int32_t key_bytes, int32_t payload_bytes, void* workspace,
                   size_t workspace_bytes, void* stream);
int tbe_debug_sort_timeouts(int64_t* count);
/* Development aid: device buffer of int64 [7 passes][256 segments][8] that tbe_sort_pairs' pass kernels fill
 * with 100 MHz wall-clock stamps per phase (NULL switches it off; tools/sstamps.py). */
int tbe_debug_set_sort_stamps(void* device_buffer);
/* Development aid of the same kind for the interaction forward (TBE_INTERACTION_ABLATION=6): 8 workgroups x 4 waves x
 * 32 samples x 6 cycle-counter stamps (uint64), NULL switches it off. */
int tbe_debug_set_interaction_stamps(void* device_buffer);
int tbe_backward_apply_f32(const uint64_t* feat_weights, const int32_t* feat_D,
                           const int64_t* feat_out_offset, const int64_t* feat_rows,
                           const int64_t* feat_row_base, const uint64_t* feat_state0,
                           const uint64_t* feat_state1, int32_t F, int32_t B,
                           int32_t max_D, int32_t key_bits, const int64_t* indices, int64_t N,
                           const int64_t* offsets, const float* per_sample_weights,
                           int32_t pooling_mode, const int32_t* feat_pooling, const float* grad_out,
                           int64_t grad_row_stride, tbe_optimizer_args opt, int32_t flags,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * HBM row cache for EmbeddingLocation.MANAGED_CACHING tables (the `batched_fused_uvm_caching`
 * compute kernel: torchrec/distributed/embedding_types.py:57-76; `flush()` before weights are
 * read: batched_embedding_kernel.py:563,664; default cache_load_factor 0.2:
 * planner/constants.py:26).  The tables stay in pinned, GPU-mapped host memory; a 64-way
 * set-associative cache in HBM holds the rows in use.  The TBE kernels above are unchanged:
 * tbe_cache_prefetch rewrites the ids of cached features into slot numbers of ONE pseudo-table
 * [num_sets*64 cache slots | staging_cap staging slots] (`rows`, row stride `row_stride`), and the
 * caller points the feat_* metadata of those features at it (feat_weights = rows, feat_rows =
 * num_sets*64 + staging_cap, feat_state0 = state).  Rows that find no evictable way are held in
 * the staging slots for this batch; tbe_cache_writeback_staging copies them home after backward.
 * Results are identical to an uncached table.
 *
 *   tags [slots] int64  cached-row key (tab_key_base[t] + local row), -1 = empty
 *   lru  [slots] int32  iteration of last use, -1 = never; `iteration` must increase per prefetch
 *   rows [(slots + staging_cap) * row_stride] float;  state: same slots, rowwise optimizer state or NULL
 *   staging_keys [staging_cap] int64;  counters [8] int32: 0 staging rows of this batch, 1 hits,
 *   2 misses, 3 evictions (write-backs), 4 unique cached rows of this batch, 5 misses of this batch
 *   tab_* : the cached tables (num_tables entries; tab_key_base has num_tables+1 = prefix sum of rows)
 * tbe_cache_prefetch: feat_cached_table[f] = index into tab_* or -1 (ids of such features are copied
 *   unchanged); key_bits = bits of the total cached rows; staging_cap must be >= N.
 * tbe_cache_flush: every valid slot -> host table; invalidate != 0 also empties the cache.
 * ---------------------------------------------------------------------------------- */
typedef struct tbe_cache_desc {
  int64_t* tags;
  int32_t* lru;
  float* rows;
  float* state;
  int64_t* staging_keys;
  int32_t* counters;
  const int64_t* tab_key_base;
  const uint64_t* tab_weights;
  const uint64_t* tab_state;
  const int32_t* tab_D;
  int32_t num_sets;
  int32_t row_stride;
  int32_t staging_cap;
  int32_t num_tables;
} tbe_cache_desc;
size_t tbe_cache_prefetch_workspace_bytes(int64_t N, int32_t key_bits);
int tbe_cache_prefetch(const tbe_cache_desc* desc, const int32_t* feat_cached_table,
                       const int64_t* feat_rows, int32_t F, int32_t B, const int64_t* indices,
                       int64_t N, const int64_t* offsets, int32_t key_bits, int32_t iteration,
                       int64_t* remapped_indices, void* workspace, size_t workspace_bytes,
                       const int64_t* feat_window, void* stream);
int tbe_cache_writeback_staging(const tbe_cache_desc* desc, void* stream);
int tbe_cache_flush(const tbe_cache_desc* desc, int32_t invalidate, void* stream);

/* ------------------------------------------------------------------------------------
 * torch.ops.fbgemm.asynchronous_complete_cumsum (torchrec/sparse/jagged_tensor.py:35-36):
 * out[0] = 0, out[i+1] = sum(in[0..i]); out has n+1 entries.  elem_size 4 (int32) or
 * 8 (int64).  workspace >= tbe_cumsum_workspace_bytes(n).
 * `mode`: 0 = complete (n+1 outputs), 1 = inclusive (n outputs), 2 = exclusive (n outputs)
 * ---------------------------------------------------------------------------------- */
size_t tbe_cumsum_workspace_bytes(int64_t n);
int tbe_cumsum(const void* in, void* out, int64_t n, int32_t elem_size, int32_t mode,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * torch.ops.fbgemm.permute_2D_sparse_data (torchrec/sparse/jagged_tensor.py:946-952,
 * torchrec/distributed/dist_data.py:257-263, comm_ops.py:633-639,691-697).
 * Two steps so that the host can size the outputs without a sync when it already
 * knows permuted_lengths_sum:
 *  1. tbe_permute_2d_lengths: out_lengths[t', b] = lengths[permute[t'], b]
 *     and in_offsets / out_offsets = exclusive complete cumsums (T*B+1 / T'*B+1 entries,
 *     int64) written to caller buffers.
 *  2. tbe_permute_2d_data: for each (t', b) copy the segment of `values` (and `weights`).
 * lengths elem size 4 or 8; values/weights element size 1, 2, 4 or 8 bytes (opaque copy).
 * ---------------------------------------------------------------------------------- */
size_t tbe_permute_2d_workspace_bytes(int32_t T_in, int32_t T_out, int32_t B);
int tbe_permute_2d_lengths(const int32_t* permute, int32_t T_in, int32_t T_out, int32_t B,
                           const void* lengths, int32_t len_elem_size, void* out_lengths,
                           int64_t* in_offsets, int64_t* out_offsets, void* workspace,
                           size_t workspace_bytes, void* stream);
int tbe_permute_2d_data(const int32_t* permute, int32_t T_out, int32_t B,
                        const int64_t* in_offsets, const int64_t* out_offsets,
                        const void* values, void* out_values, int32_t val_elem_size,
                        const void* weights, void* out_weights, int32_t w_elem_size,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * torch.ops.fbgemm.block_bucketize_sparse_features
 * (torchrec/distributed/embedding_sharding.py:121-184; python reference of the result:
 * torchrec/distributed/tests/test_utils.py:83-236).
 *   bucket = idx / block_sizes[f];  new idx = idx % block_sizes[f]
 *   new_lengths[bucket*F*B + f*B + b] counts; new_indices ordered (bucket, f, b), stable.
 *   indices whose bucket >= my_size are dropped (as the python reference does).
 * lengths [F*B] (elem 4|8), indices [N] (elem 4|8), block_sizes [F] (same elem size as indices)
 * weights [N] float or NULL; out pos [N] (same elem as indices) if bucketize_pos;
 * unbucketize_permute [N] (same elem as indices) if sequence.
 * workspace >= tbe_bucketize_workspace_bytes(F*B, my_size).
 * new_offsets_out: optional int64 [my_size*F*B + 1] complete cumsum of new_lengths
 *   (lets the host read per-bucket totals without recomputing).
 * ---------------------------------------------------------------------------------- */
size_t tbe_bucketize_workspace_bytes(int64_t lengths_size, int32_t my_size);
int tbe_block_bucketize(const void* lengths, int32_t len_elem_size, int64_t lengths_size,
                        const void* indices, int32_t idx_elem_size, int64_t N,
                        const void* block_sizes, int32_t F, int32_t my_size,
                        const float* weights, int32_t bucketize_pos, int32_t sequence,
                        void* new_lengths, void* new_indices, float* new_weights,
                        void* new_pos, void* unbucketize_permute, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Pooled-embedding all-to-all layout ops.  The reference does these with
 * torch split + cat copies (torchrec/distributed/comm_ops.py:555-561 forward,
 * :418-428 `_recat_pooled_embedding_grad_out` backward, flagged slow by its own TODO :417).
 *   tbe_a2a_pooled_unpack : recv[src][B_local][D_src] slabs -> out[B_local, sum D_src]
 *   tbe_a2a_pooled_pack   : grad[B_local, sum D_src]       -> send[src][B_local][D_src]
 * dim_sum_per_rank [W] int32 device array; dims_multiple_of_4 != 0 asserts every entry is a
 * multiple of 4 (enables 16-B accesses); scale multiplies every element (the 1/W gradient
 * division of comm_ops.py:527-528 can be fused here).
 * ---------------------------------------------------------------------------------- */
int tbe_a2a_pooled_unpack(const float* recv, float* out, const int32_t* dim_sum_per_rank,
                          int32_t W, int32_t B_local, int32_t D_total,
                          int32_t dims_multiple_of_4, float scale, void* stream);
int tbe_a2a_pooled_pack(const float* grad, float* send, const int32_t* dim_sum_per_rank,
                        int32_t W, int32_t B_local, int32_t D_total,
                        int32_t dims_multiple_of_4, float scale, void* stream);

/* ------------------------------------------------------------------------------------
 * Mixed table-wise + row-wise pooled exchange through ONE all-to-all (xGMI is point-to-point:
 * an all-to-all drives all 7 links, a ring reduce-scatter is bound by one).  Replaces, on the
 * receiver, All2All_Pooled_Wait's split+cat (torchrec/distributed/comm_ops.py:555-561), the
 * reduce_scatter of row-wise partial pools (comm_ops.py:848-930) and the cross-sharding-type
 * torch.cat (torchrec/distributed/embeddingbag.py:212-223); `pack` is the gradient transpose
 * (comm_ops.py:418-428 recat + :922-927 all_gather), with the 1/W division fused as `scale`.
 *   exchange buffer: W slabs, slab r = [B_local][slab_stride[r]] at element slab_offset[r]
 *   matrix:          [B_local, D_total], global feature g occupies columns
 *                    [feat_out_col[g], feat_out_col[g+1])
 *   feat_src[g] >= 0 : table-wise, lives in slab feat_src[g] at column feat_slab_col[g]
 *   feat_src[g] == -1: row-wise, every slab holds a partial pool at column feat_slab_col[g];
 *                      unpack sums them in rank order 0..W-1, pack broadcasts the gradient.
 *   feat_src[g] <= -2: replicated (data-parallel) feature: its columns are skipped by both
 *                      kernels (a local lookup writes / reads them, see embeddingbag.py).
 * all_multiple_of_4 != 0 asserts every column offset / dim / stride is a multiple of 4.
 * ---------------------------------------------------------------------------------- */
int tbe_pooled_exchange_unpack(const float* recv, float* out, const int32_t* feat_out_col,
                               const int32_t* feat_src, const int32_t* feat_slab_col,
                               const int64_t* slab_offset, const int32_t* slab_stride,
                               int32_t Fg, int32_t W, int32_t B_local, int32_t D_total,
                               int32_t all_multiple_of_4, float scale, void* stream);
int tbe_pooled_exchange_pack(const float* grad, float* send, const int32_t* feat_out_col,
                             const int32_t* feat_src, const int32_t* feat_slab_col,
                             const int64_t* slab_offset, const int32_t* slab_stride,
                             int32_t Fg, int32_t W, int32_t B_local, int32_t D_total,
                             int32_t all_multiple_of_4, float scale, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused DLRM dot interaction (the path's only MFMA user): InteractionArch.forward,
 * torchrec/models/dlrm.py:193-219 — cat + bmm(X, X^T) + triu gather + cat — and its autograd
 * backward, each as one kernel on v_mfma_f32_16x16x4_f32 (exact f32).
 *   dense [B, D], sparse [B, F, D] contiguous, out / grad_out [B, D + (F+1)F/2]:
 *   out[b] = [dense[b] | <X_i, X_j> for i < j in torch.triu_indices(F+1, F+1, 1) order],
 *   X = [dense[b]; sparse[b]].
 * forward: 1 <= F <= 31, D in {16, 32, 64, 128, 256}.  backward: F <= 27, D in {16, 32, 64, 128}.
 * out_row_stride / grad_row_stride (elements, >= D + (F+1)F/2): rows of `out` / `grad_out` may be padded.  The
 * reference's dense rows (stride 479 floats at F = 26, D = 128) are not 16-B aligned; with a stride that is a
 * multiple of 4 and leaves room for the pair block rounded up to 4 (480) the forward stores 16 B per lane and
 * writes the pad columns as zeros, and the GEMMs of the next layer read aligned rows (lda = 480, K = 479).
 * ---------------------------------------------------------------------------------- */
int tbe_dlrm_interaction_forward_f32(const float* dense, const float* sparse, int32_t B,
                                     int32_t F, int32_t D, float* out, int64_t out_row_stride,
                                     void* stream);
int tbe_dlrm_interaction_backward_f32(const float* dense, const float* sparse,
                                      const float* grad_out, int64_t grad_row_stride, int32_t B, int32_t F, int32_t D,
                                      float* grad_dense, float* grad_sparse, void* stream);

/* ReLU backward + bias gradient of one MLP layer in one pass (torchrec/modules/mlp.py:14-170 Perceptron
 * trained through autograd: threshold_backward + sum(0)):
 *   grad_in[b, c] = act[b, c] > 0 ? grad_out[b, c] : 0;   bias_grad[c] = sum_b grad_in[b, c]
 * [B, N] row-major, N a multiple of 4, 16-B aligned; column sums in fixed order (deterministic). */
size_t tbe_relu_backward_bias_grad_workspace_bytes(int64_t B, int32_t N);
int tbe_relu_backward_bias_grad_f32(const float* grad_out, const float* act, int64_t B, int32_t N,
                                    float* grad_in, float* bias_grad, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* Weight gradient of a Linear with ONE output feature (DLRM's last layer, torchrec/models/dlrm.py OverArch:
 * `nn.Linear(layer_sizes[-2], 1)`): out[c] = sum_b w[b] * x[b, c] for x [B, N] row-major, w [B] — the
 * N = 1 GEMM the BLAS libraries run at ~1 % of HBM speed.  N a multiple of 4; fixed summation order. */
size_t tbe_weighted_colsum_workspace_bytes(int64_t B, int32_t N);
int tbe_weighted_colsum_f32(const float* x, const float* w, int64_t B, int32_t N, float* out,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The first stages of the two above alone: the column sums of the row blocks stay in `partial`
 * [tbe_colsum_row_blocks(B, N)][N] (16-B aligned, B > 0) for ONE later tbe_multi_chunk_sum_f32 over every layer of a
 * captured backward. */
int64_t tbe_colsum_row_blocks(int64_t B, int32_t N);
int tbe_relu_backward_bias_partials_f32(const float* grad_out, const float* act, int64_t B, int32_t N,
                                        float* grad_in, float* partial, size_t partial_bytes, void* stream);
int tbe_weighted_colsum_partials_f32(const float* x, const float* w, int64_t B, int32_t N, float* partial,
                                     size_t partial_bytes, void* stream);

/* dst[off_s + i] = scale * sum_{c < chunks_s} src_s[c * numel_s + i] for every segment s of `seg_table` — device int64
 * [nseg][4] = {src address, chunks, numel, dst element offset}; chunk order fixed (deterministic); chunks = 0 writes
 * zeros.  max_numel = the largest numel.  Finishes, in one launch, the split-K weight gradients (chunks = batch slices of
 * the batched GEMM), the bias gradients (chunks = row blocks of the kernels above) and the already complete gradients
 * (chunks = 1) of the dense MLPs' backward, scaled by 1 / world size, into the flat gradient buffer that is all-reduced —
 * what the reference leaves to per-parameter autograd kernels + DistributedDataParallel's bucket copies
 * (torchrec/distributed/model_parallel.py:101-111). */
int tbe_multi_chunk_sum_f32(const int64_t* seg_table, int32_t nseg, int64_t max_numel, float* dst, float scale,
                            void* stream);
/* The same with the table in HOST memory, at most 32 segments: it travels by value in the kernel arguments (no copy the
 * GPU performs later, so the caller may reuse the host buffer at once) — for eager steps, whose sources and destinations
 * change every step.  With dst = NULL the destination offsets are absolute addresses / 4. */
int tbe_multi_chunk_sum_host_table_f32(const int64_t* host_seg_table, int32_t nseg, int64_t max_numel, float* dst,
                                       float scale, void* stream);

/* nn.BCEWithLogitsLoss (mean) forward + gradient in one launch — the loss of the reference's train wrapper
 * (examples/dlrm/modules/dlrm_train.py):  loss = mean_i [max(x_i, 0) - x_i y_i + log1p(exp(-|x_i|))],
 * dlogits_i = (sigmoid(x_i) - y_i) / B  (dlogits may be NULL).  labels: float32 (label_elem_size 4) or int64 (8).
 * Deterministic (fixed block ranges, block partials added in index order by the last block).  workspace:
 * tbe_bce_with_logits_workspace_bytes() bytes, 16-B aligned, ZEROED ONCE by the caller before its first use and then
 * reusable launch after launch (the kernel resets its ticket). */
size_t tbe_bce_with_logits_workspace_bytes(void);
int tbe_bce_with_logits_f32(const float* logits, const void* labels, int32_t label_elem_size, int64_t B, float* loss,
                            float* dlogits, void* workspace, size_t workspace_bytes, void* stream);

/* torch.ops.fbgemm.jagged_2d_to_dense (examples/bert4rec/models/bert4rec.py:394-400):
 * values [N, D] + offsets [B+1] -> dense [B, max_L, D], zero padded / truncated. */
int tbe_jagged_2d_to_dense_f32(const float* values, const int64_t* offsets, int32_t B,
                               int32_t D, int32_t max_L, float* dense, void* stream);

/* Gradient of the above: values_grad [N, D] <- dense_grad [B, max_L, D] (rows beyond max_L get 0). */
int tbe_dense_to_jagged_2d_f32(const float* dense, const int64_t* offsets, int32_t B, int32_t D,
                               int32_t max_L, int64_t N, float* values, void* stream);

/* torch.ops.fbgemm.offsets_range (torchrec/modules/feature_processor.py:65):
 * out[i] = i - offsets[bag(i)] for i in [0, range_size). offsets has n entries (no total). */
int tbe_offsets_range(const int64_t* offsets, int64_t n, int64_t range_size, int64_t* out,
                      void* stream);

/* dst[y, :] = src[rows[y], :] for y in [0, n_rows): rows of `row_bytes` bytes (a multiple of 16, 16-byte aligned bases).
 * The send-order gather of the input exchange when every feature has the same number of ids per batch (the
 * reference permutes the whole KJT: torchrec/distributed/dist_data.py:257-263); `rows` is a device int32 array whose
 * entries the caller guarantees are < src_rows (not checked on the device). */
int tbe_copy_rows(const void* src, int64_t src_rows, const int32_t* rows, int32_t n_rows,
                  int64_t row_bytes, void* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TBE_HIP_H_ */

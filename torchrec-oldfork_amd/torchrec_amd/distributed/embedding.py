"""ShardedEmbeddingCollection: table-wise sharded UNPOOLED ("sequence") embeddings
(torchrec/distributed/embedding.py:293-539, sharding/tw_sequence_sharding.py,
SequenceEmbeddingAllToAll dist_data.py:841-932, All2All_Seq_Req comm_ops.py:608-749).

Data path per step (W ranks, rank r owns the features of its tables):
  ids    : 2-phase exchange (lengths, then values; sizes are data dependent for sequences) — ids
           arrive as [src rank][local feature][sample], which is the order the local TBE
           (PoolingMode.NONE) consumes and produces, so no recat permute is needed;
  lookup : one tbe_forward_nobag_f32 launch -> [N_recv, D];
  output : ONE all-to-all of embedding rows; because the received ids of source w are contiguous, the
           send buffer is the lookup output as is, and what comes back is in the order of the ids this
           rank sent — i.e. already grouped [feature][sample] per destination, from which the
           per-feature JaggedTensors are sliced without a copy when features are sent in
           collection order;
  grads  : the same all-to-all reversed, then the fused TBE backward.  As in the reference, the gradient is
           NOT divided by the world size on this path: GRADIENT_DIVISION only acts on the pooled all-to-all
           (comm_ops.py:527-528) and the reduce-scatter (:883-885); All2All_Seq_Req_Wait.backward
           (comm_ops.py:718-749) sends the gradient as is.  `SEQUENCE_GRADIENT_DIVISION = True` opts into
           the pooled path's convention (sharded == unsharded for a mean loss).

Row-wise tables (sharding/rw_sequence_sharding.py): ids are bucketized by row block with
`fbgemm.block_bucketize_sparse_features(sequence=True)` (embedding_sharding.py:121-184), bucket r goes
to rank r, the looked-up rows come back in bucketized order and `unbucketize_permute` restores the
caller's order with one index_select (dist_data.py:820-827).  Table-wise and row-wise tables of one
collection run as two independent paths over disjoint features.
"""
from typing import Any, Callable, Dict, List, Optional, NamedTuple

import torch
import torch.distributed as dist
from torch import nn

from ..modules.embedding_configs import EmbeddingConfig
from ..sparse.jagged_tensor import JaggedTensor, KeyedJaggedTensor
from .planner import rw_block_size, rw_shard_rows
from .types import Awaitable, LazyAwaitable, NoWait, ParameterSharding, ShardingEnv, ShardingType


SEQUENCE_GRADIENT_DIVISION = False  # reference behaviour (see the module docstring)


class RowBlockSplit(NamedTuple):
    """Ids regrouped by the rank that holds their row: segment (dest, feature, sample) order."""

    lengths: torch.Tensor            # [num_dest * F * B] ids per (dest, feature, sample)
    ids: torch.Tensor                # [N] row numbers INSIDE the destination's block, segment order
    weights: Optional[torch.Tensor]  # [N] per-id weights in the same order, if given
    restore: Optional[torch.Tensor]  # [N] restore[i] = position of the caller's i-th id in `ids`


def split_ids_by_row_block(lengths: torch.Tensor, ids: torch.Tensor, rows_per_block: List[int], num_dest: int,
                           weights: Optional[torch.Tensor] = None, want_restore: bool = True) -> RowBlockSplit:
    """Row-wise input dist, step 1: feature f's id goes to rank id // rows_per_block[f] and becomes
    id % rows_per_block[f] there (block size ceil(rows / W): sharding/rw_sharding.py:229-236).  One launch of the
    bucketize kernel (csrc/sparse_ops.hip, through the `fbgemm::block_bucketize_sparse_features` op the
    reference's KJT-level wrapper calls at embedding_sharding.py:160-168); stable inside a bag."""
    blocks = torch.as_tensor(rows_per_block, dtype=ids.dtype, device=ids.device)
    if blocks.numel() * num_dest == 0 or lengths.numel() % blocks.numel():
        raise ValueError(f"split_ids_by_row_block: {lengths.numel()} lengths do not divide into {blocks.numel()} features")
    new_lengths, new_ids, new_weights, _pos, restore = torch.ops.fbgemm.block_bucketize_sparse_features(
        lengths.reshape(-1), ids, False, want_restore, blocks, num_dest, weights)
    return RowBlockSplit(new_lengths.reshape(-1), new_ids, new_weights, restore)


def _default_seq_tbe_factory(specs, ftm, device, fused_params):
    from fbgemm_gpu.split_table_batched_embeddings_ops import (
        ComputeDevice, EmbeddingLocation, PoolingMode, SplitTableBatchedEmbeddingBagsCodegen)

    return SplitTableBatchedEmbeddingBagsCodegen(
        embedding_specs=[(r, d, EmbeddingLocation.DEVICE, ComputeDevice.CUDA) for r, d in specs],
        feature_table_map=ftm, pooling_mode=PoolingMode.NONE, device=device, **fused_params)


class _SeqExchange(torch.autograd.Function):
    """rows [sum(send_counts), D] -> rows [sum(recv_counts), D]; backward is the reverse exchange."""

    @staticmethod
    def forward(ctx, emb, pg, send_counts, recv_counts):
        ctx.pg, ctx.send_counts, ctx.recv_counts = pg, send_counts, recv_counts
        D = emb.shape[1]
        out = torch.empty((sum(recv_counts), D), dtype=emb.dtype, device=emb.device)
        dist.all_to_all_single(out, emb.contiguous(), list(recv_counts), list(send_counts), group=pg)  # splits in rows
        return out

    @staticmethod
    def backward(ctx, grad):
        D = grad.shape[1]
        W = dist.get_world_size(ctx.pg)
        g = grad.contiguous()
        if SEQUENCE_GRADIENT_DIVISION:
            g = g / W
        out = torch.empty((sum(ctx.send_counts), D), dtype=g.dtype, device=g.device)
        dist.all_to_all_single(out, g, list(ctx.send_counts), list(ctx.recv_counts), group=ctx.pg)
        return out, None, None, None


class ShardedEmbeddingCollection(nn.Module):
    def __init__(self, tables: List[EmbeddingConfig], table_name_to_parameter_sharding: Dict[str, ParameterSharding],
                 env: ShardingEnv, fused_params: Optional[Dict[str, Any]] = None, device: Optional[torch.device] = None,
                 tbe_factory: Optional[Callable] = None) -> None:
        super().__init__()
        self._pg, self._W, self._me = env.process_group, env.world_size, env.rank
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self._W > 1 and self._device.type == "cuda":
            from .comm import exchange_group

            self._pg = exchange_group(env.process_group, self._device)  # (see ShardedEmbeddingBagCollection)
        dims = {c.embedding_dim for c in tables}
        if len(dims) != 1:
            raise ValueError("All tables in a EmbeddingCollection are required to have same embedding dimension.")
        self.embedding_dim = dims.pop()
        self._configs = list(tables)
        self._feature_names: List[str] = []
        g_table: List[int] = []
        for t, c in enumerate(tables):
            for f in (c.feature_names or [c.name]):
                self._feature_names.append(f)
                g_table.append(t)
        owner = []  # owning rank of a table-wise table, -1 for a row-wise table
        for c in tables:
            ps = table_name_to_parameter_sharding[c.name]
            if ps.sharding_type == ShardingType.TABLE_WISE.value:
                owner.append(int(ps.ranks[0]))
            elif ps.sharding_type == ShardingType.ROW_WISE.value:
                owner.append(-1)
            else:
                raise NotImplementedError("sequence embeddings: table_wise and row_wise sharding are implemented")
        Fg = len(self._feature_names)
        self._tw_feats = [g for g in range(Fg) if owner[g_table[g]] >= 0]
        self._rw_feats = [g for g in range(Fg) if owner[g_table[g]] < 0]
        self._init_row_wise(tables, g_table, fused_params, tbe_factory)
        self._local_feats = [[g for g in range(Fg) if owner[g_table[g]] == r] for r in range(self._W)]
        self._send_order = [g for lf in self._local_feats for g in lf]
        self._send_per_rank = [len(lf) for lf in self._local_feats]
        self._F_local = len(self._local_feats[self._me])
        local_tables: List[int] = []
        for g in self._local_feats[self._me]:
            if g_table[g] not in local_tables:
                local_tables.append(g_table[g])
        self._local_table_ids = local_tables
        ftm_local = [local_tables.index(g_table[g]) for g in self._local_feats[self._me]]
        self._emb_module = None
        if local_tables:
            factory = tbe_factory or _default_seq_tbe_factory
            self._emb_module = factory([(tables[t].num_embeddings, tables[t].embedding_dim) for t in local_tables],
                                       ftm_local * self._W, self._device, dict(fused_params or {}))
            for t, w in zip(local_tables, self._emb_module.split_embedding_weights()):
                w.uniform_(tables[t].get_weight_init_min(), tables[t].get_weight_init_max())

    def _init_row_wise(self, tables, g_table, fused_params, tbe_factory) -> None:
        W, me = self._W, self._me
        self._rw_module = None
        self._rw_table_ids: List[int] = []
        for g in self._rw_feats:
            if g_table[g] not in self._rw_table_ids:
                self._rw_table_ids.append(g_table[g])
        if not self._rw_feats:
            return
        ftm = [self._rw_table_ids.index(g_table[g]) for g in self._rw_feats]
        self._rw_row0 = {t: me * rw_block_size(tables[t].num_embeddings, W) for t in self._rw_table_ids}
        self._rw_blocks = [rw_block_size(tables[g_table[g]].num_embeddings, W) for g in self._rw_feats]
        factory = tbe_factory or _default_seq_tbe_factory
        self._rw_module = factory([(rw_shard_rows(tables[t].num_embeddings, W)[me], tables[t].embedding_dim)
                                   for t in self._rw_table_ids], ftm * W, self._device, dict(fused_params or {}))
        for t, w in zip(self._rw_table_ids, self._rw_module.split_embedding_weights()):
            if w.numel():
                w.uniform_(tables[t].get_weight_init_min(), tables[t].get_weight_init_max())

    def local_shards(self) -> Dict[str, torch.Tensor]:
        """table name -> local weight shard (whole table if table-wise, this rank's row block if row-wise)."""
        out: Dict[str, torch.Tensor] = {}
        if self._emb_module is not None:
            out.update({self._configs[t].name: w
                        for t, w in zip(self._local_table_ids, self._emb_module.split_embedding_weights())})
        if self._rw_module is not None:
            out.update({self._configs[t].name: w
                        for t, w in zip(self._rw_table_ids, self._rw_module.split_embedding_weights())})
        return out

    # ---- checkpoint surface (keys as torchrec/distributed/embedding.py: `embeddings.<table>.weight`) ----------
    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False):
        destination = {} if destination is None else destination
        for name, w in self.local_shards().items():
            destination[f"{prefix}embeddings.{name}.weight"] = w if keep_vars else w.detach()
        return destination

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Per table either the local shard or the whole table (cut at this rank's row offset)."""
        row0 = self.local_shard_row_offsets()
        rows = {c.name: c.num_embeddings for c in self._configs}
        for name, w in self.local_shards().items():
            key = f"{prefix}embeddings.{name}.weight"
            if key not in state_dict:
                if strict:
                    missing_keys.append(key)
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(w.shape):
                if src.dim() == 2 and src.shape[0] == rows[name] and src.shape[1] == w.shape[1]:
                    src = src[row0[name]:row0[name] + w.shape[0]]
                else:
                    error_msgs.append(f"size mismatch for {key}: {tuple(src.shape)} vs local {tuple(w.shape)}")
                    continue
            with torch.no_grad():
                w.copy_(src)

    @property
    def fused_optimizer(self):
        from ..optim.keyed import CombinedOptimizer
        from .embeddingbag import EmbeddingFusedOptimizer

        opts = []
        if self._emb_module is not None:
            opts.append(EmbeddingFusedOptimizer(self._emb_module, [self._configs[t].name for t in self._local_table_ids],
                                                key_prefix="embeddings."))
        if self._rw_module is not None:
            opts.append(EmbeddingFusedOptimizer(self._rw_module, [self._configs[t].name for t in self._rw_table_ids],
                                                key_prefix="embeddings."))
        return CombinedOptimizer(opts)

    def local_shard_row_offsets(self) -> Dict[str, int]:
        out = {self._configs[t].name: 0 for t in self._local_table_ids}
        if self._rw_module is not None:
            out.update({self._configs[t].name: self._rw_row0[t] for t in self._rw_table_ids})
        return out

    def _forward_row_wise(self, features: KeyedJaggedTensor) -> Dict[str, JaggedTensor]:
        W, B, pg = self._W, features.stride(), self._pg
        pos = {k: i for i, k in enumerate(features.keys())}
        order = [pos[self._feature_names[g]] for g in self._rw_feats]
        sub = features if order == list(range(len(features.keys()))) else features.permute(order)
        lengths, values = sub.lengths(), sub.values()
        Frw = len(self._rw_feats)
        split = split_ids_by_row_block(lengths, values, self._rw_blocks, W)
        nl, ni, unb = split.lengths, split.ids, split.restore
        if W > 1:
            val_in = nl.view(W, -1).sum(dim=1).cpu().tolist()  # host sync (dist_data.py:396-398)
            recv_l = torch.empty(W * Frw * B, dtype=nl.dtype, device=nl.device)
            dist.all_to_all_single(recv_l, nl, [Frw * B] * W, [Frw * B] * W, group=pg)
            val_out = recv_l.view(W, -1).sum(dim=1).cpu().tolist()
            recv_v = torch.empty(sum(val_out), dtype=ni.dtype, device=ni.device)
            dist.all_to_all_single(recv_v, ni, val_out, val_in, group=pg)
        else:
            recv_l, recv_v, val_in, val_out = nl, ni, None, None
        offsets = torch.ops.fbgemm.asynchronous_complete_cumsum(recv_l).long()
        emb = self._rw_module(recv_v, offsets)
        back = _SeqExchange.apply(emb, pg, val_out, val_in) if W > 1 else emb
        rows = back.index_select(0, unb.long())  # bucketized order -> the caller's order
        opk = sub.offset_per_key()
        return {self._feature_names[g]: JaggedTensor(values=rows[opk[i]:opk[i + 1]], lengths=lengths[i * B:(i + 1) * B])
                for i, g in enumerate(self._rw_feats)}

    def forward(self, features: KeyedJaggedTensor) -> Awaitable[Dict[str, JaggedTensor]]:
        out: Dict[str, JaggedTensor] = {}
        if self._rw_feats:
            out.update(self._forward_row_wise(features))
        if self._tw_feats:
            out.update(self._forward_table_wise(features))
        return NoWait({k: out[k] for k in self._feature_names})

    def _forward_table_wise(self, features: KeyedJaggedTensor) -> Dict[str, JaggedTensor]:
        W, B, pg = self._W, features.stride(), self._pg
        pos = {k: i for i, k in enumerate(features.keys())}
        order = [pos[self._feature_names[g]] for g in self._send_order]
        sent = features if order == list(range(len(features.keys()))) else features.permute(order)
        lengths, values = sent.lengths(), sent.values()
        lpk = sent.length_per_key()
        val_in, k = [], 0
        for n in self._send_per_rank:
            val_in.append(sum(lpk[k:k + n]))
            k += n
        if W > 1:
            recv_l = torch.empty(W * self._F_local * B, dtype=lengths.dtype, device=lengths.device)
            dist.all_to_all_single(recv_l, lengths, [self._F_local * B] * W, [n * B for n in self._send_per_rank], group=pg)
            val_out = recv_l.view(W, -1).sum(dim=1).cpu().tolist() if self._F_local else [0] * W
            recv_v = torch.empty(sum(val_out), dtype=values.dtype, device=values.device)
            dist.all_to_all_single(recv_v, values, val_out, val_in, group=pg)
        else:
            recv_l, recv_v, val_out = lengths, values, val_in
        D = self.embedding_dim
        if self._emb_module is not None and self._F_local:
            offsets = torch.ops.fbgemm.asynchronous_complete_cumsum(recv_l).long()
            emb = self._emb_module(recv_v, offsets)
        else:
            # requires_grad: this rank must still take part in the backward exchange of the other ranks' rows
            emb = torch.zeros((0, D), dtype=torch.float32, device=self._device, requires_grad=True)
        back = _SeqExchange.apply(emb, pg, val_out, val_in) if W > 1 else emb
        # `back` is ordered like `sent` (dest rank, its local features, samples)
        opk = sent.offset_per_key()
        out: Dict[str, JaggedTensor] = {}
        for i, g in enumerate(self._send_order):
            out[self._feature_names[g]] = JaggedTensor(values=back[opk[i]:opk[i + 1]], lengths=lengths[i * B:(i + 1) * B])
        return out

"""Module-level distribution primitives with the reference's names
(torchrec/distributed/dist_data.py): `_get_recat` (:40-118), `KJTAllToAll` (:137-524, two-phase
lengths / values exchange + recat permute), `PooledEmbeddingsAllToAll` (:602-697),
`PooledEmbeddingsReduceScatter` (:745-795).  The recat runs on
torch.ops.fbgemm.permute_2D_sparse_data (this repo's HIP kernel)."""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn

from ..sparse.jagged_tensor import KeyedJaggedTensor
from .comm_ops import alltoall_pooled, reduce_scatter_pooled
from .types import Awaitable, LazyAwaitable, NoWait


def _get_recat(local_split: int, num_splits: int, stagger: int = 1,
               device: Optional[torch.device] = None) -> torch.Tensor:
    """Permutation taking [src rank][local feature] row order to [local feature][src rank]
    (examples at dist_data.py:62-65: (2,4,1) -> [0,2,4,6,1,3,5,7], (2,4,2) -> [0,4,2,6,1,5,3,7])."""
    feature_order = [x + num_splits // stagger * y for x in range(num_splits // stagger) for y in range(stagger)]
    recat = [i + j * local_split for i in range(local_split) for j in feature_order]
    return torch.tensor(recat, dtype=torch.int32, device=device)


class _KJTValuesAwaitable(LazyAwaitable):
    def __init__(self, fn) -> None:
        super().__init__()
        self._fn = fn

    def _wait_impl(self) -> KeyedJaggedTensor:
        return self._fn()


class KJTAllToAll(nn.Module):
    """Redistributes a KJT so that rank r receives, from every rank, the features
    `splits[r]` owns.  `forward(kjt).wait()` has exchanged the lengths (and read the value counts
    back, dist_data.py:396-398); `.wait().wait()` is the KJT with keys = this rank's features and
    stride = sum of the ranks' batch sizes."""

    def __init__(self, pg: dist.ProcessGroup, splits: List[int], device: Optional[torch.device] = None,
                 stagger: int = 1, variable_batch_size: bool = False) -> None:
        super().__init__()
        if variable_batch_size:
            raise NotImplementedError("variable batch size is outside the MI355X hot path")
        assert len(splits) == dist.get_world_size(pg)
        self._pg, self._splits, self._stagger = pg, list(splits), stagger
        self._W, self._me = dist.get_world_size(pg), dist.get_rank(pg)
        self._recat = _get_recat(splits[self._me], self._W, stagger, device)

    def forward(self, kjt: KeyedJaggedTensor) -> Awaitable[Awaitable[KeyedJaggedTensor]]:
        W, me, pg = self._W, self._me, self._pg
        B = kjt.stride()
        F_local = self._splits[me]
        lengths, values, weights = kjt.lengths(), kjt.values(), kjt.weights_or_none()
        lpk = kjt.length_per_key()
        keys = kjt.keys()
        start = sum(self._splits[:me])
        local_keys = keys[start:start + F_local]
        len_in = [s * B for s in self._splits]
        recv_l = torch.empty(W * F_local * B, dtype=lengths.dtype, device=lengths.device)
        dist.all_to_all_single(recv_l, lengths, [F_local * B] * W, len_in, group=pg)
        val_in, k = [], 0
        for s in self._splits:
            val_in.append(sum(lpk[k:k + s]))
            k += s
        val_out = recv_l.view(W, -1).sum(dim=1).cpu().tolist()
        recv_v = torch.empty(sum(val_out), dtype=values.dtype, device=values.device)
        wk = dist.all_to_all_single(recv_v, values, val_out, val_in, group=pg, async_op=True)
        recv_w, wk2 = None, None
        if weights is not None:
            recv_w = torch.empty(sum(val_out), dtype=weights.dtype, device=weights.device)
            wk2 = dist.all_to_all_single(recv_w, weights, val_out, val_in, group=pg, async_op=True)

        def finish() -> KeyedJaggedTensor:
            wk.wait()
            if wk2 is not None:
                wk2.wait()
            if F_local == 0:
                return KeyedJaggedTensor(keys=[], values=recv_v, weights=recv_w, lengths=recv_l, stride=W * B)
            l2, v2, w2 = torch.ops.fbgemm.permute_2D_sparse_data(
                self._recat.to(recv_l.device), recv_l.view(W * F_local, B), recv_v, recv_w, recv_v.numel())
            return KeyedJaggedTensor(keys=local_keys, values=v2, weights=w2, lengths=l2.view(-1), stride=W * B)

        return NoWait(_KJTValuesAwaitable(finish))


class PooledEmbeddingsAllToAll(nn.Module):
    def __init__(self, pg: dist.ProcessGroup, dim_sum_per_rank: List[int], device: Optional[torch.device] = None,
                 callbacks=None) -> None:
        super().__init__()
        self._pg, self._dims = pg, list(dim_sum_per_rank)
        self._callbacks = callbacks or []
        self.register_buffer("_dim_sum_per_rank_tensor", torch.tensor(dim_sum_per_rank, dtype=torch.int32, device=device),
                             persistent=False)

    def forward(self, local_embs: torch.Tensor, batch_size_per_rank: Optional[List[int]] = None) -> Awaitable[torch.Tensor]:
        W = dist.get_world_size(self._pg)
        if batch_size_per_rank is None:
            batch_size_per_rank = [local_embs.shape[0] // W] * W
        aw = alltoall_pooled(local_embs, batch_size_per_rank, self._dims, self._dim_sum_per_rank_tensor, None, self._pg)
        if not self._callbacks:
            return aw
        cbs = self._callbacks

        class _CB(LazyAwaitable):
            def _wait_impl(self_inner):
                out = aw.wait()
                for cb in cbs:
                    out = cb(out)
                return out

        return _CB()


class PooledEmbeddingsReduceScatter(nn.Module):
    def __init__(self, pg: dist.ProcessGroup) -> None:
        super().__init__()
        self._pg = pg

    def forward(self, local_embs: torch.Tensor) -> Awaitable[torch.Tensor]:
        W = dist.get_world_size(self._pg)
        B_l = local_embs.shape[0] // W
        return reduce_scatter_pooled([local_embs[r * B_l:(r + 1) * B_l] for r in range(W)], self._pg)

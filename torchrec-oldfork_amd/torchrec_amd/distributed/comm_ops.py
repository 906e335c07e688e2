"""Collective wrappers with the reference's names and contracts
(torchrec/distributed/comm_ops.py): `alltoall_pooled` (:203-256, autograd :462-605),
`reduce_scatter_pooled` (:382-414, autograd :848-930), `Request` (:51-78),
`set_gradient_division` (:35-40).  Layout work runs on this repo's HIP kernels
(tbe_a2a_pooled_unpack/pack) instead of torch split+cat (:555-561) and
`_recat_pooled_embedding_grad_out` (:418-428); the reduce-scatter is an all-to-all followed by a
local sum in rank order (xGMI is point-to-point: no ring), which also makes it bitwise
reproducible.  ShardedEmbeddingBagCollection uses the fused one-exchange path in
embeddingbag.py; these functions are the drop-in pieces for callers written against the
reference API."""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from ..profiling import label
from . import _device_ops  # noqa: F401
from . import embeddingbag as _eb
from .types import Awaitable, NoWait


def set_gradient_division(val: bool) -> None:
    _eb.set_gradient_division(val)


class Request(Awaitable[Tensor]):
    """Work handle of an in-flight collective; wait() finishes it inside autograd."""

    def __init__(self, pg: dist.ProcessGroup) -> None:
        self.pg = pg
        self._fn = None

    def wait(self) -> Tensor:
        out = self._fn()
        self._fn = None
        return out


class _A2APooledReq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, st):
        ctx.st = st
        W = len(st["B_per_rank"])
        me = dist.get_rank(st["pg"])
        D_local = x.shape[1]
        st["D_local"] = D_local
        B_me = st["B_per_rank"][me]
        st["send_splits"] = [b * D_local for b in st["B_per_rank"]]
        st["recv_splits"] = [B_me * d for d in st["dims"]]
        st["recv"] = torch.empty(sum(st["recv_splits"]), dtype=x.dtype, device=x.device)
        with label("## alltoall_fwd_single ##"):  # comm_ops.py:489
            st["work"] = dist.all_to_all_single(st["recv"], x.contiguous().view(-1), st["recv_splits"], st["send_splits"],
                                                group=st["pg"], async_op=True)
        return st["recv"]

    @staticmethod
    def backward(ctx, _):
        st = ctx.st
        st["bwork"].wait()
        g = st["grecv"].view(-1, st["D_local"])
        st["grecv"] = None
        return g, None


class _A2APooledWait(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recv, st):
        ctx.st = st
        st["work"].wait()
        me = dist.get_rank(st["pg"])
        B_me = st["B_per_rank"][me]
        return torch.ops.tbe_hip.a2a_pooled_unpack(recv, st["dims_t"], B_me, sum(st["dims"]), st["vec"], 1.0)

    @staticmethod
    def backward(ctx, grad_out):
        st = ctx.st
        W = len(st["dims"])
        scale = 1.0 / W if _eb.GRADIENT_DIVISION else 1.0  # comm_ops.py:527-528
        send = torch.ops.tbe_hip.a2a_pooled_pack(grad_out, st["dims_t"], st["vec"], scale)
        st["grecv"] = torch.empty(sum(st["send_splits"]), dtype=grad_out.dtype, device=grad_out.device)
        with label("## alltoall_bwd_single ##"):  # comm_ops.py:591
            st["bwork"] = dist.all_to_all_single(st["grecv"], send, st["send_splits"], st["recv_splits"], group=st["pg"],
                                                 async_op=True)
        st["keep"] = send
        return st["recv"].new_zeros(1).expand(st["recv"].shape), None


def alltoall_pooled(a2a_pooled_embs_tensor: Tensor, batch_size_per_rank: List[int], dim_sum_per_rank: List[int],
                    dim_sum_per_rank_tensor: Optional[Tensor] = None,
                    cumsum_dim_sum_per_rank_tensor: Optional[Tensor] = None,
                    group: Optional[dist.ProcessGroup] = None) -> Awaitable[Tensor]:
    """[B_global, D_local_sum] -> Awaitable of [B_local, D_global_sum]."""
    if group is None:
        group = dist.distributed_c10d._get_default_group()
    if dist.get_world_size(group) <= 1:
        return NoWait(a2a_pooled_embs_tensor)
    dev = a2a_pooled_embs_tensor.device
    dims_t = dim_sum_per_rank_tensor if dim_sum_per_rank_tensor is not None else torch.tensor(
        dim_sum_per_rank, dtype=torch.int32, device=dev)
    st = {"pg": group, "B_per_rank": list(batch_size_per_rank), "dims": list(dim_sum_per_rank),
          "dims_t": dims_t.to(torch.int32), "vec": all(d % 4 == 0 for d in dim_sum_per_rank)}
    recv = _A2APooledReq.apply(a2a_pooled_embs_tensor, st)
    req = Request(group)
    req._fn = lambda: _A2APooledWait.apply(recv, st)
    return req


class _RSReq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st, *inputs):
        ctx.st = st
        pg = st["pg"]
        W, me = dist.get_world_size(pg), dist.get_rank(pg)
        st["sizes"] = [t.shape for t in inputs]
        send = torch.cat([t.contiguous().view(-1) for t in inputs])
        n_me = inputs[me].numel()
        st["recv"] = torch.empty(W * n_me, dtype=send.dtype, device=send.device)
        with label("## reduce_scatter ##"):  # comm_ops.py:865
            st["work"] = dist.all_to_all_single(st["recv"], send, [n_me] * W, [t.numel() for t in inputs], group=pg,
                                                async_op=True)
        return st["recv"]

    @staticmethod
    def backward(ctx, _):
        st = ctx.st
        st["bwork"].wait()
        g = st["grecv"]
        outs, o = [], 0
        for shp in st["sizes"]:
            n = 1
            for d in shp:
                n *= d
            outs.append(g[o:o + n].view(shp))
            o += n
        return (None, *outs)


class _RSWait(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recv, st):
        ctx.st = st
        st["work"].wait()
        pg = st["pg"]
        W, me = dist.get_world_size(pg), dist.get_rank(pg)
        parts = recv.view(W, *st["sizes"][me])
        out = parts[0].clone()
        for r in range(1, W):  # fixed rank order: reproducible
            out += parts[r]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        st = ctx.st
        pg = st["pg"]
        W = dist.get_world_size(pg)
        g = grad_out.contiguous().view(-1)
        if _eb.GRADIENT_DIVISION:  # comm_ops.py:883-885
            g = g / W
        # all-gather of the gradient (comm_ops.py:922-927) as an all-to-all with a replicated source
        send = g.repeat(W)
        recv_splits = []
        for shp in st["sizes"]:
            n = 1
            for d in shp:
                n *= d
            recv_splits.append(n)
        st["grecv"] = torch.empty(sum(recv_splits), dtype=g.dtype, device=g.device)
        with label("## reduce_scatter_bw (all_gather) ##"):  # comm_ops.py:921
            st["bwork"] = dist.all_to_all_single(st["grecv"], send, recv_splits, [g.numel()] * W, group=pg, async_op=True)
        st["keep"] = send
        return st["recv"].new_zeros(1).expand(st["recv"].shape), None


def reduce_scatter_pooled(inputs: List[Tensor], group: Optional[dist.ProcessGroup] = None) -> Awaitable[Tensor]:
    """inputs[r] = this rank's partial pool for rank r's samples; result = sum over ranks of their
    inputs[me]."""
    if group is None:
        group = dist.distributed_c10d._get_default_group()
    if dist.get_world_size(group) <= 1:
        return NoWait(inputs[dist.get_rank(group)])
    st = {"pg": group}
    recv = _RSReq.apply(st, *inputs)
    req = Request(group)
    req._fn = lambda: _RSWait.apply(recv, st)
    return req

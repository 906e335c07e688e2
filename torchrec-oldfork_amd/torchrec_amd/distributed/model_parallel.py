"""DistributedModelParallel for the MI355X path (torchrec/distributed/model_parallel.py:162-527):
replaces every EmbeddingBagCollection in `module` by a ShardedEmbeddingBagCollection according to
the plan (planner on rank 0 semantics: the plan is a pure function of the configs, so every
rank computes the same one — no broadcast_object_list needed), and wraps the remaining dense
parameters in DistributedDataParallel (model_parallel.py:84-111) when world_size > 1."""
import os
from typing import Dict, Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.parallel import DistributedDataParallel

from ..modules.embedding_modules import EmbeddingBagCollection
from ..optim.keyed import CombinedOptimizer
from .embeddingbag import EmbeddingBagCollectionSharder, ShardedEmbeddingBagCollection
from .planner import EmbeddingShardingPlanner, Topology
from .types import ShardingEnv, ShardingPlan


class DistributedModelParallel(nn.Module):
    def __init__(self, module: nn.Module, env: Optional[ShardingEnv] = None, device: Optional[torch.device] = None,
                 plan: Optional[ShardingPlan] = None, sharders: Optional[List[EmbeddingBagCollectionSharder]] = None,
                 init_data_parallel: bool = True, planner: Optional[EmbeddingShardingPlanner] = None) -> None:
        super().__init__()
        if env is None:
            pg = dist.group.WORLD if dist.is_initialized() else None
            env = ShardingEnv.from_process_group(pg) if pg is not None else ShardingEnv.from_local(1, 0)
        self._env = env
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        sharder = (sharders or [EmbeddingBagCollectionSharder()])[0]
        planner = planner or EmbeddingShardingPlanner(Topology(env.world_size, self.device.type))
        self._plan = plan or ShardingPlan()
        self._sharded: List[ShardedEmbeddingBagCollection] = []
        self._sharded_paths: List[str] = []
        self._shard_modules(module, "", sharder, planner)
        self._dmp_wrapped_module = module
        self._ddp_wrapped = False
        if init_data_parallel:
            self.init_data_parallel()

    def _shard_modules(self, module: nn.Module, path: str, sharder, planner) -> None:
        for name, child in list(module.named_children()):
            child_path = f"{path}.{name}" if path else name
            if isinstance(child, EmbeddingBagCollection):
                params = self._plan.get_plan_for_module(child_path)
                if params is None:
                    params = planner.plan_tables(child.embedding_bag_configs)
                    self._plan.plan[child_path] = params
                sharded = sharder.shard(child, params, self._env, self.device)
                setattr(module, name, sharded)
                self._sharded.append(sharded)
                self._sharded_paths.append(child_path)
            else:
                self._shard_modules(child, child_path, sharder, planner)

    def init_data_parallel(self) -> None:
        if self._ddp_wrapped:
            return
        m = self._dmp_wrapped_module
        dense = [p for _, p in self._dense_named_parameters(m)]
        for p in dense:
            if p.device != self.device:
                raise RuntimeError("dense parameters must live on the DMP device")
        # TORCHREC_AMD_FORCE_DDP=1: rehearsal switch — wrap in DistributedDataParallel even on a one-rank group
        force = os.environ.get("TORCHREC_AMD_FORCE_DDP", "0") == "1" and self._env.process_group is not None
        # parameters whose gradients a model reduces itself through a flat buffer (DLRMTrain.capture_hip_graphs(
        # flat_grads=True)) stay out of DDP
        flat_ids = {id(p) for p in (m.flat_grad_parameters() if hasattr(m, "flat_grad_parameters") else [])}
        if flat_ids and self._env.world_size > 1:
            # what DDP's constructor does for its own parameters: every rank starts from rank 0's values
            with torch.no_grad():
                for p in dense:
                    if id(p) in flat_ids:
                        dist.broadcast(p.data, src=dist.get_global_rank(self._env.process_group, 0)
                                       if self._env.process_group is not None else 0, group=self._env.process_group)
        dense = [p for p in dense if id(p) not in flat_ids]
        if (self._env.world_size > 1 or force) and dense:
            # sharded tables (buffers of the TBE modules) differ per rank and must neither be
            # broadcast at DDP construction nor reduced (model_parallel.py:84-100 does the same)
            ignore = []
            for path, sub in m.named_modules():
                if isinstance(sub, ShardedEmbeddingBagCollection):
                    for n, _ in nn.Module.named_buffers(sub):
                        ignore.append(f"{path}.{n}" if path else n)
                    for n, _ in nn.Module.named_parameters(sub):
                        if not n.startswith("_dp_module."):  # replicated tables ARE reduced by DDP
                            ignore.append(f"{path}.{n}" if path else n)
            ignore += [n for n, p in m.named_parameters() if id(p) in flat_ids]
            DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(m, ignore)
            self._dmp_wrapped_module = DistributedDataParallel(
                m, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                process_group=self._env.process_group, gradient_as_bucket_view=True, broadcast_buffers=False,
                static_graph=True)
        self._ddp_wrapped = True

    def is_data_parallel_wrapped(self) -> bool:
        return isinstance(self._dmp_wrapped_module, DistributedDataParallel)

    @staticmethod
    def _dense_named_parameters(m: nn.Module) -> Iterator[Tuple[str, nn.Parameter]]:
        # torch's named_parameters: sharded tables expose none (they are fused)
        yield from m.named_parameters()

    @property
    def module(self) -> nn.Module:
        m = self._dmp_wrapped_module
        return m.module if isinstance(m, DistributedDataParallel) else m

    @property
    def plan(self) -> ShardingPlan:
        return self._plan

    def sharded_modules(self) -> List[ShardedEmbeddingBagCollection]:
        return self._sharded

    @property
    def fused_optimizer(self) -> CombinedOptimizer:
        """Keys as the reference's (model_parallel.py:455-470): `<module path>.embedding_bags.<table>.weight`."""
        return CombinedOptimizer([(path, s.fused_optimizer) for path, s in zip(self._sharded_paths, self._sharded)
                                  if s.fused_optimizer is not None])

    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False):
        return self.module.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)

    def load_state_dict(self, state_dict, strict: bool = True):
        return self.module.load_state_dict(state_dict, strict=strict)

    def forward(self, *args, **kwargs):
        return self._dmp_wrapped_module(*args, **kwargs)

    def named_parameters(self, prefix: str = "", recurse: bool = True):
        yield from self.module.named_parameters(prefix=prefix, recurse=recurse)

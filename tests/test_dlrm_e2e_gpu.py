"""GPU end-to-end parity: DLRM trained for a few steps through this repo's stack
(DistributedModelParallel world 1 -> ShardedEmbeddingBagCollection -> HIP TBE fused exact SGD, fused
MFMA interaction, epilogue-fused MLP, TrainPipelineSparseDist) against a plain-PyTorch fp32 model
built the reference's way: nn.EmbeddingBag per table (torchrec/modules/embedding_modules.py:149-156),
cat + bmm + triu interaction (torchrec/models/dlrm.py:206-219), nn.Linear + relu MLPs, torch.optim.SGD
for every parameter — the ground truth the reference's own sharded-vs-unsharded test uses
(torchrec/distributed/test_utils/test_model_parallel_base.py:257-283)."""
import numpy as np
import pytest
import torch
from torch import nn

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu


class TorchRefDLRM(nn.Module):
    def __init__(self, rows, D, dense_in, dense_sizes, over_sizes):
        super().__init__()
        self.bags = nn.ModuleList([nn.EmbeddingBag(r, D, mode="sum", include_last_offset=True) for r in rows])
        sizes = [dense_in] + dense_sizes
        self.dense = nn.ModuleList([nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(dense_sizes))])
        F = len(rows)
        sizes = [D + F * (F + 1) // 2] + over_sizes
        self.over = nn.ModuleList([nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(over_sizes))])
        self.F, self.D = F, D
        self.register_buffer("tri", torch.triu_indices(F + 1, F + 1, offset=1), persistent=False)

    def forward(self, dense, values, B):
        x = dense
        for lin in self.dense:
            x = torch.relu(lin(x))
        offs = torch.arange(B + 1, device=dense.device)
        pooled = [self.bags[f](values[f * B:(f + 1) * B], offs) for f in range(self.F)]
        combined = torch.cat([x.unsqueeze(1), torch.stack(pooled, dim=1)], dim=1)
        inter = torch.bmm(combined, combined.transpose(1, 2))
        y = torch.cat([x, inter[:, self.tri[0], self.tri[1]]], dim=1)
        for i, lin in enumerate(self.over):
            y = lin(y)
            if i < len(self.over) - 1:
                y = torch.relu(y)
        return y.squeeze(-1)


@pytest.mark.parametrize("host_batches", [False, True])
def test_dlrm_training_matches_plain_torch_reference(host_batches):
    """host_batches: the batches live in PINNED HOST memory, so the pipeline's memcpy stream really copies batch
    i + 2 while batch i trains and batch i + 1's ids are in flight on the data_dist stream (device-resident
    batches make `.to()` a no-op); a missing record_stream shows as corrupted inputs here."""
    from torchrec_amd.datasets.random import RandomRecDataset
    from torchrec_amd.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec_amd.distributed.model_parallel import DistributedModelParallel
    from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist
    from torchrec_amd.distributed.types import ShardingEnv
    from torchrec_amd.models.dlrm import DLRMTrain
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
    from torchrec_amd.optim.keyed import CombinedOptimizer, KeyedOptimizerWrapper

    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    rows, D, B, lr = [1000, 3, 57, 20000, 11], 128, 512, 0.05
    keys = [f"cat_{i}" for i in range(len(rows))]
    dense_sizes, over_sizes = [64, D], [96, 32, 1]
    tables = [EmbeddingBagConfig(name=f"t_{k}", embedding_dim=D, num_embeddings=rows[i], feature_names=[k])
              for i, k in enumerate(keys)]
    ebc = EmbeddingBagCollection(tables, device=torch.device("meta"))
    train_model = DLRMTrain(ebc, 13, dense_sizes, over_sizes, dense_device=dev)
    model = DistributedModelParallel(train_model, env=ShardingEnv.from_local(1, 0), device=dev,
                                     sharders=[EmbeddingBagCollectionSharder({"learning_rate": lr})])
    opt = CombinedOptimizer([model.fused_optimizer,
                             KeyedOptimizerWrapper(dict(model.named_parameters()), lambda p: torch.optim.SGD(p, lr=lr))])
    # reference model with identical initial weights
    ref = TorchRefDLRM(rows, D, 13, dense_sizes, over_sizes).to(dev)
    shards = model.sharded_modules()[0].local_shards()
    with torch.no_grad():
        for i, k in enumerate(keys):
            ref.bags[i].weight.copy_(shards[f"t_{k}"][0])
        m = model.module.model
        for i, lin in enumerate(ref.dense):
            lin.weight.copy_(m.dense_arch.model._mlp[i]._linear.weight)
            lin.bias.copy_(m.dense_arch.model._mlp[i]._linear.bias)
        over_mlp = m.over_arch.model[0]._mlp
        for i, lin in enumerate(ref.over[:-1]):
            lin.weight.copy_(over_mlp[i]._linear.weight)
            lin.bias.copy_(over_mlp[i]._linear.bias)
        ref.over[-1].weight.copy_(m.over_arch.model[1].weight)
        ref.over[-1].bias.copy_(m.over_arch.model[1].bias)
    ref_opt = torch.optim.SGD(ref.parameters(), lr=lr)
    data = RandomRecDataset(keys, B, rows, manual_seed=5, num_generated_batches=6, num_batches=6, device=dev)
    batches = list(iter(data))
    feed = [b.to(torch.device("cpu")).pin_memory() for b in batches] if host_batches else batches
    torch.cuda.synchronize()
    pipe = TrainPipelineSparseDist(model, opt, dev)
    model.train()
    it = iter(feed)
    bce = nn.BCEWithLogitsLoss()
    for step in range(6):
        loss = pipe.progress(it)[0]  # DLRMTrain output = (loss.detach(), logits, labels)
        b = batches[step]
        ref_opt.zero_grad()
        ref_loss = bce(ref(b.dense_features, b.sparse_features.values(), B), b.labels.float())
        ref_loss.backward()
        ref_opt.step()
        torch.testing.assert_close(loss, ref_loss, rtol=2e-4, atol=2e-5)
    torch.cuda.synchronize()
    for i, k in enumerate(keys):
        torch.testing.assert_close(shards[f"t_{k}"][0], ref.bags[i].weight, rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(m.over_arch.model[1].weight, ref.over[-1].weight, rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(m.dense_arch.model._mlp[0]._linear.weight, ref.dense[0].weight, rtol=1e-3, atol=2e-5)

"""BASELINE config 5 at model level: BERT4Rec's HistoryArch (examples/bert4rec/models/bert4rec.py:323-408)
on this package's unpooled lookup + jagged_2d_to_dense with the fused ADAM of
examples/bert4rec/bert4rec_main.py:488-491, a transformer block and an output layer on top, trained for a
few steps against a plain-PyTorch twin: nn.Embedding(sparse=True) + torch.optim.SparseAdam (the same lazy
"only rows of this batch, duplicates summed first" semantics, an independent implementation) and a python
padding loop.  This is the only independent pin of the fused ADAM arithmetic (fbgemm's is absent).

eps: this package adds eps to the bias-corrected sqrt(v_hat) (the public fbgemm / torch.optim.Adam form);
torch.optim.SparseAdam adds it to sqrt(v) before the correction, i.e. an effective eps / sqrt(1 - beta2^t).
The two differ only for |g| within a few orders of eps (measured: up to 0.44 lr at |g| ~ 1e-7 with eps =
1e-8), so the twin runs with eps = 1e-30, where both forms coincide and everything else is pinned; the eps
placement itself stays "parity unpinned" (DESIGN.md §5)."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu


class _ItemHistoryEncoder(nn.Module):
    """Test-local model of BERT4Rec's embedding side (what examples/bert4rec/models/bert4rec.py:323-408 computes for
    its single "item" feature): unpooled lookup of the item ids through this package's EmbeddingCollection (fused
    optimizer inside), left-aligned zero padding / truncation of every history to `length` rows with the
    jagged_2d_to_dense kernel, a learned position term and a LayerNorm over (length, dim)."""

    def __init__(self, num_items, length, dim, dev, fused_params):
        super().__init__()
        from torchrec_amd.modules.embedding_configs import EmbeddingConfig
        from torchrec_amd.modules.embedding_modules import EmbeddingCollection

        self.length, self.dim = length, dim
        self.items = EmbeddingCollection(
            tables=[EmbeddingConfig(name="item_embedding", embedding_dim=dim, num_embeddings=num_items, feature_names=["item"],
                                    weight_init_min=-1.0, weight_init_max=1.0)], device=dev, fused_params=fused_params)
        self.position = nn.Parameter(torch.randn(length, dim, device=dev))
        self.norm = nn.LayerNorm([length, dim], device=dev)

    def forward(self, histories):
        rows = self.items(histories)["item"]  # JaggedTensor: one row per id
        dense = torch.ops.fbgemm.jagged_2d_to_dense(rows.values(), rows.offsets(), self.length)
        return self.norm(dense.view(-1, self.length, self.dim) + self.position)


class _Top(nn.Module):
    def __init__(self, D, L, vocab, dev):
        super().__init__()
        self.block = nn.TransformerEncoderLayer(d_model=D, nhead=4, dim_feedforward=2 * D, dropout=0.0, batch_first=True,
                                                device=dev)
        self.out = nn.Linear(D, vocab, device=dev)

    def forward(self, x):
        return self.out(self.block(x))


def test_item_history_encoder_with_fused_adam_matches_sparse_adam_twin():
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    vocab, L, D, B, lr = 400, 8, 64, 24, 1e-2
    hist = _ItemHistoryEncoder(vocab, L, D, dev, {"optimizer": EmbOptimType.ADAM, "learning_rate": lr, "eps": 1e-30,
                                                  "beta1": 0.9, "beta2": 0.999, "weight_decay": 0.0})
    top = _Top(D, L, vocab, dev)
    # twin
    emb2 = nn.Embedding(vocab, D, sparse=True, device=dev)
    with torch.no_grad():
        emb2.weight.copy_(hist.items.table_weights()["item_embedding"])
    pos2 = nn.Parameter(hist.position.detach().clone())
    ln2, top2 = copy.deepcopy(hist.norm), copy.deepcopy(top)
    opt_dense = torch.optim.Adam([hist.position] + list(hist.norm.parameters()) + list(top.parameters()), lr=1e-3)
    opt_dense2 = torch.optim.Adam([pos2] + list(ln2.parameters()) + list(top2.parameters()), lr=1e-3)
    opt_emb2 = torch.optim.SparseAdam(list(emb2.parameters()), lr=lr, betas=(0.9, 0.999), eps=1e-30)
    rng = np.random.default_rng(1)
    ce = nn.CrossEntropyLoss()
    for step in range(5):
        lengths = rng.integers(0, 12, size=B).astype(np.int32)  # some histories longer than L (truncated), some empty
        ids = rng.integers(0, vocab, size=int(lengths.sum())).astype(np.int64)
        ids[::5] = ids[0] if ids.size else 0  # duplicates inside a batch
        target = torch.from_numpy(rng.integers(0, vocab, size=(B, L))).to(dev)
        kjt = KeyedJaggedTensor.from_lengths_sync(["item"], torch.from_numpy(ids).to(dev), torch.from_numpy(lengths).to(dev))
        opt_dense.zero_grad()
        loss = ce(top(hist(kjt)).reshape(B * L, vocab), target.reshape(-1))
        loss.backward()
        opt_dense.step()
        # twin: pad with a python loop
        opt_dense2.zero_grad()
        opt_emb2.zero_grad()
        rows = emb2(torch.from_numpy(ids).to(dev))
        padded = torch.zeros(B, L, D, device=dev)
        chunks, pos = [], 0
        for b in range(B):
            n = int(lengths[b])
            keep = min(n, L)
            chunks.append(torch.cat([rows[pos:pos + keep], torch.zeros(L - keep, D, device=dev)]))
            pos += n
        padded = torch.stack(chunks)
        loss2 = ce(top2(ln2(padded + pos2.unsqueeze(0))).reshape(B * L, vocab), target.reshape(-1))
        loss2.backward()
        opt_dense2.step()
        opt_emb2.step()
        torch.testing.assert_close(loss, loss2, rtol=1e-4, atol=1e-5)
    torch.cuda.synchronize()
    torch.testing.assert_close(hist.items.table_weights()["item_embedding"], emb2.weight.detach(), rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(hist.position.detach(), pos2.detach(), rtol=1e-3, atol=2e-5)


def test_fused_adagrad_matches_torch_sparse_adagrad():
    """EXACT_ADAGRAD (element-wise state) against torch.optim.Adagrad on a sparse nn.Embedding gradient — an
    independent implementation of `state += g^2; w -= lr * g / (sqrt(state) + eps)` with duplicates of a row
    summed first.  Pooled lookups (SUM, ragged bags) so that the gradient of a row is a sum over bags."""
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import (
        ComputeDevice, EmbeddingLocation, PoolingMode, SplitTableBatchedEmbeddingBagsCodegen)

    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    rows, D, B, lr, eps = 300, 32, 40, 0.05, 1e-6
    mod = SplitTableBatchedEmbeddingBagsCodegen([(rows, D, EmbeddingLocation.DEVICE, ComputeDevice.CUDA)],
                                                pooling_mode=PoolingMode.SUM, device=dev, optimizer=EmbOptimType.EXACT_ADAGRAD,
                                                learning_rate=lr, eps=eps)
    bag = nn.EmbeddingBag(rows, D, mode="sum", sparse=True, include_last_offset=True, device=dev)
    w0 = torch.randn(rows, D, device=dev)
    mod.split_embedding_weights()[0].copy_(w0)
    with torch.no_grad():
        bag.weight.copy_(w0)
    opt = torch.optim.Adagrad(bag.parameters(), lr=lr, eps=eps)
    rng = np.random.default_rng(4)
    for _ in range(5):
        lengths = rng.integers(0, 5, size=B)
        ids = torch.from_numpy(rng.integers(0, rows, size=int(lengths.sum())).astype(np.int64)).to(dev)
        offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)).to(dev)
        g = torch.randn(B, D, device=dev)
        out = mod(ids, offsets)
        out.backward(g)
        opt.zero_grad()
        ref = bag(ids, offsets)
        torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-5, atol=1e-5)
        ref.backward(g)
        opt.step()
    torch.cuda.synchronize()
    torch.testing.assert_close(mod.split_embedding_weights()[0], bag.weight.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mod.split_optimizer_states()[0][0], opt.state[bag.weight]["sum"], rtol=1e-4, atol=1e-6)

"""GPU: the unpooled ("sequence") path of config 5 (BERT4Rec): EmbeddingCollection lookup
(PoolingMode.NONE TBE) -> fbgemm.jagged_2d_to_dense, forward and backward, against the oracle and the
reference's formulation (examples/bert4rec/models/bert4rec.py:380-408)."""
import numpy as np
import pytest
import torch

import _paths  # noqa: F401
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_embedding_collection_and_jagged_to_dense_forward_backward():
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from torchrec_amd.modules.embedding_configs import EmbeddingConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingCollection
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    rng = np.random.default_rng(3)
    rows, D, B, max_L = [50, 31], 64, 9, 6
    keys = ["item", "cate"]
    cfgs = [EmbeddingConfig(name=f"t{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[keys[i]]) for i in range(2)]
    lr = 0.05
    ec = EmbeddingCollection(cfgs, device=torch.device("cuda", 0),
                             fused_params={"optimizer": EmbOptimType.EXACT_SGD, "learning_rate": lr})
    tabs = oracle.Tables(rows, [D, D])
    for t, (name, w) in enumerate(ec.table_weights().items()):
        init = rng.standard_normal((rows[t], D)).astype(np.float32)
        tabs.weights[t][...] = init
        w.copy_(torch.from_numpy(init))
    lengths = rng.integers(0, 9, size=2 * B).astype(np.int32)  # some bags longer than max_L (truncated)
    values = np.concatenate([rng.integers(0, rows[f], size=int(lengths[f * B:(f + 1) * B].sum())) for f in range(2)]).astype(np.int64)
    kjt = KeyedJaggedTensor.from_lengths_sync(keys, torch.from_numpy(values).cuda(), torch.from_numpy(lengths).cuda())
    jt = ec(kjt)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    ref_emb, _ = oracle.tbe_forward(tabs, values, offsets, None, oracle.POOL_NONE)
    opk = [0, int(lengths[:B].sum()), int(lengths.sum())]
    padded, ref_padded = [], []
    for f, k in enumerate(keys):
        np.testing.assert_array_equal(jt[k].values().detach().cpu().numpy(), ref_emb[opk[f]:opk[f + 1]])
        d = torch.ops.fbgemm.jagged_2d_to_dense(values=jt[k].values(), offsets=jt[k].offsets(), max_sequence_length=max_L)
        offs_f = np.concatenate([[0], np.cumsum(lengths[f * B:(f + 1) * B])]).astype(np.int64)
        ref_d = oracle.jagged_2d_to_dense(ref_emb[opk[f]:opk[f + 1]], offs_f, max_L)
        np.testing.assert_array_equal(d.detach().cpu().numpy(), ref_d)
        padded.append(d)
        ref_padded.append(ref_d)
    x = torch.cat(padded, dim=1)  # [B, 2*max_L, D] as bert4rec.py:394-403
    g = rng.standard_normal(tuple(x.shape)).astype(np.float32)
    x.backward(torch.from_numpy(g).cuda())
    torch.cuda.synchronize()
    # reference gradient wrt the [N, D] unpooled embeddings: scatter back, truncated rows get 0
    grad_emb = np.zeros_like(ref_emb)
    for f in range(2):
        gf = g[:, f * max_L:(f + 1) * max_L]
        pos = opk[f]
        for b in range(B):
            L = int(lengths[f * B + b])
            n = min(L, max_L)
            grad_emb[pos:pos + n] = gf[b, :n]
            pos += L
    oracle.tbe_backward(tabs, values, offsets, grad_emb, oracle.OPT_EXACT_SGD, lr, None, oracle.POOL_NONE)
    for t, (name, w) in enumerate(ec.table_weights().items()):
        np.testing.assert_allclose(w.cpu().numpy(), tabs.weights[t], rtol=2e-5, atol=2e-5)

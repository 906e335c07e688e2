"""The embedding-side piece of BERT4Rec (BASELINE config 5): HistoryArch,
examples/bert4rec/models/bert4rec.py:323-408 — item ids -> EmbeddingCollection (unpooled TBE lookup)
-> fbgemm.jagged_2d_to_dense padding to [B, history_len, D] -> + positional parameter -> LayerNorm ->
dropout.  The transformer blocks and the output layer behind it are plain dense torch modules under
DistributedDataParallel in the reference (bert4rec_main.py:488-519) and are not part of this package's
hot path; callers stack them on top (tests/test_bert4rec_gpu.py does)."""
from typing import Optional

import torch
from torch import nn

from ..modules.embedding_configs import EmbeddingConfig
from ..modules.embedding_modules import EmbeddingCollection
from ..sparse.jagged_tensor import KeyedJaggedTensor


class HistoryArch(nn.Module):
    def __init__(self, vocab_size: int, history_len: int, emb_dim: int, dropout: float = 0.1,
                 device: Optional[torch.device] = None, fused_params: Optional[dict] = None,
                 embedding_collection: Optional[nn.Module] = None) -> None:
        super().__init__()
        self.emb_dim, self.history_len = emb_dim, history_len
        self.positional = nn.Parameter(torch.randn(history_len, emb_dim, device=device))
        self.layernorm = nn.LayerNorm([history_len, emb_dim], device=device)
        self.dropout = nn.Dropout(p=dropout)
        # a sharded collection (distributed/embedding.py ShardedEmbeddingCollection) can be passed in instead
        self.ec = embedding_collection if embedding_collection is not None else EmbeddingCollection(
            tables=[EmbeddingConfig(name="item_embedding", embedding_dim=emb_dim, num_embeddings=vocab_size,
                                    feature_names=["item"], weight_init_max=1.0, weight_init_min=-1.0)],
            device=device, fused_params=fused_params)

    def forward(self, id_list_features: KeyedJaggedTensor) -> torch.Tensor:
        jt_dict = self.ec(id_list_features)
        if hasattr(jt_dict, "wait"):
            jt_dict = jt_dict.wait()
        padded = [torch.ops.fbgemm.jagged_2d_to_dense(values=jt_dict[e].values(), offsets=jt_dict[e].offsets(),
                                                      max_sequence_length=self.history_len
                                                      ).view(-1, self.history_len, self.emb_dim)
                  for e in id_list_features.keys()]
        item_output = torch.cat(padded, dim=1)
        x = item_output + self.positional.unsqueeze(0)
        return self.dropout(self.layernorm(x))

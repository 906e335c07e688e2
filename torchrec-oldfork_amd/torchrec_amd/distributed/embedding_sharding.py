"""Named mirror of torchrec/distributed/embedding_sharding.py:121-184 on this build's bucketize kernel
(`torch.ops.fbgemm.block_bucketize_sparse_features` -> tbe_block_bucketize, csrc/sparse_ops.hip)."""
from typing import Optional, Tuple

import torch

from ..sparse.jagged_tensor import KeyedJaggedTensor


def bucketize_kjt_before_all2all(kjt: KeyedJaggedTensor, num_buckets: int, block_sizes: torch.Tensor,
                                 output_permute: bool = False, bucketize_pos: bool = False
                                 ) -> Tuple[KeyedJaggedTensor, Optional[torch.Tensor]]:
    """Bucketizes the ids of every feature by row block (bucket = id // block_sizes[f], id %= block_sizes[f]);
    the result holds `num_buckets` copies of the keys, bucket-major, ready for the row-wise id all-to-all.
    Returns the bucketized KeyedJaggedTensor and, with `output_permute`, the position of every original id
    in the bucketized order (`unbucketize_permute`, used by the sequence path: dist_data.py:820-827)."""
    num_features = len(kjt.keys())
    assert block_sizes.numel() == num_features, (
        f"Expecting block sizes for {num_features} features, but {block_sizes.numel()} received.")
    block_sizes_new_type = block_sizes.to(kjt.values().dtype)  # the kernel wants ids and block sizes of one type
    lengths, indices, weights, pos, unbucketize_permute = torch.ops.fbgemm.block_bucketize_sparse_features(
        kjt.lengths().view(-1), kjt.values(), bucketize_pos=bucketize_pos, sequence=output_permute,
        block_sizes=block_sizes_new_type, my_size=num_buckets, weights=kjt.weights_or_none())
    return (KeyedJaggedTensor(keys=kjt.keys() * num_buckets, values=indices,
                              weights=pos if bucketize_pos else weights, lengths=lengths.view(-1), stride=kjt.stride()),
            unbucketize_permute)

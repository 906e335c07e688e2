"""Synthetic Criteo-shaped batches (torchrec/datasets/random.py:68-110,
examples/dlrm/data/dlrm_dataloader.py:26-45, torchrec/datasets/utils.py:34-61 `Batch`).
Unlike the reference (which replays ONE generated batch, random.py:47-49) a pool of distinct
batches is generated so cache reuse is realistic; ids are uniform or Zipf."""
from dataclasses import dataclass
from typing import Iterator, List, Optional

import torch

from ..sparse.jagged_tensor import KeyedJaggedTensor

CRITEO_1TB_ROWS = [45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209,
                   11267, 128, 4, 974, 14, 48937457, 11316796, 40094537, 452104, 12606, 104, 35]
CRITEO_KAGGLE_ROWS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                      5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
INT_FEATURE_COUNT = 13
CAT_FEATURE_COUNT = 26
DEFAULT_CAT_NAMES = [f"cat_{i}" for i in range(CAT_FEATURE_COUNT)]


@dataclass
class Batch:
    dense_features: torch.Tensor
    sparse_features: KeyedJaggedTensor
    labels: torch.Tensor

    def to(self, device: torch.device, non_blocking: bool = False) -> "Batch":
        return Batch(self.dense_features.to(device, non_blocking=non_blocking),
                     self.sparse_features.to(device, non_blocking=non_blocking),
                     self.labels.to(device, non_blocking=non_blocking))

    def record_stream(self, stream) -> None:
        self.dense_features.record_stream(stream)
        self.sparse_features.record_stream(stream)
        self.labels.record_stream(stream)

    def pin_memory(self) -> "Batch":
        return Batch(self.dense_features.pin_memory(), self.sparse_features.pin_memory(), self.labels.pin_memory())


class RandomRecDataset:
    """Iterable of `Batch`; generation happens on `device` with a seeded generator.

    `manual_seeds` (instead of `manual_seed`): the batch is the concatenation, along the batch dimension, of
    len(manual_seeds) sub-batches of batch_size / len(manual_seeds) samples, sub-batch j drawn exactly as a dataset with
    manual_seed = manual_seeds[j] and that smaller batch size draws it.  One rank then sees the GLOBAL batches of a
    multi-rank run (rank r of the reference's examples seeds with `seed + r`): how a world-size-1 run is compared
    with a world-size-N run on the same samples (bench.py --data-ranks, tests/test_bench_rehearsal_gpu.py)."""

    def __init__(self, keys: List[str], batch_size: int, hash_sizes: List[int], ids_per_feature: int = 1,
                 num_dense: int = INT_FEATURE_COUNT, manual_seed: Optional[int] = None,
                 num_generated_batches: int = 32, num_batches: Optional[int] = None,
                 device: Optional[torch.device] = None, zipf_alpha: Optional[float] = None,
                 manual_seeds: Optional[List[int]] = None) -> None:
        self.keys, self.batch_size, self.hash_sizes = keys, batch_size, hash_sizes
        self.ids_per_feature, self.num_dense, self.num_batches = ids_per_feature, num_dense, num_batches
        self.device = device or torch.device("cpu")
        self.zipf_alpha = zipf_alpha
        if manual_seeds is not None and len(manual_seeds) > 1:
            if manual_seed is not None or batch_size % len(manual_seeds):
                raise ValueError("manual_seeds: give either one seed or a list whose length divides the batch size")
            subs = [RandomRecDataset(keys, batch_size // len(manual_seeds), hash_sizes, ids_per_feature, num_dense, sd,
                                     num_generated_batches, None, self.device, zipf_alpha) for sd in manual_seeds]
            self._pool = [self._concat([s._pool[i] for s in subs]) for i in range(num_generated_batches)]
            return
        if manual_seeds:
            manual_seed = manual_seeds[0]
        self.gen = torch.Generator(device=self.device)
        if manual_seed is not None:
            self.gen.manual_seed(manual_seed)
        self._pool = [self._generate() for _ in range(num_generated_batches)]

    def _concat(self, parts: List[Batch]) -> Batch:
        F, L = len(self.keys), self.ids_per_feature
        vals = torch.cat([p.sparse_features.values().view(F, -1) for p in parts], dim=1).reshape(-1)  # feature-major
        kjt = KeyedJaggedTensor.from_fixed_lengths(self.keys, vals, [L] * F)
        return Batch(torch.cat([p.dense_features for p in parts]), kjt, torch.cat([p.labels for p in parts]))

    def _ids(self, high: int, n: int) -> torch.Tensor:
        if self.zipf_alpha is None:
            return torch.randint(0, high, (n,), generator=self.gen, device=self.device, dtype=torch.int64)
        # inverse-CDF Zipf(alpha) truncated to [0, high): skewed like real Criteo ids
        u = torch.rand(n, generator=self.gen, device=self.device, dtype=torch.float64)
        a = 1.0 - self.zipf_alpha
        x = ((u * (float(high) ** a - 1.0) + 1.0) ** (1.0 / a)).floor().clamp_(1, high).to(torch.int64) - 1
        return x

    def _generate(self) -> Batch:
        B, L = self.batch_size, self.ids_per_feature
        values = torch.cat([self._ids(h, L * B) for h in self.hash_sizes])
        kjt = KeyedJaggedTensor.from_fixed_lengths(self.keys, values, [L] * len(self.keys))
        dense = torch.randn(B, self.num_dense, generator=self.gen, device=self.device)
        labels = torch.randint(0, 2, (B,), generator=self.gen, device=self.device)
        return Batch(dense, kjt, labels)

    def __iter__(self) -> Iterator[Batch]:
        i = 0
        while self.num_batches is None or i < self.num_batches:
            yield self._pool[i % len(self._pool)]
            i += 1

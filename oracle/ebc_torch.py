"""CPU baseline = the reference's unsharded design, restated: a dict of
``torch.nn.EmbeddingBag(mode="sum", include_last_offset=True)`` looped per table / feature and
``torch.cat(dim=1)`` (torchrec/modules/embedding_modules.py:149-156, 174-193), plus the
``sparse=True`` bags + ``torch.optim.SGD`` the reference's `sparse` compute kernel uses for
training (torchrec/distributed/embedding_kernel.py:221-257).

TEST INFRASTRUCTURE / cpu_baseline only (never imported by the product).  Equivalence with the
reference module itself is pinned by tests/test_oracle_golden.py::test_ebc_torch_matches_golden.
"""
import time
from typing import List, Optional

import torch
from torch import nn


class RefEmbeddingBagCollection(nn.Module):
    def __init__(self, rows: List[int], dims: List[int], pooling: str = "sum", sparse: bool = False) -> None:
        super().__init__()
        self.embedding_bags = nn.ModuleDict({
            f"t{i}": nn.EmbeddingBag(num_embeddings=r, embedding_dim=d, mode=pooling, include_last_offset=True,
                                     sparse=sparse)
            for i, (r, d) in enumerate(zip(rows, dims))})
        self.F = len(rows)

    def forward(self, values: torch.Tensor, offsets: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """values/offsets: feature-major KJT arrays (offsets has F*B+1 entries)."""
        B = (offsets.numel() - 1) // self.F
        pooled = []
        for f in range(self.F):
            o = offsets[f * B:(f + 1) * B + 1]
            s, e = int(o[0]), int(o[-1])
            pooled.append(self.embedding_bags[f"t{f}"](
                input=values[s:e], offsets=o - o[0],
                per_sample_weights=weights[s:e] if weights is not None else None))
        return torch.cat(pooled, dim=1)


def default_row_cap() -> int:
    """SURVEY.md §8d / BASELINE.md §3: tables are capped at 4 M rows only if the host has < ~100 GB of RAM
    (0 = no cap: the full 84.85 GiB of tables in host memory)."""
    try:
        import psutil

        total = psutil.virtual_memory().total
    except Exception:  # no psutil: be safe
        return 4 << 20
    return 0 if total >= 100e9 else 4 << 20


def time_cpu_baseline(rows: List[int], dim: int, batches=(4096, 65536), seconds_budget: float = 24.0,
                      row_cap: Optional[int] = None, seed: int = 1234, min_train_iters: int = 10, min_fwd_iters: int = 5):
    """Times forward and forward+backward+SGD of the reference design on the host cores with the same id
    distribution as the GPU run (uniform, pooling factor 1), at every batch size in `batches` (one model,
    built once).  `row_cap`: None = default_row_cap(), 0 = full tables.  Tables are filled with a constant
    (every page is written once, so lookups touch real memory; the values do not matter for timing).
    Per batch size: one untimed forward and one untimed train iteration, then at least `min_fwd_iters` forwards and
    `min_train_iters` train iterations (more while the time budget lasts), each timed on its own; the throughput
    quoted is batch / MEDIAN iteration time (a mean over 4 iterations moved by 2x between boxes: VERDICT round 2),
    the spread is reported next to it.
    Returns {"cores", "row_cap", "build_s", "per_batch": {batch: {...}}}."""
    g = torch.Generator()
    g.manual_seed(seed)
    if row_cap is None:
        row_cap = default_row_cap()
    capped = [min(r, row_cap) if row_cap else r for r in rows]
    F = len(rows)
    t0 = time.perf_counter()
    ebc = RefEmbeddingBagCollection.__new__(RefEmbeddingBagCollection)
    nn.Module.__init__(ebc)
    ebc.F = F
    bags = {}
    for i, r in enumerate(capped):
        w = torch.empty(r, dim)
        w.fill_(0.01)
        bags[f"t{i}"] = nn.EmbeddingBag(num_embeddings=r, embedding_dim=dim, mode="sum", include_last_offset=True,
                                        sparse=True, _weight=w)
    ebc.embedding_bags = nn.ModuleDict(bags)
    build_s = time.perf_counter() - t0
    opt = torch.optim.SGD(ebc.parameters(), lr=0.01)
    per_batch = {}
    share = seconds_budget / max(len(batches), 1)

    def median(xs):
        xs = sorted(xs)
        n = len(xs)
        return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])

    for batch in batches:
        values = torch.cat([torch.randint(0, r, (batch,), generator=g) for r in capped])
        offsets = torch.arange(F * batch + 1)
        grad = torch.randn(batch, F * dim, generator=g)

        def train_iter():
            opt.zero_grad()
            ebc(values, offsets).backward(grad)
            opt.step()

        with torch.no_grad():
            ebc(values, offsets)  # warm-up
        train_iter()  # warm-up
        fwd_t, t_begin = [], time.perf_counter()
        with torch.no_grad():
            while len(fwd_t) < min_fwd_iters or time.perf_counter() - t_begin < share / 4:
                t0 = time.perf_counter()
                ebc(values, offsets)
                fwd_t.append(time.perf_counter() - t0)
        train_t, t_begin = [], time.perf_counter()
        while len(train_t) < min_train_iters or time.perf_counter() - t_begin < share * 3 / 4:
            t0 = time.perf_counter()
            train_iter()
            train_t.append(time.perf_counter() - t0)
        per_batch[batch] = {"fwd_samples_per_s": batch / median(fwd_t), "train_samples_per_s": batch / median(train_t),
                            "fwd_iters": len(fwd_t), "train_iters": len(train_t),
                            "train_samples_per_s_min_max": [batch / max(train_t), batch / min(train_t)]}
    return {"cores": torch.get_num_threads(), "row_cap": row_cap, "build_s": build_s, "per_batch": per_batch}

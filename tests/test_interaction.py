"""DLRM dot interaction: oracle vs the reference formula / golden DLRM (CPU), HIP kernel vs oracle
(GPU, bit-exact: the MFMA f32 result is a k-ordered fmaf chain, which the oracle restates)."""
import os

import numpy as np
import pytest
import torch

import _paths  # noqa: F401
from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def torch_reference(dense, sparse):
    # torchrec/models/dlrm.py:206-219 restated on torch ops
    F = sparse.shape[1]
    combined = torch.cat((dense.unsqueeze(1), sparse), dim=1)
    inter = torch.bmm(combined, combined.transpose(1, 2))
    tri = torch.triu_indices(F + 1, F + 1, offset=1)
    return torch.cat((dense, inter[:, tri[0], tri[1]]), dim=1)


@pytest.mark.parametrize("B,F,D", [(5, 26, 128), (3, 2, 16), (4, 7, 32), (2, 27, 64), (1, 1, 4), (3, 31, 12)])
def test_oracle_matches_reference_formula(B, F, D):
    rng = np.random.default_rng(B * 100 + F)
    dense = rng.standard_normal((B, D)).astype(np.float32)
    sparse = rng.standard_normal((B, F, D)).astype(np.float32)
    td, ts = torch.from_numpy(dense).requires_grad_(), torch.from_numpy(sparse).requires_grad_()
    ref = torch_reference(td, ts)
    out = oracle.interaction_forward(dense, sparse)
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    go = rng.standard_normal(out.shape).astype(np.float32)
    ref.backward(torch.from_numpy(go))
    gd, gs = oracle.interaction_backward(dense, sparse, go)
    np.testing.assert_allclose(gd, td.grad.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(gs, ts.grad.numpy(), rtol=1e-5, atol=2e-5)


def test_dlrm_model_state_dict_keys_and_logits_match_reference_golden():
    """The mirror DLRM (torchrec_amd.models.dlrm) loads the REFERENCE model's state_dict by key and
    reproduces its logits (golden: reference DLRM.forward, models/dlrm.py:387-406) when fed the
    reference's pooled embeddings (computed by the oracle from the same tables)."""
    from torchrec_amd.models.dlrm import DenseArch, InteractionArch, OverArch

    g = np.load(os.path.join(GOLD, "dlrm_small.npz"))
    B, D = int(g["B"]), int(g["D"])
    rows = g["rows"].tolist()
    F = len(rows)
    tabs = oracle.Tables(rows, [D] * F)
    for i in range(F):
        tabs.weights[i][...] = g[f"sd::sparse_arch.embedding_bag_collection.embedding_bags.t{i}.weight"]
    offsets = np.concatenate([[0], np.cumsum(g["lengths"])]).astype(np.int64)
    pooled, _ = oracle.tbe_forward(tabs, g["values"], offsets)
    dense_arch = DenseArch(13, [16, D])
    over_arch = OverArch(D + (F + 1) * F // 2, [12, 1])
    inter = InteractionArch(F)
    sd_dense = {k[len("sd::dense_arch."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::dense_arch.")}
    sd_over = {k[len("sd::over_arch."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::over_arch.")}
    dense_arch.load_state_dict(sd_dense, strict=True)
    over_arch.load_state_dict(sd_over, strict=True)
    with torch.no_grad():
        e = dense_arch(torch.from_numpy(g["dense"]))
        x = inter(e, torch.from_numpy(pooled).view(B, F, D))
        logits = over_arch(x)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-5, atol=1e-5)
    # and the oracle interaction agrees with the torch formulation on the same activations
    np.testing.assert_allclose(oracle.interaction_forward(e.numpy(), pooled.reshape(B, F, D)), x.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("pad_rows", [False, True], ids=["dense_rows", "padded_rows"])
@pytest.mark.parametrize("B,F,D", [(65, 26, 128), (1, 26, 128), (300, 2, 16), (17, 7, 32), (9, 27, 64), (1030, 26, 128),
                                   (8192, 26, 128), (4099, 27, 128), (2600, 5, 64), (257, 26, 128)])
def test_hip_interaction_bit_exact_vs_oracle(B, F, D, pad_rows):
    """pad_rows: the output is a [B, D + P] view of a buffer whose rows are padded to a multiple of 4 floats (16-B
    stores); the gradient comes back both as a dense tensor and as a row-padded view."""
    from torchrec_amd.models.dlrm import InteractionArch

    rng = np.random.default_rng(B + F + D)
    dense = rng.standard_normal((B, D)).astype(np.float32)
    sparse = rng.standard_normal((B, F, D)).astype(np.float32)
    td = torch.from_numpy(dense).cuda().requires_grad_()
    ts = torch.from_numpy(sparse).cuda().requires_grad_()
    arch = InteractionArch(F).cuda()
    arch.pad_rows = pad_rows
    out = arch(td, ts)
    width = D + (F + 1) * F // 2
    assert tuple(out.shape) == (B, width)
    if pad_rows and width % 4:
        assert out.stride(0) == D + ((F + 1) * F // 2 + 3) // 4 * 4 and out.stride(1) == 1 and out.data_ptr() % 16 == 0
        pad = torch.as_strided(out, (B, out.stride(0) - width), (out.stride(0), 1), out.storage_offset() + width)
        assert float(pad.abs().max()) == 0.0  # pad columns are written as zeros
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.interaction_forward(dense, sparse))
    go = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    g_dev = torch.from_numpy(go).cuda()
    if pad_rows:  # the gradient as a row-padded view too (what the first over-arch layer hands back)
        gbuf = torch.full((B, out.stride(0)), float("nan"), device="cuda")
        gbuf[:, :width] = g_dev
        g_dev = gbuf[:, :width]
    out.backward(g_dev)
    gd, gs = oracle.interaction_backward(dense, sparse, go)
    np.testing.assert_array_equal(td.grad.cpu().numpy(), gd)
    np.testing.assert_array_equal(ts.grad.cpu().numpy(), gs)
    # and within tolerance of the reference's torch formulation
    arch.fused = False
    td2, ts2 = td.detach().clone().requires_grad_(), ts.detach().clone().requires_grad_()
    ref = arch(td2, ts2)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-4)
    ref.backward(torch.from_numpy(go).cuda())
    torch.testing.assert_close(td.grad, td2.grad, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(ts.grad, ts2.grad, rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_interaction_forward_full_batch_matches_torch_formulation():
    """Full-size launch (batch 65 536, F = 26, D = 128): the last samples of the batch — the last iterations of every
    wave's sample loop — against the reference's bmm + triu formulation."""
    import torch

    from torchrec_amd.models.dlrm import _FusedDotInteraction

    dense = torch.randn(65536, 128, device="cuda")
    sparse = torch.randn(65536, 26, 128, device="cuda")
    out = _FusedDotInteraction.apply(dense, sparse)
    for sl in (slice(0, 256), slice(65536 - 512, 65536)):
        x = torch.cat([dense[sl, None, :], sparse[sl]], dim=1)
        z = torch.bmm(x, x.transpose(1, 2))
        iu = torch.triu_indices(27, 27, offset=1, device="cuda")
        torch.testing.assert_close(out[sl, 128:], z[:, iu[0], iu[1]], rtol=1e-5, atol=1e-4)
        assert torch.equal(out[sl, :128], dense[sl])


@pytest.mark.gpu
def test_interaction_refuses_mismatched_batches_before_launch():
    """Shapes are checked on the host: a kernel launched with mismatched operands would read out of bounds."""
    import torch

    from torchrec_amd.models.dlrm import _FusedDotInteraction

    dense = torch.randn(8, 128, device="cuda")
    sparse = torch.randn(16, 26, 128, device="cuda")
    with pytest.raises(RuntimeError, match="does not match"):
        _FusedDotInteraction.apply(dense, sparse)

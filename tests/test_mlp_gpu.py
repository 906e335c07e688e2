"""GPU: the MLP layer with epilogue-fused bias+ReLU and split-K weight gradient equals the plain
torch formulation (torchrec/modules/mlp.py Perceptron = relu(linear(x)))."""
import pytest
import torch

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,I,O", [(4096, 512, 256), (8192, 13, 512), (2048, 479, 1024), (65536, 256, 128)])
def test_perceptron_matches_plain_torch(B, I, O):
    from torchrec_amd.modules.mlp import Perceptron

    torch.manual_seed(0)
    p = Perceptron(I, O, device=torch.device("cuda"))
    x = torch.randn(B, I, device="cuda", requires_grad=True)
    y = p(x)
    g = torch.randn_like(y)
    y.backward(g)
    x2 = x.detach().clone().requires_grad_()
    w2, b2 = p._linear.weight.detach().clone().requires_grad_(), p._linear.bias.detach().clone().requires_grad_()
    y2 = torch.relu(torch.nn.functional.linear(x2, w2, b2))
    y2.backward(g)
    torch.testing.assert_close(y, y2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(p._linear.weight.grad, w2.grad, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(p._linear.bias.grad, b2.grad, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("B,N", [(65536, 1024), (8192, 128), (1000, 52), (3, 256), (70, 4)])
def test_relu_backward_bias_grad_one_pass(B, N):
    """csrc/mlp_epilogue.hip: masked gradient is bit-exact (a select), column sums to fp32 summation-order
    tolerance against a float64 sum; two runs are bitwise identical (fixed order, no atomics)."""
    import torchrec_amd.distributed._device_ops  # noqa: F401

    torch.manual_seed(1)
    gy = torch.randn(B, N, device="cuda")
    act = torch.relu(torch.randn(B, N, device="cuda"))
    gx, gb = torch.ops.tbe_hip.relu_backward_bias_grad(gy, act)
    ref = gy * (act > 0)
    assert torch.equal(gx, ref)
    torch.testing.assert_close(gb, ref.double().sum(0).float(), rtol=1e-4, atol=1e-3)
    gx2, gb2 = torch.ops.tbe_hip.relu_backward_bias_grad(gy, act)
    assert torch.equal(gb, gb2) and torch.equal(gx, gx2)


def test_tuned_gemm_replay_keeps_results():
    """torchrec_amd/tuning: replaying the recorded hipBLASLt / rocBLAS kernel choice changes which fp32 GEMM
    kernel runs, not what is computed: forward, dgrad, wgrad and bias gradient stay within fp32 rounding of
    a float64 reference (whose ReLU mask is the path's own: a pre-activation within rounding of 0 may land
    on either side).  tools/check_tuned_gemms.py audits every recorded shape this way."""
    import torch.cuda.tunable as tunable

    from torchrec_amd.modules.mlp import Perceptron
    from torchrec_amd.tuning import enable_tuned_gemms

    torch.manual_seed(0)
    p = Perceptron(479, 1024, device=torch.device("cuda"))
    x = torch.randn(8192, 479, device="cuda")
    g = torch.randn(8192, 1024, device="cuda")

    def run():
        xi = x.clone().requires_grad_()
        p.zero_grad()
        y = p(xi)
        y.backward(g)
        return y.detach().clone(), xi.grad.clone(), p._linear.weight.grad.clone(), p._linear.bias.grad.clone()

    def errors(res):
        w, b = p._linear.weight.detach().double(), p._linear.bias.detach().double()
        y = torch.relu(x.double() @ w.t() + b)
        gm = g.double() * (res[0] > 0)
        ref = (y, gm @ w, gm.t() @ x.double(), gm.sum(0))
        return [float((a.double() - r).abs().max() / r.abs().mean()) for a, r in zip(res, ref)]

    base = errors(run())
    ok = enable_tuned_gemms()
    try:
        assert ok, "the recorded GEMM choices must load on the image they were recorded on"
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        tuned = errors(run())
    finally:
        tunable.enable(False)
    assert max(base) < 1e-4 and max(tuned) < 1e-4, (base, tuned)


@pytest.mark.parametrize("B,N", [(65536, 256), (8192, 256), (1000, 52), (5, 8)])
def test_weighted_colsum_and_one_output_linear(B, N):
    """csrc/mlp_epilogue.hip wcolsum: dW of a one-output Linear in one pass; the LinearOut module equals nn.Linear."""
    import torchrec_amd.distributed._device_ops  # noqa: F401
    from torchrec_amd.modules.mlp import LinearOut

    torch.manual_seed(2)
    x = torch.randn(B, N, device="cuda")
    w = torch.randn(B, device="cuda")
    out = torch.ops.tbe_hip.weighted_colsum(x, w)
    ref = (x.double() * w.double()[:, None]).sum(0)
    assert float((out.double() - ref).abs().max() / ref.abs().mean()) < 1e-4
    assert torch.equal(out, torch.ops.tbe_hip.weighted_colsum(x, w))  # fixed order: bitwise reproducible
    lin = LinearOut(N, 1, device=torch.device("cuda"))
    ref_lin = torch.nn.Linear(N, 1, device=torch.device("cuda"))
    ref_lin.load_state_dict(lin.state_dict())
    xi, xr = x.clone().requires_grad_(), x.clone().requires_grad_()
    g = torch.randn(B, 1, device="cuda")
    lin(xi).backward(g)
    ref_lin(xr).backward(g)
    torch.testing.assert_close(xi.grad, xr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lin.weight.grad, ref_lin.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(lin.bias.grad, ref_lin.bias.grad, rtol=1e-4, atol=1e-3)


def test_wgrad_on_side_stream_matches_inline():
    """modules/mlp.py _WgradOverlap: weight gradients computed on the side stream and accumulated out of band are
    bit-identical to the ones autograd accumulates (same kernels, same order), also when a gradient already exists."""
    from torchrec_amd.modules.mlp import MLP, _WgradOverlap

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    mlp = MLP(96, [256, 128, 64], device=dev)
    x = torch.randn(8192, 96, device=dev, requires_grad=True)
    g = torch.randn(8192, 64, device=dev)

    def run(overlap, passes):
        for p in mlp.parameters():
            p.grad = None
        x.grad = None
        for _ in range(passes):
            if overlap:
                _WgradOverlap.enable(dev)
            try:
                mlp(x).backward(g)
            finally:
                if overlap:
                    _WgradOverlap.disable()  # joins
        torch.cuda.synchronize()
        return [p.grad.clone() for p in mlp.parameters()] + [x.grad.clone()]

    for passes in (1, 2):
        want = run(False, passes)
        got = run(True, passes)
        assert not _WgradOverlap.on and not _WgradOverlap.pending
        for a, b in zip(want, got):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,dtype", [(8192, torch.int64), (65536, torch.int64), (65536, torch.float32), (1, torch.float32),
                                     (3001, torch.int64), (1 << 20, torch.float32)])
def test_fused_bce_with_logits_matches_torch(B, dtype):
    """csrc/mlp_epilogue.hip bce_with_logits_kernel: nn.BCEWithLogitsLoss (mean; the reference's train wrapper,
    examples/dlrm/modules/dlrm_train.py) and its gradient in ONE launch, labels int64 or float; deterministic."""
    from torchrec_amd.models.dlrm import bce_with_logits_mean

    torch.manual_seed(B)
    x = (torch.randn(B, device="cuda") * 6).requires_grad_()
    y = torch.randint(0, 2, (B,), device="cuda").to(dtype)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    loss = bce_with_logits_mean(loss_fn, x, y)
    (3.0 * loss).backward()
    x2 = x.detach().clone().requires_grad_()
    ref = loss_fn(x2, y.float())
    (3.0 * ref).backward()
    torch.testing.assert_close(loss, ref, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(x.grad, x2.grad, rtol=2e-6, atol=1e-9)
    again = bce_with_logits_mean(loss_fn, x.detach(), y)
    assert torch.equal(again, loss.detach())  # fixed summation order
    # everything the fused kernel does not cover goes to torch's
    w = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(2.0, device="cuda"))
    torch.testing.assert_close(bce_with_logits_mean(w, x.detach(), y), w(x.detach(), y.float()))


def test_multi_chunk_sum_finishes_every_gradient_in_one_launch():
    """csrc/mlp_epilogue.hip multi_chunk_sum_kernel: split-K chunk sums, row-block bias sums, complete gradients and
    gradient-less parameters (chunks 32 / 128 / 1 / 0; numel multiples of 4 and not; misaligned destinations) summed in
    fixed order, scaled, into one flat buffer."""
    import torchrec_amd.distributed._device_ops  # noqa: F401

    torch.manual_seed(4)
    shapes = [(32, 1024 * 479), (128, 256), (1, 512 * 13), (0, 64), (7, 1), (5, 1023), (1, 3), (256, 128)]
    srcs = [torch.randn(max(c, 1), n, device="cuda") for c, n in shapes]
    total = sum(n for _, n in shapes)
    dst = torch.full((total + 8,), 7.0, device="cuda")[4:4 + total]  # 16-B aligned start, unaligned segments inside
    rows, off = [], 0
    for (c, n), s in zip(shapes, srcs):
        rows.append([s.data_ptr(), c, n, off])
        off += n
    table = torch.tensor(rows, dtype=torch.int64, device="cuda")
    torch.ops.tbe_hip.multi_chunk_sum(table, len(rows), max(n for _, n in shapes), dst, 0.5)
    off = 0
    for (c, n), s in zip(shapes, srcs):
        ref = (s[:c].double().sum(0) * 0.5).float() if c else torch.zeros(n, device="cuda")
        torch.testing.assert_close(dst[off:off + n], ref, rtol=2e-5, atol=2e-5)
        off += n
    first = dst.clone()
    torch.ops.tbe_hip.multi_chunk_sum(table, len(rows), max(n for _, n in shapes), dst, 0.5)
    assert torch.equal(first, dst)


@pytest.mark.parametrize("B,N", [(8192, 1024), (8192, 128), (65536, 256), (1000, 52)])
def test_row_block_partials_sum_to_the_one_pass_results(B, N):
    import torchrec_amd.distributed._device_ops  # noqa: F401

    torch.manual_seed(2)
    gy = torch.randn(B, N, device="cuda")
    act = torch.relu(torch.randn(B, N, device="cuda"))
    gx, gb = torch.ops.tbe_hip.relu_backward_bias_grad(gy, act)
    gx2, part = torch.ops.tbe_hip.relu_backward_bias_partials(gy, act)
    assert torch.equal(gx, gx2) and part.shape[1] == N
    # the same row-block sums, added in float64 here and in fp32 (another order) by the second-stage kernel
    torch.testing.assert_close(part.double().sum(0).float(), gb, rtol=1e-4, atol=1e-3)
    w = torch.randn(B, device="cuda")
    torch.testing.assert_close(torch.ops.tbe_hip.weighted_colsum_partials(gy, w).double().sum(0).float(),
                               torch.ops.tbe_hip.weighted_colsum(gy, w), rtol=1e-4, atol=1e-3)


def test_deferred_finish_of_an_eager_backward_equals_the_plain_path():
    """modules/mlp.py _DeferredFinish: with the switch on, an eager backward leaves the split-K weight gradients and the
    bias gradients to ONE launch (segment table by value in the kernel arguments) that also attaches them as .grad
    (accumulating into an existing one); same values as the plain path to fp32 summation order, nothing pending afterwards,
    off again after disable()."""
    from torchrec_amd.modules.mlp import MLP, LinearOut, _DeferredFinish

    def run(deferred):
        torch.manual_seed(3)
        mlp = MLP(479, [1024, 512, 256], device=torch.device("cuda"))
        last = LinearOut(256, 1, device=torch.device("cuda"))
        x = torch.randn(16384, 479, device="cuda")
        y = last(mlp(x))
        if deferred:
            _DeferredFinish.enable()
        try:
            y.sum().backward()
            if deferred:
                assert len(_DeferredFinish.pending) >= 6  # 3 split-K weights, 3 biases, the one-output weight
        finally:
            if deferred:
                _DeferredFinish.disable()
        assert not _DeferredFinish.pending and not _DeferredFinish.on
        return [p.grad.clone() for p in list(mlp.parameters()) + list(last.parameters())]

    a, b = run(False), run(True)
    for g0, g1 in zip(a, b):
        torch.testing.assert_close(g1, g0, rtol=2e-4, atol=2e-3)
    # accumulation into an existing .grad
    lin = LinearOut(256, 1, device=torch.device("cuda"))
    x = torch.randn(4096, 256, device="cuda")
    lin(x).sum().backward()
    once = lin.weight.grad.clone()
    _DeferredFinish.enable()
    lin(x).sum().backward()
    _DeferredFinish.disable()
    torch.testing.assert_close(lin.weight.grad, 2 * once, rtol=2e-4, atol=2e-3)

"""optim/flat.py FlatSGD (CPU): one `add_` over flat parameter / gradient buffers equals torch.optim.SGD on the same
parameters, and falls back to per-parameter updates when a gradient is not the flat view."""
import torch

import _paths  # noqa: F401
from torchrec_amd.optim.flat import FlatSGD
from torchrec_amd.optim.keyed import KeyedOptimizerWrapper


def _params(seed):
    torch.manual_seed(seed)
    shapes = [(7, 5), (5,), (3, 7), (1,), (4, 4)]
    return [torch.nn.Parameter(torch.randn(s)) for s in shapes]


def _flatten(params):
    n = sum(p.numel() for p in params)
    flat_p, flat_g = torch.zeros(n), torch.zeros(n)
    views, off = [], 0
    with torch.no_grad():
        for p in params:
            v = flat_p[off:off + p.numel()].view_as(p)
            v.copy_(p)
            p.data = v
            views.append(flat_g[off:off + p.numel()].view_as(p))
            off += p.numel()
    return flat_p, flat_g, views


def test_flat_sgd_equals_torch_sgd_and_falls_back():
    ref, mine = _params(0), _params(0)
    extra_ref, extra_mine = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(3))  # not in the flat buffer
    flat_p, flat_g, views = _flatten(mine)
    opt_ref = torch.optim.SGD(ref + [extra_ref], lr=0.1)
    opt = KeyedOptimizerWrapper({f"p{i}": p for i, p in enumerate(mine + [extra_mine])},
                                lambda ps: FlatSGD(ps, 0.1, flat_param=flat_p, flat_grad=flat_g, covered=mine, grad_views=views))
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        grads = [torch.randn(p.shape, generator=g) for p in ref]
        ge = torch.randn(3, generator=g)
        for p, x in zip(ref, grads):
            p.grad = x.clone()
        extra_ref.grad = ge.clone()
        if step == 2:  # gradients NOT delivered through the flat views: the per-parameter path must take over
            for p, x in zip(mine, grads):
                p.grad = x.clone()
        else:
            for p, v, x in zip(mine, views, grads):
                v.copy_(x)
                p.grad = v
        extra_mine.grad = ge.clone()
        opt_ref.step()
        opt.step()
        opt.zero_grad()
        for a, b in zip(ref + [extra_ref], mine + [extra_mine]):
            assert torch.equal(a.detach(), b.detach()), step
    assert all(p.grad is None for p in mine)
    # the parameters still live in the flat buffer
    assert mine[0].data_ptr() == flat_p.data_ptr()


def test_flat_sgd_rejects_foreign_coverage():
    a, b = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    try:
        FlatSGD([a], 0.1, flat_param=torch.zeros(2), flat_grad=torch.zeros(2), covered=[b], grad_views=[torch.zeros(2)])
    except ValueError:
        return
    raise AssertionError("expected ValueError")


def test_flat_sgd_momentum_and_weight_decay_equal_torch_sgd_on_both_paths():
    """hyper-parameters of the torch.optim.SGD it replaces are honoured (or rejected), not ignored."""
    import pytest

    ref, mine = _params(3), _params(3)
    flat_p, flat_g, views = _flatten(mine)
    kw = dict(lr=0.05, momentum=0.9, weight_decay=0.01)
    opt_ref = torch.optim.SGD(ref, **kw)
    opt = FlatSGD(mine, flat_param=flat_p, flat_grad=flat_g, covered=mine, grad_views=views, **kw)
    g = torch.Generator().manual_seed(5)
    for step in range(6):
        grads = [torch.randn(p.shape, generator=g) for p in ref]
        for p, x in zip(ref, grads):
            p.grad = x.clone()
        if step in (2, 3):  # per-parameter path on the SAME momentum state
            for p, x in zip(mine, grads):
                p.grad = x.clone()
        else:
            for p, v, x in zip(mine, views, grads):
                v.copy_(x)
                p.grad = v
        opt_ref.step()
        opt.step()
        for a, b in zip(ref, mine):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        FlatSGD(_params(0), 0.1, nesterov=True, momentum=0.9)
    with pytest.raises(ValueError):
        FlatSGD([{"params": _params(0), "lr": 0.1}, {"params": _params(1), "lr": 0.2}], 0.1)
    with pytest.raises(ValueError):
        opt.add_param_group({"params": _params(2)})


def test_flat_sgd_parameter_without_gradient_is_not_updated_from_a_stale_slice():
    """A covered parameter that got no gradient this step must not move, whatever an earlier step left in its slice of
    the flat gradient buffer (set_to_none=True: `.grad is None` -> per-parameter path skips it)."""
    mine = _params(4)
    flat_p, flat_g, views = _flatten(mine)
    opt = FlatSGD(mine, 0.1, flat_param=flat_p, flat_grad=flat_g, covered=mine, grad_views=views)
    for p, v in zip(mine, views):
        v.fill_(1.0)
        p.grad = v
    opt.step()
    opt.zero_grad(set_to_none=True)  # the slices still hold ones
    before = [p.detach().clone() for p in mine]
    views[1].fill_(2.0)
    mine[1].grad = views[1]  # only this one received a gradient
    opt.step()
    for i, (a, b) in enumerate(zip(before, mine)):
        if i == 1:
            assert torch.equal(b.detach(), a - 0.2)
        else:
            assert torch.equal(b.detach(), a), i

"""Fused fp32-MFMA chain kernels (dim=128) vs the plain-PyTorch fp32 formulation of the same ops on the same GPU and
vs an fp64 evaluation.  Tolerance: err(fused, fp64) <= max(2e-6, 2 * err(torch_fp32, fp64)) per tensor."""
import numpy as np
import pytest
import torch

from conftest import maxnorm_err

pytestmark = pytest.mark.gpu
D = 128


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _ok(a, b32, b64, floor=2e-6):
    e, f = maxnorm_err(a.detach().cpu(), b64.detach().cpu()), maxnorm_err(b32.detach().cpu(), b64.detach().cpu())
    return e <= max(floor, 2 * f), (e, f)


def _run(layer_fn, inputs, params, impl):
    from pamnet_amd import modules
    modules.IMPL = impl
    try:
        for t in list(inputs) + list(params):
            t.grad = None
        outs = layer_fn()
        gen = torch.Generator().manual_seed(7)
        w = [torch.randn(o.shape, generator=gen, dtype=torch.float64).to(o.dtype).to(o.device) for o in outs]
        sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
        return ([o.detach().clone() for o in outs],
                [None if t.grad is None else t.grad.detach().clone() for t in list(inputs) + list(params)])
    finally:
        modules.IMPL = 'fused'


@pytest.mark.parametrize('n', [1, 16, 37, 2286])
def test_node_tail_and_pre(dev, n):
    from pamnet_amd import modules
    torch.manual_seed(n)
    g32 = modules.GlobalMP(D).to(dev)
    l32 = modules.LocalMP(D).to(dev)
    x = torch.randn(n, D, device=dev, requires_grad=True)
    r = torch.randn(n, D, device=dev, requires_grad=True)
    for layer in (g32, l32):
        params = [p for p in layer.parameters()]
        fn = lambda: modules.update_and_heads(layer, x, r)
        of, gf = _run(fn, [x, r], params, 'fused')
        ot, gt = _run(fn, [x, r], params, 'torch')
        l64 = type(layer)(D).to(dev).double()
        l64.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
        x64, r64 = x.detach().double().requires_grad_(), r.detach().double().requires_grad_()
        p64 = [p for p in l64.parameters()]
        o64, g64 = _run(lambda: modules.update_and_heads(l64, x64, r64), [x64, r64], p64, 'torch')
        names = ['x', 'res_x'] + [k for k, _ in layer.named_parameters()]
        for a, b, c in zip(of, ot, o64):
            ok, info = _ok(a, b, c)
            assert ok, ('out', info)
        used = {id(p) for p in modules.fused.tail_params(layer)} if hasattr(modules, 'fused') else set()
        for nm, a, b, c in zip(names, gf, gt, g64):
            if a is None or c is None:
                continue
            ok, info = _ok(a, b, c, floor=5e-6)
            assert ok, (nm, info)


@pytest.mark.parametrize('n', [5, 2286])
def test_node_pre(dev, n):
    from pamnet_amd import fused
    torch.manual_seed(3)
    lin = torch.nn.Linear(D, D).to(dev)
    wm = torch.randn(D, 3 * D, device=dev, requires_grad=True)
    wk = torch.randn(D, 3 * D, device=dev, requires_grad=True)
    x = torch.randn(n, D, device=dev, requires_grad=True)
    blocks = lambda: [wm[:, :D], wk[:, :D], wm[:, D:2 * D], wk[:, D:2 * D]]
    x1, P = fused.node_pre(x, lin, blocks(), 3 * D)
    w1, w2 = torch.randn_like(x1), torch.randn_like(P)
    ((x1 * w1).sum() + (P * w2).sum()).backward()
    got = [x1.detach(), P.detach(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone(), wm.grad.clone(), wk.grad.clone()]
    res = {}
    for dt in (torch.float32, torch.float64):
        xx = x.detach().to(dt).requires_grad_()
        W, b = lin.weight.detach().to(dt).requires_grad_(), lin.bias.detach().to(dt).requires_grad_()
        m, k = wm.detach().to(dt).requires_grad_(), wk.detach().to(dt).requires_grad_()
        a = torch.nn.functional.silu(xx @ W.t() + b)
        p = a @ torch.cat([m[:, :D], k[:, :D], m[:, D:2 * D], k[:, D:2 * D]], 0).t()
        ((a * w1.to(dt)).sum() + (p * w2.to(dt)).sum()).backward()
        res[dt] = [a.detach(), p.detach(), xx.grad, W.grad, b.grad, m.grad, k.grad]
    for a, b, c in zip(got, res[torch.float32], res[torch.float64]):
        ok, info = _ok(a, b, c, floor=5e-6)
        assert ok, info

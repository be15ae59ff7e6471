"""GPU parity tests of the narrow-width (dim 16 / 32 / 64) row kernels (csrc/narrow.hip) against fp64 torch
restatements of the reference formulas (layers/global_message_passing.py:52-53, layers/local_message_passing.py:49,
models.py:185-188).  Tolerance: max-normalised error <= 1e-5 (north_star), forward and every gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import maxnorm_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs an MI355X'
    from pamnet_amd import lib
    lib.load()
    return torch.device('cuda:0')


def _graph(rng, n, max_deg, dev):
    """Random directed graph as CSR over targets + the transposed CSR over sources."""
    from pamnet_amd import graph as G
    lens = rng.integers(0, max_deg + 1, size=n)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    m = int(ptr[-1])
    row_of = np.repeat(np.arange(n), lens).astype(np.int32)
    col = rng.integers(0, n, m).astype(np.int32)
    csr = G.CSR(torch.from_numpy(ptr).to(dev), torch.from_numpy(row_of).to(dev), torch.from_numpy(col).to(dev))
    tr = G.Transpose(csr.col, n)
    return csr, tr


def _check_grads(params, ref_params, names):
    for p, r, nm in zip(params, ref_params, names):
        assert p.grad is not None, nm
        err = maxnorm_err(p.grad.cpu(), r.grad.cpu())
        assert err < TOL, (nm, err)


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('n,max_deg', [(5, 3), (300, 40), (1000, 9)])
def test_global_message(dev, d, n, max_deg):
    from pamnet_amd import narrow
    rng = np.random.default_rng(d * 7 + n)
    csr, tr = _graph(rng, n, max_deg, dev)
    m = csr.m
    torch.manual_seed(d + n)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.5).requires_grad_(True)
    x1, P, e, wm, bm, wea = mk(n, d), mk(n, 2 * d), mk(m, d), mk(d, 3 * d), mk(d), mk(d, d)
    out = narrow.global_message(x1, P, e, wm, bm, wea, csr, tr)
    gout = torch.randn_like(out)
    out.backward(gout)
    ref_in = [t.detach().double().requires_grad_(True) for t in (x1, P, e, wm, bm, wea)]
    rx1, rP, re, rwm, rbm, rwea = ref_in
    i, j = csr.row_of.long(), csr.col.long()
    z = rP[i, :d] + rP[j, d:] + re @ rwm[:, 2 * d:].t() + rbm
    msg = F.silu(z) * (re @ rwea.t())
    ref = rx1 + torch.zeros(n, d, dtype=torch.float64, device=dev).index_add_(0, i, msg)
    ref.backward(gout.double())
    assert maxnorm_err(out.detach().cpu(), ref.detach().cpu()) < TOL
    _check_grads((x1, P, e, wm, bm, wea), ref_in, ('x1', 'P', 'e', 'wm', 'bm', 'wea'))
    assert torch.equal(wm.grad[:, :2 * d], torch.zeros_like(wm.grad[:, :2 * d]))    # node blocks belong to P's producer
    # run-to-run bitwise determinism of the weight-gradient reduction
    for t in (x1, P, e, wm, bm, wea):
        t.grad = None
    g1 = None
    for _ in range(2):
        o = narrow.global_message(x1, P, e, wm, bm, wea, csr, tr)
        o.backward(gout)
        cur = [t.grad.clone() for t in (P, e, wm, bm, wea)]
        for t in (x1, P, e, wm, bm, wea):
            t.grad = None
        if g1 is not None:
            assert all(torch.equal(a, b) for a, b in zip(g1, cur))
        g1 = cur


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('m', [1, 17, 4099])
@pytest.mark.parametrize('res_x,with_r', [(False, False), (True, False), (True, True)])
def test_mlp2(dev, d, m, res_x, with_r):
    """mlp_sbf (plain) and the Res blocks (MLP2(x) + x [+ layer input], layers/basic.py:25-33)."""
    from pamnet_amd import narrow
    torch.manual_seed(d * 3 + m)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.4).requires_grad_(True)
    x, w1, b1, w2, b2, r = mk(m, d), mk(d, d), mk(d), mk(d, d), mk(d), mk(m, d)
    y = narrow._Mlp2.apply(x, w1, b1, w2, b2, res_x, r if with_r else None)
    g = torch.randn_like(y)
    y.backward(g)
    ref_in = [t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2, r)]
    rx, rw1, rb1, rw2, rb2, rr = ref_in
    ref = F.silu(F.linear(F.silu(F.linear(rx, rw1, rb1)), rw2, rb2))
    if res_x:
        ref = ref + rx
    if with_r:
        ref = ref + rr
    ref.backward(g.double())
    assert maxnorm_err(y.detach().cpu(), ref.detach().cpu()) < TOL
    n = 6 if with_r else 5
    _check_grads((x, w1, b1, w2, b2, r)[:n], ref_in[:n], ('x', 'w1', 'b1', 'w2', 'b2', 'r')[:n])


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('m', [1, 1000, 4099])
def test_linear_and_projection(dev, d, m):
    """One dense block (bias + SiLU) and the four node-side projections of the split 3d-wide message weights."""
    from pamnet_amd import narrow
    torch.manual_seed(d + m)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.4).requires_grad_(True)
    x, w, b = mk(m, d), mk(d, d), mk(d)
    lin = torch.nn.Linear(d, d).to(dev)
    with torch.no_grad():
        lin.weight.copy_(w), lin.bias.copy_(b)
    y = narrow.linear(x, lin, act=True)
    g = torch.randn_like(y)
    y.backward(g)
    rx, rw, rb = [t.detach().double().requires_grad_(True) for t in (x, w, b)]
    ref = F.silu(F.linear(rx, rw, rb))
    ref.backward(g.double())
    assert maxnorm_err(y.detach().cpu(), ref.detach().cpu()) < TOL
    _check_grads((x, lin.weight, lin.bias), (rx, rw, rb), ('x', 'w', 'b'))

    x2, wj, wk = mk(m, d), mk(d, 3 * d), mk(d, 3 * d)
    p = narrow.project(x2, ((0, 0), (1, 0), (0, d), (1, d)), wj, wk)
    g = torch.randn_like(p)
    p.backward(g)
    rx2, rwj, rwk = [t.detach().double().requires_grad_(True) for t in (x2, wj, wk)]
    ref = F.linear(rx2, torch.cat([rwj[:, :d], rwk[:, :d], rwj[:, d:2 * d], rwk[:, d:2 * d]], 0))
    ref.backward(g.double())
    assert maxnorm_err(p.detach().cpu(), ref.detach().cpu()) < TOL
    _check_grads((x2, wj, wk), (rx2, rwj, rwk), ('x', 'wj', 'wk'))


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('k,two', [(16, False), (42, False), (42, True)])
@pytest.mark.parametrize('m', [3, 1000, 4099])
def test_embed(dev, d, k, two, m):
    from pamnet_amd import narrow
    torch.manual_seed(d + k + m)
    rng = np.random.default_rng(m)
    need_df = (k == 16)
    f = torch.randn(m, k, device=dev).requires_grad_(need_df)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.3).requires_grad_(True)
    wa, ba, wb, bb = mk(d, k), mk(d), mk(d, k), mk(d)
    kind = torch.from_numpy(rng.integers(0, 2, m).astype(np.int32)).to(dev) if two else None
    y = narrow._Embed.apply(f, kind, wa, ba, wb if two else None, bb if two else None)
    g = torch.randn_like(y)
    y.backward(g)
    rf = f.detach().double().requires_grad_(need_df)
    rwa, rba, rwb, rbb = [t.detach().double().requires_grad_(True) for t in (wa, ba, wb, bb)]
    ya = F.silu(F.linear(rf, rwa, rba))
    if two:
        yb = F.silu(F.linear(rf, rwb, rbb))
        ref = torch.where(kind.bool().unsqueeze(1), yb, ya)
    else:
        ref = ya
    ref.backward(g.double())
    assert maxnorm_err(y.detach().cpu(), ref.detach().cpu()) < TOL
    names, ps, rs = ['wa', 'ba'], [wa, ba], [rwa, rba]
    if two:
        names, ps, rs = names + ['wb', 'bb'], ps + [wb, bb], rs + [rwb, rbb]
    if need_df:
        names, ps, rs = names + ['f'], ps + [f], rs + [rf]
    _check_grads(ps, rs, names)


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('m', [1, 1000, 4099])
def test_embed_rbf_in_kernel(dev, d, m):
    """The edge embedding on Bessel rows formed inside the kernels: the forward returns the floats of the two-step form
    (pamnet_rbf_fwd_f32 into [m, 16], then the k = 16 embedding); the gradients of the frequencies, W and b match the
    fp64 reference of the whole expression (and the two-step path to rounding).  Distances on both sides of the cutoff."""
    from pamnet_amd import narrow, ops
    torch.manual_seed(d + m)
    dist = torch.rand(m, device=dev) * 5.6 + 0.3
    freq0 = (torch.arange(1, 17, dtype=torch.float32, device=dev) * np.pi) + torch.randn(16, device=dev) * 0.01
    mk = lambda t: t.detach().clone().requires_grad_(True)
    lin = torch.nn.Linear(16, d).to(dev)
    f1, w1, b1 = mk(freq0), mk(lin.weight), mk(lin.bias)
    y1 = narrow._EmbedRbf.apply(dist, f1, 5.0, w1, b1)
    g = torch.randn_like(y1)
    y1.backward(g)
    f2, w2, b2 = mk(freq0), mk(lin.weight), mk(lin.bias)
    y2 = narrow._Embed.apply(ops.rbf(dist, f2, 5.0), None, w2, b2, None, None)
    y2.backward(g)
    assert torch.equal(y1, y2)
    # fp64 reference of SiLU(W (u(x) sin(f x)) + b)
    x = (dist.double() / 5.0)
    fr, wr, br = [mk(t.double()) for t in (freq0, lin.weight, lin.bias)]
    x5 = x ** 5
    u = torch.where(x < 1, 1.0 / x + x5 * (-21.0 + x * (35.0 - 15.0 * x)), torch.zeros_like(x))
    ref = F.silu(F.linear(u.unsqueeze(1) * torch.sin(fr * x.unsqueeze(1)), wr, br))
    ref.backward(g.double())
    _check_grads((f1, w1, b1), (fr, wr, br), ('freq', 'w', 'b'))
    _check_grads((f2, w2, b2), (fr, wr, br), ('freq (two-step)', 'w', 'b'))


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('n,max_deg', [(4, 2), (500, 7)])
def test_local_gate(dev, d, n, max_deg):
    """m_ji / m_nb of the local layer from node- and edge-side projections (layers/local_message_passing.py:46-48)."""
    from pamnet_amd import narrow
    rng = np.random.default_rng(d + n)
    csr, tr = _graph(rng, n, max_deg, dev)
    m = csr.m
    torch.manual_seed(d * n)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.6).requires_grad_(True)
    P, Q, b1, b2 = mk(n, 4 * d), mk(m, 4 * d), mk(d), mk(d)
    m_ji, m_nb = narrow.local_gate(P, Q, b1, b2, csr, tr)
    g1, g2 = torch.randn_like(m_ji), torch.randn_like(m_nb)
    torch.autograd.backward([m_ji, m_nb], [g1, g2])
    ref_in = [t.detach().double().requires_grad_(True) for t in (P, Q, b1, b2)]
    rP, rQ, rb1, rb2 = ref_in
    i, j = csr.row_of.long(), csr.col.long()
    z1 = rP[i, :d] + rP[j, 2 * d:3 * d] + rQ[:, :d] + rb1
    z2 = rP[i, d:2 * d] + rP[j, 3 * d:] + rQ[:, d:2 * d] + rb2
    r_ji, r_nb = F.silu(z1), F.silu(z2) * rQ[:, 2 * d:3 * d]
    torch.autograd.backward([r_ji, r_nb], [g1.double(), g2.double()])
    assert maxnorm_err(m_ji.detach().cpu(), r_ji.detach().cpu()) < TOL
    assert maxnorm_err(m_nb.detach().cpu(), r_nb.detach().cpu()) < TOL
    _check_grads((P, Q, b1, b2), ref_in, ('P', 'Q', 'b_ji', 'b_kj'))


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('n', [5, 1000, 2287, 70001])       # (> 65 536 rows: workgroups take a second row group and add)
def test_node_tail_and_heads(dev, d, n):
    """The whole node-update tail (10 dense layers, three residual blocks, both heads) as one autograd node against
    the fp64 composition of the reference formulas (layers/global_message_passing.py:39-50)."""
    import copy
    from pamnet_amd import modules, narrow
    torch.manual_seed(d + n)
    layer = modules.GlobalMP(d).to(dev)
    ref = copy.deepcopy(layer).double()
    x = (torch.randn(n, d, device=dev) * 0.7).requires_grad_(True)
    res = (torch.randn(n, d, device=dev) * 0.7).requires_grad_(True)
    xo, out, att = narrow.tail(layer, x, res)
    gx, go, ga = torch.randn_like(xo), torch.randn_like(out), torch.randn_like(att)
    torch.autograd.backward([xo, out, att], [gx, go, ga])
    rx, rres = x.detach().double().requires_grad_(True), res.detach().double().requires_grad_(True)
    import torch_formulation as T                        # (the plain-PyTorch statement of the tail: tests/torch_formulation.py)
    h, r_out, r_att = T.update_and_heads(ref, rx, rres)
    torch.autograd.backward([h, r_out, r_att], [gx.double(), go.double(), ga.double()])
    for a, b, nm in ((xo, h, 'x'), (out, r_out, 'out'), (att, r_att, 'att')):
        assert maxnorm_err(a.detach().cpu(), b.detach().cpu()) < TOL, nm
    assert maxnorm_err(x.grad.cpu(), rx.grad.cpu()) < TOL
    assert maxnorm_err(res.grad.cpu(), rres.grad.cpu()) < TOL
    rp = dict(ref.named_parameters())
    used = 0
    for nm, p in layer.named_parameters():
        if p.grad is None:
            continue                                       # message-side parameters are not part of the tail
        used += 1
        assert maxnorm_err(p.grad.cpu(), rp[nm].grad.cpu()) < TOL, nm
    assert used == 23


@pytest.mark.parametrize('dim,n_layer', [(16, 1), (64, 2)])
def test_rna_model_matches_generic_path(dev, dim, n_layer):
    """Whole model at the reference's RNA widths: narrow kernels vs the generic path (torch dense layers + HIP
    graph / basis / segment kernels), outputs and all parameter gradients."""
    import models
    from pamnet_amd import narrow, synth
    cfg = models.Config(dataset='rna_native', dim=dim, n_layer=n_layer, cutoff_l=2.6, cutoff_g=20.0,
                        flow='target_to_source')
    torch.manual_seed(3)
    model = models.PAMNet(cfg).to(dev)
    batch = synth.rna_batch(5, 0, 2).to(dev)
    res = {}
    for on in (True, False):
        narrow.ENABLED = on
        try:
            model.zero_grad()
            out = model(batch)
            out.sum().backward()
            res[on] = (out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()})
        finally:
            narrow.ENABLED = True
    assert maxnorm_err(res[True][0].cpu(), res[False][0].cpu()) < TOL
    for n, gr in res[True][1].items():
        err = maxnorm_err(gr.cpu(), res[False][1][n].cpu())
        # fp32 vs fp32: both sides sum ~10^5 edge rows in their own order (each side is checked against fp64 at 1e-5:
        # the kernel tests above and the oracle fixtures of test_hip_model.py)
        assert err < 1e-4, (n, err)


@pytest.mark.parametrize('dim,n_layer', [(16, 1), (32, 2), (64, 2)])
@pytest.mark.parametrize('mode', ['autograd', 'trainer'])
def test_engine_matches_per_operator_path(dev, dim, n_layer, mode):
    """csrc/narrow_engine.hip (node chains as single launches, gradients reduced straight into the parameter buffers, one
    call per direction) against the per-operator row kernels on the same model and batch: outputs, per-layer node features
    and every parameter gradient; under plain autograd and under the trainer's preallocated-gradient tape."""
    import models
    from pamnet_amd import narrow, synth
    from pamnet_amd.train import FlatParams
    cfg = models.Config(dataset='rna_native', dim=dim, n_layer=n_layer, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    torch.manual_seed(4)
    model = models.PAMNet(cfg).to(dev)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1 and p.numel() == dim:
                p.add_(0.05 * torch.randn_like(p))
    data = synth.rna_batch(2, 3, 3).to(dev)
    w = torch.randn(3, device=dev)
    if mode == 'trainer':
        fp = FlatParams(model, direct=True)
    params = [p for p in model.parameters()]

    def run(engine):
        narrow.ENGINE = engine
        try:
            if mode == 'trainer':
                fp.zero_grad()
            else:
                for p in params:
                    p.grad = None
            out = model(data)
            xs = [x.detach().clone() for x in model._x_layers]
            (out * w).sum().backward()
            return out.detach().clone(), xs, [p.grad.detach().clone() if p.grad is not None else None for p in params]
        finally:
            narrow.ENGINE = True

    out_e, xs_e, g_e = run(True)
    out_o, xs_o, g_o = run(False)
    assert maxnorm_err(out_e.cpu(), out_o.cpu()) < 2e-6
    assert len(xs_e) == len(xs_o) == 2 * n_layer
    for a, b in zip(xs_e, xs_o):
        assert maxnorm_err(a.cpu(), b.cpu()) < 2e-6
    names = [n for n, _ in model.named_parameters()]
    for nm, a, b in zip(names, g_e, g_o):
        assert (a is None) == (b is None), nm
        if a is None:
            continue
        scale = float(b.abs().max())
        if scale == 0.0:
            assert float(a.abs().max()) == 0.0, nm
            continue
        err = float((a - b).abs().max()) / scale
        assert err < 2e-5, (nm, err)
    # deterministic
    out_e2, _, g_e2 = run(True)
    assert torch.equal(out_e, out_e2)
    assert all(torch.equal(a, b) for a, b in zip(g_e, g_e2) if a is not None)

"""csrc/dense.hip: dense layers of any width as fp32-accurate GEMMs on the bf16 matrix pipe.

Kernel level: forward / dx / dW / db against a plain PyTorch product of the same operands in fp64 (and never worse than
2x torch's own fp32 GEMM on them), ragged shapes, every stride pattern, the split row sum of the weight gradient.
Model level: hidden sizes above 128 (models.py:25) -- every dense layer of the model on these kernels -- against the CPU
oracle's fp64 run, outputs and every parameter gradient, under the bounds of tests/test_hip_model.py.
"""
import ctypes

import pytest
import torch

from conftest import maxnorm_err
from test_hip_model import TOL, _check_gradients

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _bound(err, floor32, tol=2e-6):
    return err <= max(tol, 2 * floor32), (err, floor32)


SHAPES = [(1, 1, 1), (5, 42, 16), (37, 18, 136), (64, 64, 64), (65, 33, 63), (1000, 200, 132), (2286, 256, 256),
          (3, 160, 512), (70001, 136, 140)]


@pytest.mark.parametrize('n,k,m', SHAPES)
@pytest.mark.parametrize('act', [False, True])
@pytest.mark.parametrize('bias', [False, True])
def test_dense_forward_backward_vs_fp64(dev, n, k, m, act, bias):
    from pamnet_amd import ops
    gen = torch.Generator().manual_seed(1000 * n + 10 * k + m)
    x64 = torch.randn(n, k, generator=gen, dtype=torch.float64)
    w64 = torch.randn(m, k, generator=gen, dtype=torch.float64) / k ** 0.5
    b64 = torch.randn(m, generator=gen, dtype=torch.float64) if bias else None
    g64 = torch.randn(n, m, generator=gen, dtype=torch.float64)

    def run(dtype, device, own):
        x = x64.clone().to(dtype).to(device).requires_grad_(True)
        w = w64.clone().to(dtype).to(device).requires_grad_(True)
        b = b64.clone().to(dtype).to(device).requires_grad_(True) if bias else None
        if own:
            y = ops.dense(x, w, b, act=act)
        else:
            y = torch.nn.functional.linear(x, w, b)
            y = torch.nn.functional.silu(y) if act else y
        y.backward(g64.to(dtype).to(device))
        return [t.detach().cpu().double() for t in (y, x.grad, w.grad)] + ([b.grad.detach().cpu().double()] if bias else [])
    ref = run(torch.float64, 'cpu', False)
    t32 = run(torch.float32, dev, False)                       # plain PyTorch fp32 on the same device: the floor
    own = run(torch.float32, dev, True)
    for name, a, f, r in zip(('y', 'dx', 'dw', 'db'), own, t32, ref):
        ok, info = _bound(maxnorm_err(a, r), maxnorm_err(f, r))
        assert ok, (name, info)


def test_dense_without_gradients_and_partial_needs(dev):
    """No-grad mode calls the forward body directly (no z saved); an input that needs no gradient gets none; the weight
    gradient is bitwise reproducible (fixed-order reduction of the row splits)."""
    from pamnet_amd import ops
    torch.manual_seed(0)
    x = torch.randn(5000, 144, device=dev)
    w = torch.randn(160, 144, device=dev, requires_grad=True)
    b = torch.randn(160, device=dev, requires_grad=True)
    with torch.no_grad():
        y0 = ops.dense(x, w, b, act=True)
    y = ops.dense(x, w, b, act=True)
    assert torch.equal(y, y0)
    g = torch.randn_like(y)
    y.backward(g)
    assert x.grad is None
    dw, db = w.grad.clone(), b.grad.clone()
    w.grad = b.grad = None
    ops.dense(x, w, b, act=True).backward(g)
    assert torch.equal(dw, w.grad) and torch.equal(db, b.grad)
    # a transposed view / a column slice are made contiguous on the way in
    xt = torch.randn(144, 300, device=dev).t()
    assert maxnorm_err(ops.dense(xt, w.detach()).cpu(), (xt.double().cpu() @ w.detach().double().cpu().t())) < 2e-6


def test_dense_cabi_argument_errors(dev):
    from pamnet_amd import lib
    x = torch.zeros(4, 8, device=dev)
    w = torch.zeros(3, 8, device=dev)
    y = torch.zeros(4, 3, device=dev)
    st = lib.stream_of(x)
    handle = lib.load()
    f = handle.pamnet_dense_fwd_f32
    assert f(lib.ptr(x), 8, lib.ptr(w), 8, None, 4, 8, 3, 0, None, None, st) == -2          # Y missing
    assert f(lib.ptr(x), 4, lib.ptr(w), 8, None, 4, 8, 3, 0, None, lib.ptr(y), st) == -1    # ldx < k
    assert f(lib.ptr(x), 8, lib.ptr(w), 8, None, 4, 0, 3, 0, None, lib.ptr(y), st) == -1    # k < 1
    assert f(None, 8, None, 8, None, 0, 8, 3, 0, None, None, st) == 0                        # no rows: nothing to do
    bwd = handle.pamnet_dense_bwd_f32
    dw = torch.zeros(3, 8, device=dev)
    assert bwd(lib.ptr(y), None, lib.ptr(x), 8, lib.ptr(w), 8, 4, 8, 3, 1, None, lib.ptr(dw), None, None, st) == -2   # act, no Z
    assert bwd(lib.ptr(y), None, lib.ptr(x), 8, lib.ptr(w), 8, 4, 8, 3, 0, None, lib.ptr(dw), None, None, st) == -2   # dW, no scratch
    need = ctypes.c_int64(0)
    assert handle.pamnet_dense_scratch_floats(4, 8, 3, ctypes.addressof(need)) == 0 and need.value >= 3 * 8 + 3
    assert handle.pamnet_dense_scratch_floats(4, 8, 3, None) == -1


@pytest.mark.parametrize('dataset,dim,small', [('QM9', 256, False), ('QM9', 132, False), ('QM9', 160, True),
                                               ('PDBbind', 192, False), ('rna_native', 144, False)])
def test_wide_models_vs_oracle(dev, dataset, dim, small):
    """Hidden sizes above 128: every Linear of the model is a launch of csrc/dense.hip (no library GEMM on the way:
    torch's matmul entry points are made to raise for the duration); outputs and every parameter gradient against the
    oracle's fp64 run."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    if dataset == 'PDBbind':
        cfg = models.Config(dataset='PDBbind', dim=dim, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)
        b = synth.pdbbind_batch(3, 0, 2, n_pocket=60, n_ligand=12)
    elif dataset.startswith('rna'):
        cfg = models.Config(dataset=dataset, dim=dim, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
        b = synth.rna_batch(5, 0, 2, n_nodes=150)
    else:
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(8, 0, 6)
    fwd = O.pamnet_s_forward if small else O.pamnet_forward
    sd = O.init_state_dict(cfg, seed=5, small=small)
    model = (models.PAMNet_s if small else models.PAMNet)(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    assert model.dim == dim
    data = b.to(dev)

    def refuse(*a, **k):
        raise AssertionError('a library GEMM was called on the product path')
    saved = {}
    names = [(torch.nn.functional, 'linear'), (torch, 'mm'), (torch, 'matmul'), (torch, 'addmm'), (torch, 'bmm'),
             (torch.Tensor, '__matmul__'), (torch.Tensor, 'matmul'), (torch.Tensor, 'mm')]
    for owner, nm in names:
        saved[(owner, nm)] = getattr(owner, nm)
        setattr(owner, nm, refuse)
    try:
        out = model(data)
        torch.nn.functional.l1_loss(out, data.y).backward()
    finally:
        for (owner, nm), fn in saved.items():
            setattr(owner, nm, fn)
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    x64 = b.x if dataset == 'QM9' else b.x.double()
    inter = {}
    ref = fwd(p64, cfg, x64, b.batch, pos, ei, dtype=torch.float64, intermediates=inter)
    torch.nn.functional.l1_loss(ref, b.y.double()).backward()
    scale = float(ref.detach().abs().max())
    if dataset == 'PDBbind':
        scale = max(float(inter['pool_in'].detach().abs()[b.batch == k].sum()) for k in range(2))
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) / scale < TOL
    _check_gradients(model, p64, fwd, sd, cfg, b,
                     head_bias_terms=(b.x.size(0) / (2 * 2.0 * cfg.n_layer)) if dataset == 'PDBbind' else None)


def test_non_default_basis_embedding_runs_on_the_dense_kernels(dev):
    """models.py:187-188 with num_spherical * num_radial != 42: the spherical-basis embedding has no embedding kernel of
    its own and runs csrc/dense.hip (ops.dense_act), two weight sets selected per row in the small model."""
    from pamnet_amd import ops
    torch.manual_seed(3)
    x = torch.randn(777, 30, device=dev)
    la, lb = torch.nn.Linear(30, 128).to(dev), torch.nn.Linear(30, 128).to(dev)
    kind = (torch.arange(777, device=dev) % 3 == 0).to(torch.int32)
    y = ops.dense_act(x, la, lb, kind)
    g = torch.randn_like(y)
    y.backward(g)
    got = [y.detach()] + [p.grad.clone() for p in (la.weight, la.bias, lb.weight, lb.bias)]
    for p in (la.weight, la.bias, lb.weight, lb.bias):
        p.grad = None
    x64 = x.double()
    za = torch.nn.functional.linear(x64, la.weight.double(), la.bias.double())
    zb = torch.nn.functional.linear(x64, lb.weight.double(), lb.bias.double())
    y64 = torch.nn.functional.silu(torch.where((kind == 0).unsqueeze(1), za, zb))
    y64.backward(g.double())
    ref = [y64.detach()] + [p.grad.double() for p in (la.weight, la.bias, lb.weight, lb.bias)]
    for a, r in zip(got, ref):
        assert maxnorm_err(a.cpu().double(), r.cpu()) < 2e-6


def test_trainer_on_a_wide_model_matches_torch_adam(dev):
    """Trainer (flat buffers, fused clip + Adam + EMA) over a dim = 136 model -- the layer-by-layer path under plain autograd,
    gradients accumulated into the trainer's preallocated views -- against torch.optim.Adam + clip_grad_norm_ on a twin."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth, train
    cfg = models.Config(dataset='QM9', dim=136, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)
    sd = O.init_state_dict(cfg, seed=7)
    model, twin = models.PAMNet(cfg), models.PAMNet(cfg)
    model.load_state_dict(sd, strict=True), twin.load_state_dict(sd, strict=True)
    model, twin = model.to(dev), twin.to(dev)
    data = synth.qm9_batch(4, 0, 6).to(dev)
    tr = train.Trainer(model, lr=1e-3)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    for _ in range(4):
        tr.step(data)
        opt.zero_grad()
        torch.nn.functional.l1_loss(twin(data), data.y).backward()
        torch.nn.utils.clip_grad_norm_(twin.parameters(), 1000.0)
        opt.step()
    tr.drain()
    a, c = model.state_dict(), twin.state_dict()
    for k in a:
        assert maxnorm_err(a[k].cpu().numpy(), c[k].cpu().numpy()) < 2e-4, k      # (Adam amplifies rounding: g / |g|)


@pytest.mark.parametrize('name', ['wide_qm9_d192_l2', 'wide_qm9s_d136_l2', 'wide_pdbbind_d160_l2', 'wide_rna_d144_l1'])
def test_wide_models_vs_reference_runs(dev, golden, name):
    """Hidden sizes above 128 against runs of the REFERENCE ITSELF (tests/golden/gen/gen_golden.py --wide-only; fp32 and fp64
    runs of models.py at dim 192 / 136 / 160 / 144): graph outputs and pooled node values under the bound of
    test_hip_model.py (never tighter than the reference's own fp32 run), the integer sizes of its graphs exactly, and the
    backward of the mean L1 loss against the reference's fp64 autograd -- loss, global gradient norm, every parameter
    gradient's L2 norm and the stored full gradient tensors."""
    import numpy as np
    import models
    from oracle import pamnet_oracle as O
    from test_hip_model import CANCEL_TOL, GRAD_TOL, grad_tol, _cfg_from, _ok
    from test_oracle_golden import _wide_batch
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    small = 'qm9s' in name
    b = _wide_batch(name)
    assert b.x.size(0) == int(g['num_nodes']) and abs(float(b.x.double().abs().sum()) - float(g['x_checksum'])) < 1e-6
    sd = O.init_state_dict(cfg, seed=int(g['seed']), small=small)
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(g['weights_checksum'])) < 1e-6
    model = (models.PAMNet_s if small else models.PAMNet)(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    assert model.dim == cfg.dim > 128
    data = b.to(dev)
    out_t = model(data)
    loss = torch.nn.functional.l1_loss(out_t, data.y)
    loss.backward()
    out, node_out = out_t.detach().cpu().numpy(), model._node_out.cpu().numpy()
    ok, info_n = _ok(node_out, g['node_out32'], g['node_out64'])
    assert ok, ('node_out', info_n)
    gc = model._graph_cache
    assert gc.loc.m == int(g['num_edges_l']) and gc.n_trip == int(g['num_triplets']) and gc.n_pair == int(g['num_pairs'])
    if cfg.dataset == 'PDBbind':
        scale = max(float(np.abs(g['node_out64'][b.batch.numpy() == k]).sum()) for k in range(len(g['out64'])))
        ok, info = _ok(out, g['out32'], g['out64'], scale)
    else:
        ok, info = _ok(out, g['out32'], g['out64'])
    assert ok, ('out', info)
    assert abs(float(loss) - float(g['loss64'])) < 2e-5 * max(1.0, abs(float(g['loss64'])))
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    gn = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
    assert abs(gn / float(g['grad_norm64']) - 1) < 1e-4, (gn, float(g['grad_norm64']))
    worst = 0.0
    for k, l2 in zip(g['grad_keys'].tolist(), g['grad_l2_64']):
        scale = float(l2)
        if k.endswith('W_out.bias') and cfg.dataset == 'PDBbind':
            # d loss / d b_out = the sum of the signed pooling weights over the nodes (complex - pocket - ligand): almost pure
            # cancellation, so it is judged on the scale of its Linear's weight gradient (test_hip_model._check_gradients)
            scale = max(scale, float(grads[k[:-4] + 'weight'].abs().max()))
        e = abs(float(grads[k].double().norm()) - float(l2)) / max(scale, 1e-300)
        assert e < (CANCEL_TOL if (k.endswith('W_out.bias') and cfg.dataset == 'PDBbind') else grad_tol(cfg.dataset)), (k, e)
        worst = max(worst, e)
    for k in g.files:
        if k.startswith('grad64/'):
            assert maxnorm_err(grads[k[7:]].cpu().numpy(), g[k]) < grad_tol(cfg.dataset), k
    print('%s vs the reference: out %.2e (ref fp32 %.2e), node_out %.2e; |grad| rel %.1e, worst per-tensor L2 %.1e'
          % (name, info[0], info[1], info_n[0], abs(gn / float(g['grad_norm64']) - 1), worst))

"""End-to-end parity of the MI355X path (models.PAMNet / PAMNet_s through libpamnet_hip.so) against

  * golden vectors produced by the reference's own code (tests/golden/*.npz: fp32 and fp64 reference runs), and
  * the CPU oracle on fresh seeded inputs at sizes it finishes in seconds,

plus size-independent properties at the BASELINE configuration (B=128, d=128, L=6).

Parity protocol: per tensor, err(a, b) = max|a-b| / max|b|.  The HIP path must satisfy
    err(hip, ref_fp64) <= max(1e-5, 2 * err(ref_fp32, ref_fp64))
i.e. the north star's 1e-5, never tighter than the reference's own fp32 noise.
"""
import os
import numpy as np
import pytest
import torch

from conftest import maxnorm_err

TOL = 1e-5


def _cfg_from(g, Config):
    cfg = Config(dataset=str(g['cfg_dataset']), dim=int(g['cfg_dim']), n_layer=int(g['cfg_n_layer']),
                 cutoff_l=float(g['cfg_cutoff_l']), cutoff_g=float(g['cfg_cutoff_g']), flow=str(g['cfg_flow']))
    if 'cfg_basis' in g.files:             # PAMNet(config, num_spherical, num_radial, envelope_exponent), models.py:22
        cfg.basis = tuple(int(v) for v in g['cfg_basis'])
    return cfg


def _basis_args(cfg):
    return tuple(getattr(cfg, 'basis', ()))


def _batch_from(g, dev):
    from pamnet_amd.synth import Batch
    kw = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('in/')}
    kw['num_graphs'] = int(kw['batch'].max()) + 1
    return Batch(**kw).to(dev)


def _ok(a, ref32, ref64, scale=None):
    if scale is not None:
        e = float(np.max(np.abs(np.asarray(a, np.float64) - ref64))) / scale
        floor = float(np.max(np.abs(ref32.astype(np.float64) - ref64))) / scale
    else:
        e, floor = maxnorm_err(a, ref64), maxnorm_err(ref32, ref64)
    return e <= max(TOL, 2 * floor), (e, floor)


# (one exception: the head bias of the signed PDBbind pooling -- a scalar that is a sum over all nodes of +-1-weighted terms
# cancelling ~1000x; its error moves 3x with the summation order.  Measured worst 3.6e-5 on the weight gradient's scale (a wide
# model); it keeps the round-5 bound.)
CANCEL_TOL = 1e-4
# PDBbind: complex - pocket - ligand pooling cancels ~1000x, every gradient is a difference of large terms: the reference's own
# fp32 backward sits at 6e-6 .. 9e-6 of its fp64 one there and the HIP path at 1.4e-5 .. 1.8e-5 (measured round 6) -- judged at
# 3e-5 (and never tighter than twice the fp32 oracle's own error).  Everything else: 1e-5 (measured worst 4e-7 on QM9).
PDBBIND_GRAD_TOL = 3e-5


def grad_tol(dataset):
    return PDBBIND_GRAD_TOL if str(dataset) == 'PDBbind' else GRAD_TOL


GRAD_TOL = 1e-5          # gradients within 1e-5 of the reference's fp64 autograd (measured worst: 4e-7) ...


def _check_gradients(model, p64, fwd, sd, cfg, b, report=None, head_bias_terms=None):
    """Every parameter gradient of the HIP backward against the oracle's fp64 autograd (p64[k].grad already filled):
        err(hip, fp64) <= max(GRAD_TOL, 2 * err(oracle_fp32, fp64))
    -- the 1e-5 of DESIGN.md (round 6: tightened from 1e-4), never tighter than what the reference's OWN fp32 backward achieves on the same inputs
    (its sympy closed forms and the signed PDBbind pooling cancel catastrophically for a few tensors; the fp32 oracle
    is run here to measure exactly that floor instead of hard-coding a looser bound)."""
    p32 = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    torch.nn.functional.l1_loss(fwd(p32, cfg, b.x, b.batch, pos, ei), b.y).backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    gn64 = float(torch.sqrt(sum((p.grad ** 2).sum() for p in p64.values() if p.grad is not None)))
    gn32 = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in p32.values() if p.grad is not None)))
    assert abs(gn / gn64 - 1) <= max(GRAD_TOL, 2 * abs(gn32 / gn64 - 1)), (gn, gn64, gn32)
    worst = (0.0, 0.0, None)
    for k, p in model.named_parameters():
        if p64[k].grad is None:
            continue
        e = maxnorm_err(p.grad.cpu().numpy(), p64[k].grad.numpy())
        floor = maxnorm_err(p32[k].grad.numpy(), p64[k].grad.numpy())
        tol = grad_tol(cfg.dataset)
        if p.numel() == 1 and k.endswith('W_out.bias'):
            tol = CANCEL_TOL
            # d loss / d b = sum over nodes of the signed pooling weights: a scalar that is almost pure cancellation
            # (PDBbind: complex - pocket - ligand), so its own magnitude says nothing about the size of the terms and its
            # relative error moves by 3x with the summation order.  Judged on the scale of its Linear's weight gradient
            # (the same per-node terms, weighted by the node features instead of by 1).
            wk = k[:-4] + 'weight'
            scale = max(abs(float(p64[k].grad)), float(p64[wk].grad.abs().max()))
            if head_bias_terms is not None:
                # ... and never tighter than fp32 accumulation itself: the scalar is a sum of `head_bias_terms` worth of
                # magnitude (sum over nodes of |d node_out| * softmax weight), allowed 1e-7 of it (~2 ulp) -- a model whose
                # weight gradients happen to be small does not make this cancellation sum any more exact
                scale = max(scale, 1e-3 * head_bias_terms)
            e = abs(float(p.grad) - float(p64[k].grad)) / scale
            floor = abs(float(p32[k].grad) - float(p64[k].grad)) / scale
        assert e <= max(tol, 2 * floor), (k, e, floor)
        if e > worst[0]:
            worst = (e, floor, k)
    if report is not None:
        report['grad_worst'] = worst
        report['grad_norm_rel'] = abs(gn / gn64 - 1)
    return worst


# ------------------------------------------------------------------------------------------------------ CPU (not gpu)
def test_state_dict_layout_matches_reference(golden):
    """Keys and shapes == the reference's (oracle.init_state_dict mirrors them; the RNA checkpoint is the real thing)."""
    import models
    from oracle import pamnet_oracle as O
    for cls, small, ds in ((models.PAMNet, False, 'QM9'), (models.PAMNet_s, True, 'QM9'), (models.PAMNet, False, 'PDBbind')):
        cfg = models.Config(dataset=ds, dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        sd = cls(cfg).state_dict()
        ref = O.init_state_dict(cfg, seed=0, small=small)
        assert set(sd.keys()) == set(ref.keys())
        assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    g = golden('rna_native')
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    m = models.PAMNet(cfg)
    ck = {k: torch.from_numpy(g['ckpt/' + k]) for k in g['ckpt_keys'].tolist()}
    m.load_state_dict(ck, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 11714
    big = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0))
    assert sum(p.numel() for p in big.parameters() if p.requires_grad) == 3581100


@pytest.mark.parametrize('dataset,dim,small', [('QM9', 96, False), ('QM9', 20, True), ('PDBbind', 48, False), ('rna_x', 10, False),
                                                ('QM9', 100, False), ('QM9', 4, False), ('QM9', 130, False), ('QM9', 141, True)])
def test_any_dim_keeps_the_reference_state_dict_layout(dataset, dim, small):
    """The reference accepts any `dim` (models.py:25).  Widths without a kernel family of their own run zero-padded at the next
    engine width (models._PAMNetBase): `state_dict()` / `load_state_dict()` still speak the reference's keys and shapes,
    a round trip is exact, and everything outside the logical blocks is zero."""
    import models
    from oracle import pamnet_oracle as O
    cfg = models.Config(dataset=dataset, dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    m = (models.PAMNet_s if small else models.PAMNet)(cfg)
    assert m.config_dim == dim and m.dim == models.engine_width(dim)
    assert m.dim in models.ENGINE_WIDTHS if dim <= 128 else (m.dim % 4 == 0 and 0 < m.dim - dim < 4)
    ref = O.init_state_dict(cfg, seed=1, small=small)
    sd = m.state_dict()
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    # the initial values follow the initialisation law on the LOGICAL shapes (an unpadded twin): a Linear(dim, dim)'s
    # default bound is 1 / sqrt(dim), not 1 / sqrt(engine width)
    w = sd['global_layer.0.mlp_x1.0.0.weight']
    assert float(w.abs().max()) <= 1.0 / dim ** 0.5 + 1e-6 and float(w.abs().max()) > 0.8 / dim ** 0.5
    m.load_state_dict(ref, strict=True)
    back = m.state_dict()
    assert all(torch.equal(back[k], ref[k]) for k in ref)
    for k, p in m.named_parameters():
        assert float(p.detach()[~m.logical_mask(k)].abs().sum()) == 0.0, k
        assert torch.equal(p.detach()[m.logical_mask(k)].reshape(ref[k].shape), ref[k])
    with pytest.raises(RuntimeError):                       # a tensor of some other shape is still refused
        bad = dict(ref)
        bad['embeddings'] = torch.zeros(ref['embeddings'].size(0), dim + 1)
        m.load_state_dict(bad, strict=True)
    # above 128: the next multiple of 4 (the layer-by-layer path on csrc/dense.hip), exact multiples run unpadded
    assert models.engine_width(128) == 128 and models.engine_width(129) == 132 and models.engine_width(192) == 192


def test_padded_models_draw_the_unpadded_models_random_stream():
    """Seed for seed a model whose dim has no kernel width of its own starts from the weights the reference's shapes would get,
    and leaves the generator where that construction leaves it (the twin takes the first draw)."""
    import models
    for dim, small in ((100, False), (130, False), (24, True)):
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)
        cls = models.PAMNet_s if small else models.PAMNet
        torch.manual_seed(3)
        a, ra = cls(cfg), torch.rand(1)
        torch.manual_seed(3)
        b, rb = cls(cfg, _pad=False), torch.rand(1)
        sa, sb = a.state_dict(), b.state_dict()
        assert a.dim != dim and b.dim == dim and sa.keys() == sb.keys()
        assert all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(ra, rb)


def test_parameter_walk_cache_follows_the_live_module_tree():
    """`parameters()` / `named_parameters()` are cached walks (the reference loop walks the tree three times per step); a
    module or parameter replaced or added INSIDE a child must show up (a replaced module still holds its old
    tensors, so the cache is validated against the live tree, not against its old owners)."""
    import models
    m = models.PAMNet(models.Config(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0))
    first = list(m.parameters())
    assert [id(p) for p in m.parameters()] == [id(p) for p in first]            # (served from the cache)
    old = m.global_layer[0].W_out
    m.global_layer[0].W_out = torch.nn.Linear(32, 1)
    named = dict(m.named_parameters())
    assert named['global_layer.0.W_out.weight'] is m.global_layer[0].W_out.weight
    assert all(p is not old.weight and p is not old.bias for p in m.parameters())
    assert set(named) == set(k for k, _ in torch.nn.Module.named_parameters(m))
    m.local_layer[1].register_parameter('extra', torch.nn.Parameter(torch.zeros(3)))
    assert 'local_layer.1.extra' in dict(m.named_parameters())
    assert any(p is m.local_layer[1].extra for p in m._all_params())
    m.local_layer[1].mlp_x1[0][0].weight = torch.nn.Parameter(torch.ones(32, 32))   # a parameter swapped in place
    assert dict(m.named_parameters())['local_layer.1.mlp_x1.0.0.weight'] is m.local_layer[1].mlp_x1[0][0].weight
    assert not any(n.startswith(('global_layer.', 'local_layer.')) for n in
                   [k for k, p in m.named_parameters() if any(p is q for q in m._top_params())])


def test_parameter_walk_is_checked_once_per_forward_and_again_between_forwards():
    """forward() validates the cached parameter walk against the live tree once (its dtype check); the other users of the cached
    lists inside the same forward take that result, and the flag is dropped when the forward leaves -- normally or by an
    exception -- so a parameter swapped between two forwards is still seen."""
    import models
    m = models.PAMNet(models.Config(dataset='QM9', dim=16, n_layer=1, cutoff_l=5.0, cutoff_g=5.0))
    list(m.parameters())
    walks = []
    real = m._param_cache_valid
    m.__dict__['_param_cache_valid'] = lambda cache: (walks.append(1), real(cache))[1]
    seen = {}

    def inside(data):
        seen['a'], seen['b'] = m._all_params(), m._top_params()
        seen['flag'] = m.__dict__['_params_checked']
        if data == 'raise':
            raise KeyError('inside the forward')
        return 7
    m.__dict__['_forward'] = inside
    m._check_dtype()                                     # (what forward() does first: one walk)
    n0 = len(walks)
    assert n0 >= 1 and m._checked_forward(None) == 7
    assert len(walks) == n0 and seen['flag'] is True and m.__dict__['_params_checked'] is False
    with pytest.raises(KeyError):
        m._checked_forward('raise')
    assert m.__dict__['_params_checked'] is False
    m.global_layer[0].W_out = torch.nn.Linear(16, 1)     # between forwards: the next list is walked again and is current
    assert any(p is m.global_layer[0].W_out.weight for p in m._all_params()) and len(walks) > n0


def test_cast_models_are_refused():
    """The kernels read parameters as fp32 through raw pointers: a model cast to another floating type raises instead of
    being read with the wrong element size (the reference runs in whatever dtype it is cast to; this path is fp32 by
    construction)."""
    import models

    class D:
        x = torch.zeros(3)
        batch = torch.zeros(3, dtype=torch.long)
        pos = torch.zeros(3, 3)
        edge_index = torch.zeros(2, 0, dtype=torch.long)
    for cls in (models.PAMNet, models.PAMNet_s):
        m = cls(models.Config(dataset='QM9', dim=16, n_layer=1, cutoff_l=5.0, cutoff_g=5.0))
        for cast in ('double', 'half', 'bfloat16'):
            with pytest.raises(TypeError, match='float32'):
                getattr(m, cast)()(D())
        m.float()
        with pytest.raises(RuntimeError):               # fp32 again: gets as far as "kernels run on an MI355X only" (CPU tensors)
            m(D())


def test_invalid_dataset_raises():
    import models
    cfg = models.Config(dataset='nope', dim=16, n_layer=1, cutoff_l=2.0, cutoff_g=5.0)
    m = models.PAMNet(cfg)

    class D:
        x = torch.zeros(3)
        batch = torch.zeros(3, dtype=torch.long)
    with pytest.raises(ValueError):
        m(D())
    with pytest.raises(ValueError):
        models.PAMNet_s(models.Config(dataset='PDBbind', dim=16, n_layer=1, cutoff_l=2.0, cutoff_g=5.0))(D())


# ------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs an MI355X'
    return torch.device('cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('name,small', [('qm9_d32_l2', False), ('qm9s_d32_l2', True), ('pdbbind_d32_l2', False),
                                        ('qm9_ragged_d32_l2', False), ('qm9s_ragged_d32_l2', True),
                                        ('qm9_flow_t2s_d32_l2', False), ('pdbbind_flow_t2s_d32_l2', False),
                                        ('pdbbind_d128_l3', False), ('qm9s_d128_l2', True),
                                        ('qm9_d128_l6', False), ('qm9_basis_5x4_p6_d32_l2', False),
                                        ('qm9_basis_8x7_p4_d128_l2', False)])
def test_forward_vs_reference_golden(dev, golden, name, small):
    import models
    from oracle import pamnet_oracle as O
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    model = (models.PAMNet_s if small else models.PAMNet)(cfg, *_basis_args(cfg))
    model.load_state_dict(O.init_state_dict(cfg, seed=int(g['seed']), small=small), strict=True)
    model = model.to(dev)
    data = _batch_from(g, dev)
    with torch.no_grad():
        out = model(data)
    pool_in = model._node_out.cpu().numpy()
    ok, info = _ok(pool_in, g['node_out32'], g['node_out64'])
    assert ok, ('node_out', info)
    if 'x_layers64' in g.files:
        xl = torch.stack(list(model._x_layers)).cpu().numpy()
        ok, info = _ok(xl, g['x_layers32'], g['x_layers64'])
        assert ok, ('x_layers', info)
    scale = None
    if cfg.dataset == 'PDBbind':       # complex - pocket - ligand cancellation: normalise by the summed magnitude
        scale = max(float(np.abs(g['node_out64'][g['in/batch'] == b]).sum()) for b in range(len(g['out64'])))
    ok, info = _ok(out.cpu().numpy(), g['out32'], g['out64'], scale)
    assert ok, ('out', info)
    if scale is not None:
        # the raw figure max|d| / max|out| beside the magnitude-normalised one: held to the reference's own fp32 noise
        raw, raw_floor = maxnorm_err(out.cpu().numpy(), g['out64']), maxnorm_err(g['out32'], g['out64'])
        print('PDBbind %s: err/sum|node_out| = %.2e (ref fp32 %.2e);  raw max|d|/max|out| = %.2e (ref fp32 %.2e)'
              % (name, info[0], info[1], raw, raw_floor))
        assert raw <= max(TOL, 2 * raw_floor), (raw, raw_floor)
    # graph sizes are integers: exact
    assert model._graph_cache.loc.m == int(g['num_edges_l'])
    assert model._graph_cache.n_pair == int(g['num_pairs'])
    if not small:
        assert model._graph_cache.n_trip == int(g['num_triplets'])
    # deterministic: bitwise identical on a second run
    with torch.no_grad():
        assert torch.equal(out, model(data))


@pytest.mark.gpu
def test_rna_checkpoint_end_to_end(dev, golden):
    """Shipped checkpoint + shipped RNA-Puzzles graphs: kNN graph, both cutoffs, flow=target_to_source, mean pool."""
    import models
    from pamnet_amd.synth import Batch
    g = golden('rna_native')
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    model = models.PAMNet(cfg)
    model.load_state_dict({k: torch.from_numpy(g['ckpt/' + k]) for k in g['ckpt_keys'].tolist()}, strict=True)
    model = model.to(dev).eval()
    for gid in (6, 4, 17):
        x = torch.from_numpy(g['g%d/x' % gid]).to(dev)
        data = Batch(x=x, batch=torch.zeros(x.size(0), dtype=torch.long, device=dev), num_graphs=1)
        with torch.no_grad():
            out = model(data)
        ok, info = _ok(out.cpu().numpy(), g['g%d/out32' % gid], g['g%d/out64' % gid])
        assert ok, (gid, info)
        gc = model._graph_cache
        assert gc.loc.m == int(g['g%d/num_edges_l' % gid]) and gc.n_trip == int(g['g%d/num_triplets' % gid])
        assert gc.n_pair == int(g['g%d/num_pairs' % gid])
        if gid == 6:
            ok, info = _ok(torch.stack(list(model._x_layers)).cpu().numpy(), g['g6/x_layers32'], g['g6/x_layers64'])
            assert ok, info
            ok, info = _ok(model._node_out.cpu().numpy(), g['g6/node_out32'], g['g6/node_out64'])
            assert ok, info
    # two graphs batched == each alone
    x = torch.cat([torch.from_numpy(g['g4/x']), torch.from_numpy(g['g6/x'])]).to(dev)
    batch = torch.cat([torch.zeros(g['g4/x'].shape[0], dtype=torch.long), torch.ones(g['g6/x'].shape[0], dtype=torch.long)])
    with torch.no_grad():
        out = model(Batch(x=x, batch=batch.to(dev), num_graphs=2))
    assert maxnorm_err(out.cpu().numpy(), g['batched_4_6_out32']) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['qm9_d32_l2', 'pdbbind_d32_l2', 'pdbbind_d128_l3', 'qm9_basis_5x4_p6_d32_l2',
                                  'qm9_basis_8x7_p4_d128_l2', 'qm9_flow_t2s_d32_l2', 'pdbbind_flow_t2s_d32_l2'])
def test_gradients_vs_reference_golden(dev, golden, name):
    """d L1-loss / d params through the HIP backward kernels vs the reference's fp64 autograd."""
    import models
    from oracle import pamnet_oracle as O
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    model = models.PAMNet(cfg, *_basis_args(cfg))
    model.load_state_dict(O.init_state_dict(cfg, seed=int(g['seed'])), strict=True)
    model = model.to(dev)
    data = _batch_from(g, dev)
    out = model(data)
    loss = torch.nn.functional.l1_loss(out, data.y)
    loss.backward()
    assert abs(loss.item() - float(g['loss64'])) < 2e-5 * max(1.0, abs(float(g['loss64'])))
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(gn / float(g['grad_norm64']) - 1) < 1e-4
    sd = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('grad64/'):
            assert maxnorm_err(sd[k[7:]].grad.cpu().numpy(), g[k]) < 1e-4, k       # fp32 backward vs fp64 reference


@pytest.mark.gpu
def test_fresh_inputs_vs_oracle_rna_flow_variants(dev):
    """Seeded synthetic RNA-schema graphs, both flow conventions (asymmetric kNN graph -> the flow matters)."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    b = synth.rna_batch(3, 0, 2, n_nodes=260)
    for flow in ('target_to_source', 'source_to_target'):
        cfg = models.Config(dataset='rna_x', dim=16, n_layer=2, cutoff_l=2.6, cutoff_g=20.0, flow=flow)
        sd = O.init_state_dict(cfg, seed=21)
        model = models.PAMNet(cfg)
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out = model.to(dev)(b.to(dev))
        ref64 = O.pamnet_forward({k: v.double() for k, v in sd.items()}, cfg, b.x.double(), b.batch, dtype=torch.float64)
        ref32 = O.pamnet_forward(sd, cfg, b.x, b.batch)
        ok, info = _ok(out.cpu().numpy(), ref32.numpy(), ref64.numpy())
        assert ok, (flow, info)


@pytest.mark.gpu
def test_baseline_config_properties(dev):
    """BASELINE configs[1] (QM9, d=128, L=6, B=128): properties that need no oracle at full size."""
    import models
    from pamnet_amd import synth
    torch.manual_seed(0)
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    model = models.PAMNet(cfg).to(dev)
    mols = [synth.qm9_molecule(0, i) for i in range(128)]
    with torch.no_grad():
        full = model(synth.collate(mols).to(dev))
        assert full.shape == (128,) and torch.isfinite(full).all()
        # molecules are independent units: any sub-batch / permutation reproduces the same per-molecule outputs
        perm = np.random.default_rng(0).permutation(128)
        shuf = model(synth.collate([mols[i] for i in perm]).to(dev))
        assert maxnorm_err(shuf.cpu().numpy(), full.cpu().numpy()[perm]) < 2e-6
        sub = model(synth.collate(mols[40:50]).to(dev))
        assert maxnorm_err(sub.cpu().numpy(), full.cpu().numpy()[40:50]) < 2e-6
        # rigid motion invariance (distances / angles only)
        rot = torch.linalg.qr(torch.randn(3, 3))[0]
        moved = synth.collate(mols)
        moved.pos = moved.pos @ rot + torch.tensor([3.0, -2.0, 0.5])
        assert maxnorm_err(model(moved.to(dev)).cpu().numpy(), full.cpu().numpy()) < 1e-4


@pytest.mark.gpu
def test_pdbbind_baseline_config_properties(dev):
    """BASELINE configs[3] (PDBbind schema, dim=128, n_layer=3, 32 complexes, ~19 k nodes, ~0.7 M global edges) at full
    size: complexes are independent units (outputs and, through the loss, parameter gradients of the batch equal those
    accumulated over its four 8-complex shards) and the step is run-to-run bitwise deterministic.  The pooled output
    is complex - pocket - ligand (a ~1000x cancellation), so it is compared relative to the summed magnitude."""
    import models
    from pamnet_amd import synth
    torch.manual_seed(0)
    cfg = models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    model = models.PAMNet(cfg).to(dev)
    graphs = [synth.pdbbind_complex(1, i) for i in range(32)]

    def run(gs, scale):
        b = synth.collate(gs).to(dev)
        out = model(b)
        (torch.nn.functional.l1_loss(out, b.y, reduction='sum') * scale).backward()
        return out.detach()

    model.zero_grad()
    full = run(graphs, 1.0 / 32)
    batch_ids = synth.collate(graphs).batch.to(dev)
    mag = float(torch.zeros(32, device=dev).index_add_(0, batch_ids, model._node_out.detach().abs()).max())
    g_full = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    model.zero_grad()
    again = run(graphs, 1.0 / 32)
    assert torch.equal(full, again)
    assert all(torch.equal(a, p.grad) for a, p in zip(g_full, [p for p in model.parameters() if p.grad is not None]))
    model.zero_grad()
    parts = torch.cat([run(graphs[8 * i:8 * i + 8], 1.0 / 32) for i in range(4)])
    assert float((parts - full).abs().max()) < 2e-6 * mag      # relative to the summed per-complex magnitude
    for a, p in zip(g_full, [p for p in model.parameters() if p.grad is not None]):
        if a.numel() > 1:                                  # scalar W_out.bias: signed-sum cancellation (see above)
            assert maxnorm_err(p.grad.cpu().numpy(), a.cpu().numpy()) < 5e-5


@pytest.mark.gpu
def test_rna_baseline_config_properties(dev):
    """BASELINE configs[4] (RNA-Puzzles schema, dim=16, n_layer=1, 8 graphs of 800-3900 nodes, kNN global graph) at
    full size: graphs are independent units (a graph's score does not depend on what it is batched with, in any order),
    scores are invariant to rigid motion, and the whole forward is run-to-run bitwise deterministic."""
    import models
    from pamnet_amd import synth
    torch.manual_seed(0)
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    model = models.PAMNet(cfg).to(dev).eval()
    graphs = [synth.rna_chain(2, i) for i in range(8)]
    with torch.no_grad():
        full = model(synth.collate(graphs).to(dev))
        assert full.shape == (8,) and torch.isfinite(full).all()
        assert torch.equal(full, model(synth.collate(graphs).to(dev)))
        perm = [5, 2, 7, 0, 3, 6, 1, 4]
        shuf = model(synth.collate([graphs[i] for i in perm]).to(dev))
        assert maxnorm_err(shuf.cpu().numpy(), full.cpu().numpy()[perm]) < 2e-6
        for i in (1, 4):                                   # the smallest graphs of the batch, alone
            one = model(synth.collate([graphs[i]]).to(dev))
            assert maxnorm_err(one.cpu().numpy(), full.cpu().numpy()[i:i + 1]) < 2e-6
        rot = torch.linalg.qr(torch.randn(3, 3))[0]
        moved = synth.collate(graphs)
        moved.x = torch.cat([moved.x[:, :3] @ rot + torch.tensor([11.0, -4.0, 2.5]), moved.x[:, 3:]], 1)
        assert maxnorm_err(model(moved.to(dev)).cpu().numpy(), full.cpu().numpy()) < 2e-4


def _oracle_threads():
    """The oracle's small CPU ops oversubscribe badly on the GPU box's 256-core host: a bounded pool."""
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.mark.gpu
def test_baseline_qm9_b128_vs_oracle(dev):
    """BASELINE configs[1] EXACTLY (QM9 schema, dim=128, n_layer=6, B=128 -- the batch bench.py times): graph outputs,
    node features after every layer and every parameter gradient of the L1 loss against the CPU oracle (fp32 + fp64)."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    _oracle_threads()
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    b = synth.qm9_batch(0, 0, 128)
    sd = O.init_state_dict(cfg, seed=0)
    model = models.PAMNet(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    data = b.to(dev)
    out = model(data)
    torch.nn.functional.l1_loss(out, data.y).backward()
    ref32 = O.pamnet_forward(sd, cfg, b.x, b.batch, b.pos, b.edge_index)
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    inter = {}
    ref64 = O.pamnet_forward(p64, cfg, b.x, b.batch, b.pos, b.edge_index, dtype=torch.float64, intermediates=inter)
    torch.nn.functional.l1_loss(ref64, b.y.double()).backward()
    gc = model._graph_cache
    assert gc.n == b.x.numel() and gc.glob.m == inter['edge_index_g'].size(1) and gc.loc.m == inter['edge_index_l'].size(1)
    assert gc.n_trip == inter['idx_kj'].numel() and gc.n_pair == inter['idx_jj_pair'].numel()
    ok, info = _ok(out.detach().cpu().numpy(), ref32.numpy(), ref64.detach().numpy())
    assert ok, ('out', info)
    ok, info_x = _ok(torch.stack(list(model._x_layers)).detach().cpu().numpy(),
                     inter['x_layers'].detach().float().numpy(), inter['x_layers'].detach().numpy())
    assert ok, ('x_layers', info_x)
    ok, info_n = _ok(model._node_out.detach().cpu().numpy(), inter['node_out'].detach().float().numpy(),
                     inter['node_out'].detach().numpy())
    assert ok, ('node_out', info_n)
    rep = {}
    _check_gradients(model, p64, O.pamnet_forward, sd, cfg, b, rep)
    print('QM9 B=128 d128 L6 vs oracle: out %.2e (ref fp32 %.2e), x_layers %.2e, node_out %.2e, worst grad %.2e '
          '(ref fp32 %.2e, %s), |grad| rel %.1e' % (info + (info_x[0], info_n[0]) + rep['grad_worst'] + (rep['grad_norm_rel'],)))


@pytest.mark.gpu
def test_baseline_rna_b8_vs_oracle(dev):
    """BASELINE configs[4] EXACTLY (RNA-Puzzles schema, dim=16, n_layer=1, B=8 graphs of 800-3 900 nodes, kNN global
    graph, flow=target_to_source, mean pool): scores and gradients against the CPU oracle."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    _oracle_threads()
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    b = synth.rna_batch(2, 0, 8)
    sd = O.init_state_dict(cfg, seed=5)
    model = models.PAMNet(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    data = b.to(dev)
    out = model(data)
    torch.nn.functional.l1_loss(out, data.y).backward()
    ref32 = O.pamnet_forward(sd, cfg, b.x, b.batch)
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    inter = {}
    ref64 = O.pamnet_forward(p64, cfg, b.x.double(), b.batch, dtype=torch.float64, intermediates=inter)
    torch.nn.functional.l1_loss(ref64, b.y.double()).backward()
    gc = model._graph_cache
    assert gc.glob.m == inter['edge_index_g'].size(1) and gc.loc.m == inter['edge_index_l'].size(1)
    assert gc.n_trip == inter['idx_kj'].numel() and gc.n_pair == inter['idx_jj_pair'].numel()
    ok, info = _ok(out.detach().cpu().numpy(), ref32.numpy(), ref64.detach().numpy())
    assert ok, ('out', info)
    ok, info_x = _ok(torch.stack(list(model._x_layers)).detach().cpu().numpy(),
                     inter['x_layers'].detach().float().numpy(), inter['x_layers'].detach().numpy())
    assert ok, ('x_layers', info_x)
    rep = {}
    _check_gradients(model, p64, O.pamnet_forward, sd, cfg, b, rep)
    print('RNA B=8 d16 L1 vs oracle: out %.2e (ref fp32 %.2e), x_layers %.2e, worst grad %.2e (ref fp32 %.2e, %s)'
          % (info + (info_x[0],) + rep['grad_worst']))


@pytest.mark.gpu
def test_baseline_pdbbind_b32_vs_oracle_by_shard(dev):
    """BASELINE configs[3] (PDBbind schema, dim=128, n_layer=3, 32 complexes, ~19 k nodes, ~0.7 M global edges): the
    full batch on the GPU against the CPU oracle evaluated shard by shard (4 x 8 complexes: complexes are independent
    units, and an 8-complex shard is what the oracle finishes in seconds).  Outputs are reported both relative to the
    per-complex summed |node_out| (complex - pocket - ligand cancels ~1000x) and raw."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    _oracle_threads()
    cfg = models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    graphs = [synth.pdbbind_complex(1, i) for i in range(32)]
    sd = O.init_state_dict(cfg, seed=3)
    model = models.PAMNet(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    with torch.no_grad():
        out = model(synth.collate(graphs).to(dev)).cpu().numpy()
        node_out = model._node_out.cpu().numpy()
    p64 = {k: v.double() for k, v in sd.items()}
    worst = (0.0, 0.0)
    raw = (0.0, 0.0)
    n0 = 0
    for s in range(4):
        b = synth.collate(graphs[8 * s:8 * s + 8])
        with torch.no_grad():
            inter = {}
            r64 = O.pamnet_forward(p64, cfg, b.x.double(), b.batch, dtype=torch.float64, intermediates=inter).numpy()
            r32 = O.pamnet_forward(sd, cfg, b.x, b.batch).numpy()
        n = b.x.size(0)
        # the model's inspection hook holds the pooling INPUT: node_out * all_index (+-1 per copy, models.py:218-219)
        ok, info = _ok(node_out[n0:n0 + n], inter['pool_in'].float().numpy(), inter['pool_in'].numpy())
        assert ok, ('pool_in shard %d' % s, info)
        pin = inter['pool_in'].abs()
        scale = max(float(pin[b.batch == g].sum()) for g in range(8))
        ok, info = _ok(out[8 * s:8 * s + 8], r32, r64, scale)
        assert ok, ('out shard %d' % s, info)
        worst = max(worst, info)
        raw = max(raw, (maxnorm_err(out[8 * s:8 * s + 8], r64), maxnorm_err(r32, r64)))
        n0 += n
    assert raw[0] <= max(TOL, 2 * raw[1]), raw
    print('PDBbind B=32 d128 L3 vs oracle (4 shards): err/sum|node_out| %.2e (ref fp32 %.2e); raw max|d|/max|out| %.2e '
          '(ref fp32 %.2e)' % (worst + raw))


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['pdbbind_d128_l3', 'qm9s_d128_l2', 'qm9_d128_l6_b16', 'qm9_ragged_d128_l2',
                                  'qm9s_ragged_d128_l2', 'qm9_no_edges_d128_l2', 'qm9_d128_l1',
                                  'pdbbind_d64_l3', 'qm9s_d64_l2', 'qm9_d16_l6_b16', 'qm9_ragged_d64_l2',
                                  'qm9s_ragged_d16_l2', 'qm9_no_edges_d16_l2', 'qm9_d64_l1', 'qm9_d128_l9'])
def test_fused_engine_fresh_inputs_vs_oracle(dev, case):
    """dim=128 (fused MFMA engine) and dim=16/64 (narrow-width row kernels) on fresh seeded inputs vs the CPU oracle:
    forward (fp32+fp64 oracle) and the fp64 loss gradient, for the PDBbind branch (init_linear, +-1 pooling signs,
    local = global edges <= cutoff_l), PAMNet_s (pairs only), the headline QM9 configuration, degenerate molecules and
    a batch without any edge."""
    import re
    import models
    dim = int(re.search(r'_d(\d+)_', case).group(1))
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    small = case.startswith('qm9s')
    if 'ragged' in case or 'no_edges' in case:
        # degenerate molecules: single atoms, a single bond, atoms out of range; and a batch without any edge at all
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.ragged_qm9_batch()
        if 'no_edges' in case:
            b = synth.collate([dict(x=np.array([k % 5], np.float32), pos=np.array([[20.0 * k, 0, 0]], np.float32),
                                    edge_index=np.zeros((2, 0), np.int64), y=np.float32(k)) for k in range(3)])
    elif case.endswith('_l1') or case.endswith('_l9'):
        # single layer pair: no fused next-layer head, no layer-to-layer hand-over; nine pairs: 18 chains = two launches of
        # the batched head kernels (16 chains per launch)
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=int(case[-1]), cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(19, 0, 5)
    elif case.startswith('pdbbind'):
        cfg = models.Config(dataset='PDBbind', dim=dim, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
        b = synth.pdbbind_batch(9, 0, 2, n_pocket=90, n_ligand=16)
    elif small:
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(13, 0, 12)
    else:
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(17, 0, 16)
    sd = O.init_state_dict(cfg, seed=31, small=small)
    model = (models.PAMNet_s if small else models.PAMNet)(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    data = b.to(dev)
    out = model(data)
    loss = torch.nn.functional.l1_loss(out, data.y)
    loss.backward()
    fwd = O.pamnet_s_forward if small else O.pamnet_forward
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    ref32 = fwd(sd, cfg, b.x, b.batch, pos, ei)
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    x64 = b.x.double() if cfg.dataset == 'PDBbind' else b.x
    inter = {}
    ref64 = fwd(p64, cfg, x64, b.batch, pos, ei, dtype=torch.float64, intermediates=inter)
    torch.nn.functional.l1_loss(ref64, b.y.double()).backward()
    scale = None
    if cfg.dataset == 'PDBbind':
        pin = inter['pool_in'].detach().abs()
        scale = max(float(pin[b.batch == g].sum()) for g in range(int(b.batch.max()) + 1))
    ok, info = _ok(out.detach().cpu().numpy(), ref32.numpy(), ref64.detach().numpy(), scale)
    assert ok, ('out', info)
    ok, info = _ok(torch.stack(list(model._x_layers)).detach().cpu().numpy(), inter['x_layers'].detach().float().numpy(),
                   inter['x_layers'].detach().numpy())
    assert ok, ('x_layers', info)
    _check_gradients(model, p64, fwd, sd, cfg, b)


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [128, 32])
def test_self_loops_in_the_bond_list_are_dropped(dev, dim):
    """remove_self_loops (models.py:63): a bond list with self loops gives the same graph, outputs and gradients as the
    clean one (the graph builder assumes a clean list and redoes the local graph when the device-side check fails)."""
    import copy
    import models
    from pamnet_amd import synth
    cfg = models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    torch.manual_seed(11)
    model = models.PAMNet(cfg).to(dev)
    clean = synth.qm9_batch(23, 0, 6).to(dev)
    dirty = copy.copy(clean)
    n = clean.x.numel()
    loops = torch.tensor([[0, 3, n - 1], [0, 3, n - 1]], dtype=clean.edge_index.dtype, device=dev)
    dirty.edge_index = torch.cat([clean.edge_index[:, :5], loops[:, :2], clean.edge_index[:, 5:], loops[:, 2:]], 1)
    res = []
    for b in (clean, dirty):
        model.zero_grad()
        out = model(b)
        out.sum().backward()
        g = model._graph_cache
        res.append((out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None],
                    g.loc.m, g.tp.m))
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))


@pytest.mark.gpu
def test_large_batch_equals_its_shards(dev):
    """BASELINE configs[2] in its single-box form (QM9 schema, dim=128, n_layer=6, B=1 024 = 8 x 128, all 12 targets of
    main_qm9.py:60-66): N ~ 18 k, E_g ~ 260 k -- multi-chunk edge workgroups, unsplit segment sums, many weight-gradient
    slots.  Graphs are independent units, so for EVERY target column the outputs and the loss gradient of
    the 1 024-molecule batch equal those assembled from its eight 128-molecule shards, each shard's gradient scaled by
    local/global graphs exactly as the 8 ranks of the data-parallel run scale theirs before the all-reduce
    (train.Trainer.forward_backward)."""
    import models
    from pamnet_amd import synth
    torch.manual_seed(3)
    model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
    big = synth.qm9_batch(5, 0, 1024).to(dev)
    shards = [synth.qm9_batch(5, 128 * i, 128).to(dev) for i in range(8)]
    table = torch.from_numpy(synth.qm9_label_table(5, 0, 1024)).to(dev)            # [1024, 16] label columns
    named = [(k, p) for k, p in model.named_parameters()]
    worst_out = worst_grad = 0.0
    for target in range(12):
        col = target + 5 if target in (7, 8, 9, 10) else target                    # main_qm9.py:61-66
        y = table[:, col].contiguous()
        for _, p in named:
            p.grad = None
        out = model(big)
        torch.nn.functional.l1_loss(out, y).backward()
        g_big = {k: p.grad.clone() for k, p in named if p.grad is not None}
        for _, p in named:
            p.grad = None
        outs = []
        for i, b in enumerate(shards):
            o = model(b)
            outs.append(o.detach())
            (torch.nn.functional.l1_loss(o, y[128 * i:128 * i + 128]) * (128 / 1024)).backward()
        e = maxnorm_err(out.detach().cpu().numpy(), torch.cat(outs).cpu().numpy())
        assert e < 2e-6, (target, e)
        worst_out = max(worst_out, e)
        assert len(g_big) == len(named) - 1 and 'init_linear.weight' not in g_big     # (unused on the QM9 branch)
        for k, p in named:
            if k not in g_big:
                continue
            e = maxnorm_err(p.grad.cpu().numpy(), g_big[k].cpu().numpy())
            assert e < 2e-5, (target, k, e)
            worst_grad = max(worst_grad, e)
    print('QM9 B=1024 d128 L6, 12 targets: batch vs its 8 shards: outputs %.1e, worst parameter gradient %.1e'
          % (worst_out, worst_grad))


def _baseline_batch(name):
    from pamnet_amd import synth
    return {'baseline_qm9_b32': lambda: synth.qm9_batch(0, 0, 32),
            'baseline_qm9_b128': lambda: synth.qm9_batch(0, 0, 128),
            'baseline_qm9s_b128': lambda: synth.qm9_batch(0, 0, 128),
            'baseline_pdbbind_b8': lambda: synth.collate([synth.pdbbind_complex(1, i) for i in range(8)]),
            'baseline_pdbbind_b32': lambda: synth.collate([synth.pdbbind_complex(1, i) for i in range(32)]),
            'baseline_rna_b8': lambda: synth.rna_batch(2, 0, 8)}[name]()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['baseline_qm9_b32', 'baseline_qm9_b128', 'baseline_pdbbind_b8', 'baseline_pdbbind_b32',
                                  'baseline_rna_b8', 'baseline_qm9s_b128'])
def test_baseline_sizes_vs_reference_runs(dev, golden, name):
    """The HIP path against runs of the REFERENCE ITSELF (tests/golden/gen/gen_golden.py --baseline-only, fp32 and fp64)
    at the BASELINE.json batch sizes: configs[0] (QM9 d=128 L=6 B=32), configs[1] (B=128 -- the exact batch bench.py
    times), configs[3] (PDBbind d=128 L=3 B=32, and its first 8-complex shard), configs[4] (RNA d=16 L=1 B=8).  Graph
    outputs, pooled node values, and the integer sizes of the local graph / triplets / pairs (exact)."""
    import models
    from oracle import pamnet_oracle as O
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    b = _baseline_batch(name)
    assert b.x.size(0) == int(g['num_nodes']) and abs(float(b.x.double().abs().sum()) - float(g['x_checksum'])) < 1e-6
    small = 'qm9s' in name                       # PAMNet_s (pairs only, models.py:283-353) at the batch bench.py times
    sd = O.init_state_dict(cfg, seed=int(g['seed']), small=small)
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(g['weights_checksum'])) < 1e-6
    model = (models.PAMNet_s if small else models.PAMNet)(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    with torch.no_grad():
        out = model(b.to(dev)).cpu().numpy()
    node_out = model._node_out.cpu().numpy()
    ok, info_n = _ok(node_out, g['node_out32'], g['node_out64'])
    assert ok, ('node_out', info_n)
    gc = model._graph_cache
    assert gc.loc.m == int(g['num_edges_l']) and gc.n_trip == int(g['num_triplets']) and gc.n_pair == int(g['num_pairs'])
    if cfg.dataset == 'PDBbind':
        batch = b.batch.numpy()
        scale = max(float(np.abs(g['node_out64'][batch == k]).sum()) for k in range(len(g['out64'])))
        ok, info = _ok(out, g['out32'], g['out64'], scale)
        assert ok, ('out', info)
        raw, raw_floor = maxnorm_err(out, g['out64']), maxnorm_err(g['out32'], g['out64'])
        assert raw <= max(TOL, 2 * raw_floor), (raw, raw_floor)
        print('%s vs the reference: err/sum|node_out| %.2e (ref fp32 %.2e); raw max|d|/max|out| %.2e (ref fp32 %.2e); '
              'node_out %.2e' % (name, info[0], info[1], raw, raw_floor, info_n[0]))
    else:
        ok, info = _ok(out, g['out32'], g['out64'])
        assert ok, ('out', info)
        print('%s vs the reference: out %.2e (ref fp32 %.2e), node_out %.2e (ref fp32 %.2e)' % ((name,) + info + info_n))


@pytest.mark.gpu
def test_trainer_step_path_at_configs1_vs_oracle_and_reference(dev, golden):
    """The EXACT path bench.py times -- train.Trainer.forward_backward: flat parameters, gradients written in place by the
    kernels, the forward recorded on the model's tape and replayed directly, bf16x6 weight gradients -- at BASELINE
    configs[1] (QM9 schema, d=128, L=6, B=128, the bench batch), checked DIRECTLY:
      * every parameter gradient against the CPU oracle's fp64 autograd (floor measured with the fp32 oracle),
      * loss, global gradient norm, every parameter gradient's L2 norm and a set of full gradient tensors against the
        REFERENCE's own fp64 autograd at this batch (tests/golden/baseline_qm9_b128.npz)."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth, train
    _oracle_threads()
    g = golden('baseline_qm9_b128')
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    b = synth.qm9_batch(0, 0, 128)
    sd = O.init_state_dict(cfg, seed=0)
    model = models.PAMNet(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    tr = train.Trainer(model, lr=1e-4)
    assert model._one_node()                               # the one-node tape path with direct gradients is what runs
    data = b.to(dev)
    loss = tr.forward_backward(data)
    first = tr.fp.grad.clone()
    loss2 = tr.forward_backward(data)                      # overwrite semantics: a second pass gives the same bits
    assert torch.equal(first, tr.fp.grad) and torch.equal(loss, loss2)
    # -- the reference's own fp64 backward at this batch
    assert abs(float(loss) - float(g['loss64'])) < 2e-5 * max(1.0, abs(float(g['loss64'])))
    gn = float(torch.linalg.vector_norm(tr.fp.grad.double()))
    assert abs(gn / float(g['grad_norm64']) - 1) < 1e-4, (gn, float(g['grad_norm64']))
    grads = dict(zip(tr.fp.names, tr.fp.grad_views))
    keys = g['grad_keys'].tolist()
    assert set(grads) - set(keys) == {'init_linear.weight'}            # unused on the QM9 branch: no gradient either side
    assert float(grads['init_linear.weight'].abs().max()) == 0.0
    worst_l2 = 0.0
    for k, l2 in zip(keys, g['grad_l2_64']):
        e = abs(float(grads[k].double().norm()) - float(l2)) / max(float(l2), 1e-300)
        assert e < GRAD_TOL, (k, e)
        worst_l2 = max(worst_l2, e)
    worst_t = 0.0
    for k in g.files:
        if k.startswith('grad64/'):
            e = maxnorm_err(grads[k[7:]].cpu().numpy(), g[k])
            assert e < GRAD_TOL, (k, e)
            worst_t = max(worst_t, e)
    # -- every element of every gradient against the oracle's fp64 autograd
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    ref64 = O.pamnet_forward(p64, cfg, b.x, b.batch, b.pos, b.edge_index, dtype=torch.float64)
    torch.nn.functional.l1_loss(ref64, b.y.double()).backward()
    rep = {}
    _check_gradients(model, p64, O.pamnet_forward, sd, cfg, b, rep)
    print('Trainer.forward_backward at configs[1] (B=128 d128 L6): vs reference fp64: |grad| rel %.1e, worst per-tensor '
          'L2 %.1e, worst full tensor %.1e; vs oracle fp64: worst element %.2e (ref fp32 %.2e, %s)'
          % ((abs(gn / float(g['grad_norm64']) - 1), worst_l2, worst_t) + rep['grad_worst']))


@pytest.mark.gpu
@pytest.mark.parametrize('dataset,dim', [('PDBbind', 64), ('PDBbind', 16), ('PDBbind', 128), ('QM9', 32), ('rna_x', 16),
                                         ('QM9:basis', 128), ('QM9:basis', 32)])
def test_trainer_tape_path_trains_every_parameter(dev, dataset, dim):
    """Under train.Trainer the whole forward is ONE recorded node (ops.Tape) that runs with grad mode off: a stage without
    a tape-aware Function would silently leave its parameters without a gradient (the PDBbind `init_linear` at the narrow
    widths once did).  Every parameter gradient of the trainer path -- none identically zero -- against the CPU oracle's
    fp64 autograd, for every schema / engine combination."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth, train
    if dataset == 'PDBbind':
        cfg = models.Config(dataset='PDBbind', dim=dim, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)
        b = synth.pdbbind_batch(9, 0, 2, n_pocket=90, n_ligand=16)
    elif dataset.startswith('QM9'):
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        if dataset.endswith(':basis'):       # a basis the one-launch input stage is not built for (dense fallbacks on the tape)
            cfg.basis = (6, 5, 7)
        b = synth.qm9_batch(17, 0, 16)
        dataset = 'QM9'
    else:
        cfg = models.Config(dataset=dataset, dim=dim, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
        b = synth.rna_batch(3, 0, 2, n_nodes=260)
    sd = O.init_state_dict(cfg, seed=13)
    model = models.PAMNet(cfg, *_basis_args(cfg))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    tr = train.Trainer(model, lr=1e-4)
    assert model._one_node()
    tr.forward_backward(b.to(dev))
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    x64 = b.x.double() if (cfg.dataset == 'PDBbind' or dataset.startswith('rna')) else b.x
    ref64 = O.pamnet_forward(p64, cfg, x64, b.batch, pos, ei, dtype=torch.float64)
    torch.nn.functional.l1_loss(ref64, b.y.double()).backward()
    used = 0
    for k, p in model.named_parameters():
        if p64[k].grad is None or float(p64[k].grad.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k              # a parameter this branch of the model never reads
        else:
            assert float(p.grad.abs().max()) > 0.0, 'no gradient reached %s' % k
            used += 1
    assert used >= len(p64) - 2 and (dataset != 'PDBbind' or float(model.init_linear.weight.grad.abs().max()) > 0.0)
    _check_gradients(model, p64, O.pamnet_forward, sd, cfg, b)


@pytest.mark.gpu
def test_l1_loss_target_shapes_and_dtypes(dev):
    """ops.l1_loss_with_grad takes what F.l1_loss takes: a [B, 1] / float64 / host-resident target gives the same loss and
    gradient as the plain fp32 [B] one; a target of another size raises instead of being read out of bounds."""
    from pamnet_amd import ops
    torch.manual_seed(0)
    out, y = torch.randn(37, device=dev), torch.randn(37, device=dev)
    loss, g = ops.l1_loss_with_grad(out, y, 0.5)
    assert abs(float(loss) - float(torch.nn.functional.l1_loss(out, y))) < 1e-6
    for yy in (y.view(37, 1), y.double(), y.cpu(), y.cpu().double().view(1, 37)):
        l2, g2 = ops.l1_loss_with_grad(out, yy, 0.5)
        assert torch.equal(l2, loss) and torch.equal(g2, g)
    with pytest.raises(ValueError):
        ops.l1_loss_with_grad(out, y[:36])


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['l1', 'mse', 'smooth_l1'])
def test_loss_kernels_vs_torch_losses(dev, kind):
    """The one-launch loss + gradient kernels (pamnet_l1_loss_f32 / pamnet_mse_loss_f32 / pamnet_smooth_l1_loss_f32) against
    F.l1_loss / F.mse_loss / F.smooth_l1_loss (main_qm9.py:108, main_pdbbind.py:93, main_rna_puzzles.py:92) evaluated in
    fp64: value and gradient within 2e-7 (max-normalised), residuals on both sides of the smooth-L1 knee and exactly on it,
    exact zeros, sizes that are not multiples of the workgroup; the DP pre-scaling (grad_scale) scales the gradient only."""
    from pamnet_amd import ops
    F = torch.nn.functional
    fn = {'l1': F.l1_loss, 'mse': F.mse_loss, 'smooth_l1': F.smooth_l1_loss}[kind]
    gen = torch.Generator().manual_seed(5)
    for n in (1, 7, 256, 1000):
        y = torch.randn(n, generator=gen) * 3
        out = y + torch.randn(n, generator=gen) * 1.5
        if n >= 7:
            out[0], out[1], out[2], out[3] = y[0], y[1] + 1.0, y[2] - 1.0, y[3] + 0.999999
        o64 = out.double().requires_grad_(True)
        l64 = fn(o64, y.double())
        l64.backward()
        for scale in (1.0, 0.37):
            loss, g = ops.loss_with_grad(kind, out.to(dev), y.to(dev), scale)
            assert abs(float(loss) - float(l64)) <= 2e-7 * max(1.0, abs(float(l64))), (kind, n)
            ref = o64.grad * scale
            assert float((g.cpu().double() - ref).abs().max()) <= 2e-7 * float(ref.abs().max()) + 1e-30, (kind, n, scale)
    with pytest.raises(ValueError):
        ops.loss_with_grad('huber', out.to(dev), y.to(dev))


@pytest.mark.gpu
def test_trainer_checkpoint_round_trip_and_close(dev):
    """Trainer.state_dict() / load_state_dict(): a resumed trainer continues bit for bit; state_dict() and close() drain the
    steps in flight (the deferred device-side checks of the last batches are read before anything is saved)."""
    import models
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer
    cfg = models.Config(dataset='QM9', dim=128, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)
    bs = [synth.qm9_batch(4, 8 * k, 8).to(dev) for k in range(4)]

    def fresh():
        torch.manual_seed(3)
        return Trainer(models.PAMNet(cfg).to(dev), lr=1e-3)
    a = fresh()
    for k in range(2):
        a.step(bs[k])
    ck = a.state_dict()
    assert not a._inflight and set(ck['model']) == set(a.model.state_dict())
    for k in range(2, 4):
        a.step(bs[k])
    b = fresh()
    b.load_state_dict(ck)
    for k in range(2, 4):
        b.step(bs[k])
    with a, b:                                           # __exit__ -> close() -> drain()
        pass
    assert not a._inflight and not b._inflight
    assert torch.equal(a.fp.flat, b.fp.flat) and torch.equal(a.shadow, b.shadow) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['qm9_d128', 'qm9_d32', 'qm9s_d128', 'pdbbind_d128'])
def test_max_num_neighbors_binding_vs_oracle(dev, case):
    """radius(..., max_num_neighbors) where it BINDS (models.py:110,128: 1000; :301: 500 -- lowered here through the model's
    `max_num_neighbors` attribute so that molecule-sized graphs reach it): the capped global graph is no longer symmetric, the
    model takes the general transposes; outputs and every parameter gradient against the oracle's fp64 run of the same capped
    search, and the uncapped model differs (the cap did something).  A resident store notices the cap when it counts the
    data set and hands such batches over without sizes; a batch that carries sizes anyway is flagged, not mis-built."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, synth
    from pamnet_amd.store import MoleculeStore
    small = case.startswith('qm9s')
    dim = int(case.rsplit('_d', 1)[1])
    if case.startswith('pdbbind'):
        cfg = models.Config(dataset='PDBbind', dim=dim, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)
        graphs = [synth.pdbbind_complex(4, i, n_pocket=60, n_ligand=12) for i in range(3)]
        cap, fwd = 24, O.pamnet_forward
    else:
        cfg = models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        graphs = [synth.qm9_molecule(6, i) for i in range(7)]
        cap, fwd = 12, (O.pamnet_s_forward if small else O.pamnet_forward)
    b = synth.collate(graphs)
    sd = O.init_state_dict(cfg, seed=11, small=small)
    model = (models.PAMNet_s if small else models.PAMNet)(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    data = b.to(dev)
    with torch.no_grad():
        free = model(data).cpu()
    assert not model._graph_cache.capped
    model.max_num_neighbors = cap
    out = model(data)
    assert model._graph_cache.capped
    torch.nn.functional.l1_loss(out, data.y).backward()
    p64 = O.as_params({k: v.double() for k, v in sd.items()})
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    x64 = b.x.double() if cfg.dataset == 'PDBbind' else b.x
    ref = fwd(p64, cfg, x64, b.batch, pos, ei, dtype=torch.float64, max_num_neighbors=cap)
    torch.nn.functional.l1_loss(ref, b.y.double()).backward()
    scale = None
    if cfg.dataset == 'PDBbind':
        inter = {}
        fwd({k: v.double() for k, v in sd.items()}, cfg, x64, b.batch, pos, ei, dtype=torch.float64, intermediates=inter,
            max_num_neighbors=cap)
        scale = max(float(inter['pool_in'].abs()[b.batch == k].sum()) for k in range(len(graphs)))
    e = float((out.detach().cpu().double() - ref.detach()).abs().max()) / (scale or float(ref.detach().abs().max()))
    assert e < TOL, e
    assert float((free.double() - ref.detach()).abs().max()) / (scale or float(ref.detach().abs().max())) > 100 * TOL
    fwd_cap = lambda *a, **k: fwd(*a, max_num_neighbors=cap, **k)
    # (W_out.bias of a PDBbind head: sum over N nodes of +-1/B times a softmax weight that sums to 1 over the 2L heads)
    _check_gradients(model, p64, fwd_cap, sd, cfg, b, head_bias_terms=b.x.size(0) / (len(graphs) * 2.0 * cfg.n_layer))
    # the store: counts taken with the cap in force; the data set is marked and its batches carry no sizes for this model
    st = MoleculeStore(graphs, dev).prepare_for(model)
    bt = st.collate(list(range(len(graphs))))
    assert not bt.sizes
    with torch.no_grad():
        assert torch.equal(model(bt), out.detach())
    model.verify()
    # sizes forced onto such a batch (counted without the cap): flagged by the count pass, never silently mis-built
    model.max_num_neighbors = 0
    st2 = MoleculeStore(graphs, dev).prepare_for(model)
    bt = st2.collate(list(range(len(graphs))))
    model.max_num_neighbors = cap
    from pamnet_amd.store import size_key
    bt.sizes = {size_key(model): next(iter(bt.sizes.values()))}
    bt.mol_local = False
    with torch.no_grad():
        model(bt)
    with pytest.raises(G.GraphCheckError, match='max_num_neighbors'):
        model.verify()


@pytest.mark.gpu
@pytest.mark.parametrize('dataset,dim,small', [('QM9', 96, False), ('QM9', 100, True), ('PDBbind', 96, False), ('QM9', 48, False),
                                                ('QM9', 20, True), ('rna_x', 12, False), ('PDBbind', 40, False),
                                                ('QM9', 130, False), ('QM9', 141, True)])
def test_any_dim_runs_on_the_engines_vs_oracle(dev, dataset, dim, small):
    """dim = 96 / 100 (-> the fused 128-wide engine), 48 / 40 (-> 64), 20 (-> 32), 12 (-> 16): the oracle's weights at the
    LOGICAL width load through the padding hooks; outputs and every parameter gradient (compared on the logical blocks;
    the padded rest must be exactly zero) against the oracle's fp64 run; the engines ran (one recorded node under the
    trainer); five optimiser steps leave the padding exactly zero and the logical weights equal to a run of the same model
    on torch.optim.Adam (i.e. padding does not leak into the update)."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth, train
    rna = dataset.startswith('rna')
    if dataset == 'PDBbind':
        cfg = models.Config(dataset='PDBbind', dim=dim, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)
        b = synth.pdbbind_batch(3, 0, 2, n_pocket=60, n_ligand=12)
    elif rna:
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
    # (dim = 130 / 141 -> 132 / 144 on the any-width GEMM kernels of csrc/dense.hip: the same padding above 128)
    assert model.dim != dim and (model.dim in models.ENGINE_WIDTHS if dim <= 128 else model.dim == (dim + 3) // 4 * 4)
    data = b.to(dev)
    out = model(data)
    torch.nn.functional.l1_loss(out, data.y).backward()
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

    class _Logical(object):                      # the model seen through its logical blocks, for _check_gradients
        def named_parameters(self_):
            for k, p in model.named_parameters():
                mask, shape = model.logical_mask(k), model._logical_shapes[k]
                q = p.detach()[mask].reshape(shape)
                q.grad = None if p.grad is None else p.grad[mask].reshape(shape)
                if p.grad is not None and not bool(mask.all()):
                    assert float(p.grad[~mask].abs().max()) == 0.0, ('padding gradient', k)
                yield k, q

        def parameters(self_):
            return [q for _, q in self_.named_parameters()]
    _check_gradients(_Logical(), p64, fwd, sd, cfg, b,
                     head_bias_terms=(b.x.size(0) / (2 * 2.0 * cfg.n_layer)) if dataset == 'PDBbind' else None)
    # training: Trainer (flat buffers, fused Adam + EMA over the PADDED parameters) against torch Adam on a twin
    twin = (models.PAMNet_s if small else models.PAMNet)(cfg)
    twin.load_state_dict(sd, strict=True)
    twin = twin.to(dev)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    tr = train.Trainer(model, lr=1e-3)
    assert model._one_node() or dim > 128
    for _ in range(5):
        tr.step(data)
        opt.zero_grad()
        torch.nn.functional.l1_loss(twin(data), data.y).backward()
        torch.nn.utils.clip_grad_norm_(twin.parameters(), 1000.0)
        opt.step()
    tr.drain()
    a, c = model.state_dict(), twin.state_dict()
    for k in a:
        assert maxnorm_err(a[k].cpu().numpy(), c[k].cpu().numpy()) < 2e-4, k      # (Adam amplifies rounding: g / |g|)
    for k, p in model.named_parameters():
        assert float(p.detach()[~model.logical_mask(k)].abs().sum()) == 0.0 and float(tr.shadow.abs().sum()) > 0, k


@pytest.mark.gpu
def test_out_of_range_inputs_raise_index_error(dev):
    """Atom types beyond the embedding table or bond endpoints beyond the node count raise
    IndexError (as indexing does in the reference, models.py:107) instead of writing out of bounds on the device."""
    import copy
    import models
    from pamnet_amd import synth
    model = models.PAMNet(models.Config(dataset='QM9', dim=32, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
    good = synth.qm9_batch(3, 0, 4).to(dev)
    with torch.no_grad():
        ref = model(good)
        for field, mutate in (('x', lambda t: torch.cat([t[:-1], t.new_tensor([7.0])])),
                              ('edge_index', lambda t: torch.cat([t[:, :-1], t.new_tensor([[0], [10 ** 6]])], 1))):
            bad = copy.copy(good)
            setattr(bad, field, mutate(getattr(good, field)))
            with pytest.raises(IndexError):
                model(bad)
        assert torch.equal(model(good), ref)                  # the model is still usable afterwards
    rna = models.PAMNet(models.Config(dataset='rna_x', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0)).to(dev)
    b = synth.rna_batch(3, 0, 1, n_nodes=120).to(dev)
    b.x = b.x.clone()
    b.x[5, -1] = 3.0                                           # three atom types: 0, 1, 2
    with pytest.raises(IndexError), torch.no_grad():
        rna(b)


@pytest.mark.gpu
def test_flat_parameter_view_semantics(monkeypatch):
    """PAMNET_FLAT_PARAMS=1 (models._FlatView): parameters() / named_parameters() yield ONE flat nn.Parameter, state_dict() keeps
    the reference keys; its .grad is what the per-tensor interface computes, tensor by tensor; a second backward without
    zero_grad accumulates (the kernels overwrite: the view sets the old gradient aside and adds it back); zero_grad in place
    works; a device move re-builds the view; a Trainer on the model switches it off."""
    import models
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer
    dev = torch.device('cuda:0')
    cfg = models.Config(dataset='QM9', dim=128, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)
    b = synth.qm9_batch(3, 0, 6).to(dev)
    torch.manual_seed(5)
    monkeypatch.setenv('PAMNET_FLAT_PARAMS', '0')
    plain = models.PAMNet(cfg).to(dev)
    sd = {k: v.clone() for k, v in plain.state_dict().items()}
    assert len(list(plain.parameters())) == len(sd)
    plain(b).sum().backward()
    want = {k: p.grad.clone() for k, p in plain.named_parameters() if p.grad is not None}

    monkeypatch.setenv('PAMNET_FLAT_PARAMS', '1')
    model = models.PAMNet(cfg)
    model.load_state_dict(sd)
    assert len(list(model.parameters())) == len(sd)              # on the CPU: nothing to flatten into
    model = model.to(dev)
    named = list(model.named_parameters())
    assert [n for n, _ in named] == [models.FLAT_NAME] and len(list(model.parameters())) == 1
    flat = named[0][1]
    assert isinstance(flat, torch.nn.Parameter) and flat.requires_grad and flat.numel() >= sum(v.numel() for v in sd.values())
    assert set(model.state_dict().keys()) == set(sd.keys())
    assert all(torch.equal(v, sd[k]) for k, v in model.state_dict().items())
    view = model._flat_view()

    def grads_by_name():
        out = {}
        for n, p in zip(view.fp.names, view.fp.params):
            o = view.fp.offsets[n]
            out[n] = flat.grad[o:o + p.numel()].view_as(p)
        return out

    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    opt.zero_grad()                                               # set_to_none: .grad is None going into the forward
    assert flat.grad is None
    model(b).sum().backward()
    assert flat.grad is not None
    got = grads_by_name()
    for k, w in want.items():
        scale = float(w.abs().max())
        assert float((got[k] - w).abs().max()) <= 1e-6 * max(scale, 1e-30), k
    g1 = flat.grad.clone()
    model(b).sum().backward()                                     # no zero_grad in between: accumulate
    assert torch.allclose(flat.grad, 2 * g1, rtol=1e-6, atol=0)
    opt.zero_grad(set_to_none=False)
    assert float(flat.grad.abs().max()) == 0.0
    model(b)                                                      # a forward that is never differentiated changes nothing
    model(b).sum().backward()
    assert torch.equal(flat.grad, g1)
    # load_state_dict writes through the views
    model.load_state_dict({k: v * 0.5 for k, v in sd.items()})
    assert torch.equal(model.state_dict()['rbf_g.freq'], sd['rbf_g.freq'] * 0.5) and float(flat.detach().abs().sum()) > 0
    # a device round trip re-allocates the parameters: the view is rebuilt, with a new flat parameter
    model = model.cpu()
    assert len(list(model.parameters())) == len(sd)
    model = model.to(dev)
    again = list(model.parameters())
    assert len(again) == 1 and again[0] is not flat
    # a Trainer owns flat buffers of its own: the interface goes back to per-tensor
    Trainer(model, lr=1e-4)
    assert len(list(model.parameters())) == len(sd)


@pytest.mark.gpu
def test_edge_recompute_switch_gives_the_same_step_bit_for_bit():
    """PAMNET_EDGE_RECOMPUTE=1 (round 6 A/B, csrc/engine.hip): the training forward of the fused global-edge step saves no z / ea
    rows, the backward re-runs the forward kernel for them.  Same kernel, same inputs: the output and every gradient of a
    PDBbind-sized step are bit for bit those of the saving form (each form in a process of its own: the switch is read once)."""
    import hashlib
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import models\n"
        "from pamnet_amd import synth\n"
        "from pamnet_amd.train import Trainer\n"
        "dev = torch.device('cuda:0'); torch.manual_seed(3)\n"
        "cfg = models.Config(dataset='PDBbind', dim=128, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)\n"
        "model = models.PAMNet(cfg).to(dev)\n"
        "tr = Trainer(model, loss='mse', max_grad_norm=None, ema_decay=None, lr=1e-3)\n"
        "b = synth.collate([synth.pdbbind_complex(1, i) for i in range(12)]).to(dev)\n"
        "loss = tr.forward_backward(b)\n"
        "torch.cuda.synchronize()\n"
        "g = model._graph_cache\n"
        "print('EG', int(g.glob.m))\n"
        "print('HASH', hashlib.sha256(tr.fp.grad.cpu().numpy().tobytes()).hexdigest(), float(loss), float(tr.fp.grad.abs().sum()))\n"
    ) % (repo, os.path.join(repo, 'physics-aware-multiplex-gnn_amd'))
    outs = []
    for v in ('0', '1'):
        env = dict(os.environ, PAMNET_EDGE_RECOMPUTE=v)
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l.startswith(('EG', 'HASH'))}
        assert int(lines['EG'][0]) >= 131072          # the fused weight-gradient backward: where the switch applies
        outs.append(lines['HASH'])
    assert float(outs[0][2]) > 0 and outs[0] == outs[1], outs


@pytest.mark.gpu
@pytest.mark.parametrize('switch', ['PAMNET_EDGE_IMAGES', 'PAMNET_FUSE_LOCAL_AGG'])
@pytest.mark.parametrize('dataset', ['QM9', 'PDBbind'])
def test_edge_fragment_images_leave_every_bit_of_a_step_unchanged(dataset, switch):
    """Round 6: the edge-level kernels read their weight slices as ready-made bf16x3 fragment images packed once per step
    direction (PAMNET_EDGE_IMAGES, default on) instead of splitting the fp32 slices in every workgroup.  The pieces are the
    same pieces: loss and the whole flat gradient of a training step are bit for bit those of PAMNET_EDGE_IMAGES=0 (each form
    in a process of its own: the switch is read once).  The same protocol for the local layer's aggregations formed inside its
    forward chain launch (PAMNET_FUSE_LOCAL_AGG): the same rows in the same order as the stand-alone kernel.  (The segment sums
    formed inside the fused head + chain backward launch, PAMNET_FUSE_SEGSUM, add a node's rows strictly in CSR order where the
    stand-alone kernels keep several partial sums: same values to fp32 rounding, not the same bits -- covered by the gradient
    goldens.)"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    make = ("cfg = models.Config(dataset='QM9', dim=128, n_layer=3, cutoff_l=5.0, cutoff_g=5.0)\n"
            "b = synth.qm9_batch(24, 0, 7).to(dev)\n") if dataset == 'QM9' else (
            "cfg = models.Config(dataset='PDBbind', dim=128, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)\n"
            "b = synth.pdbbind_batch(3, 0, 2, n_pocket=60, n_ligand=12).to(dev)\n")
    code = (
        "import sys, hashlib, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import models\n"
        "from pamnet_amd import synth\n"
        "from pamnet_amd.train import Trainer\n"
        "dev = torch.device('cuda:0'); torch.manual_seed(3)\n"
        + make +
        "model = models.PAMNet(cfg).to(dev)\n"
        "tr = Trainer(model, loss='l1', max_grad_norm=None, ema_decay=None, lr=1e-3)\n"
        "loss = tr.forward_backward(b)\n"
        "torch.cuda.synchronize()\n"
        "print('HASH', hashlib.sha256(tr.fp.grad.cpu().numpy().tobytes()).hexdigest(), float(loss), float(tr.fp.grad.abs().sum()))\n"
    ) % (repo, os.path.join(repo, 'physics-aware-multiplex-gnn_amd'))
    outs = []
    for v in ('0', '1'):
        env = dict(os.environ, **{switch: v})
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l.split()[1:] for l in r.stdout.splitlines() if l.startswith('HASH')][0])
    assert float(outs[0][2]) > 0 and outs[0] == outs[1], outs


@pytest.mark.gpu
@pytest.mark.parametrize('setting', ['PAMNET_CHAIN_BF16=0', 'PAMNET_CHAIN_BF16=2', 'PAMNET_CHAIN_WAVES=4'])
def test_node_chain_forms_give_the_same_step(setting, tmp_path):
    """Round 6: the node chains of single-round batches run as bf16x6 piece products (default: both directions, 8 waves).  The
    other forms behind their switches -- fp32 MFMAs (PAMNET_CHAIN_BF16=0), bf16x6 forward only (=2), the 4-wave forward geometry
    (PAMNET_CHAIN_WAVES=4) -- must give the same training step to fp32 accuracy: the loss to 1e-6, the flat gradient within 2e-5
    of its largest entry (each form in a process of its own: the switches are read once)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import models\n"
        "from pamnet_amd import synth\n"
        "from pamnet_amd.train import Trainer\n"
        "dev = torch.device('cuda:0'); torch.manual_seed(3)\n"
        "cfg = models.Config(dataset='QM9', dim=128, n_layer=3, cutoff_l=5.0, cutoff_g=5.0)\n"
        "b = synth.qm9_batch(24, 0, 7).to(dev)\n"
        "model = models.PAMNet(cfg).to(dev)\n"
        "tr = Trainer(model, loss='l1', max_grad_norm=None, ema_decay=None, lr=1e-3)\n"
        "loss = tr.forward_backward(b)\n"
        "torch.cuda.synchronize()\n"
        "torch.save({'loss': float(loss), 'grad': tr.fp.grad.cpu()}, sys.argv[1])\n"
    ) % (repo, os.path.join(repo, 'physics-aware-multiplex-gnn_amd'))
    res = []
    for i, extra in enumerate(({}, dict([setting.split('=')]))):
        out = str(tmp_path / ('step%d.pt' % i))
        env = {k: v for k, v in os.environ.items() if not k.startswith('PAMNET_CHAIN_')}
        env.update(extra)
        r = subprocess.run([sys.executable, '-c', code, out], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(out))
    a, b = res
    assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(a['loss'])
    scale = float(a['grad'].abs().max())
    assert scale > 0 and float((a['grad'] - b['grad']).abs().max()) <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['pamnet_s_d128', 'pamnet_d32', 'pdbbind_d128'])
def test_flat_parameter_view_gradients_other_models(kind, monkeypatch):
    """PAMNET_FLAT_PARAMS=1 on the other model kinds (PAMNet_s, a narrow width, the PDBbind branch): one step of the reference's
    loop body gives the parameters the per-tensor interface gives (SGD, so that the comparison is of the gradients themselves)."""
    import models
    from pamnet_amd import synth
    dev = torch.device('cuda:0')
    if kind == 'pamnet_s_d128':
        cls, cfg = models.PAMNet_s, models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(4, 0, 5).to(dev)
    elif kind == 'pamnet_d32':
        cls, cfg = models.PAMNet, models.Config(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(4, 0, 5).to(dev)
    else:
        cls, cfg = models.PAMNet, models.Config(dataset='PDBbind', dim=128, n_layer=1, cutoff_l=2.0, cutoff_g=6.0)
        b = synth.pdbbind_batch(3, 0, 2, n_pocket=60, n_ligand=12).to(dev)
    results = []
    for flat in ('0', '1'):
        monkeypatch.setenv('PAMNET_FLAT_PARAMS', flat)
        torch.manual_seed(9)
        model = cls(cfg).to(dev)
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        assert len(opt.param_groups[0]['params']) == (1 if flat == '1' else len(model.state_dict()))
        for _ in range(2):
            opt.zero_grad()
            loss = torch.nn.functional.l1_loss(model(b), b.y)
            loss.backward()
            opt.step()
        results.append(({k: v.clone() for k, v in model.state_dict().items()}, float(loss.detach())))
    (a, la), (f, lf) = results
    assert abs(la - lf) <= 1e-6 * max(1.0, abs(la))
    for k in a:
        scale = float(a[k].abs().max())
        assert float((a[k] - f[k]).abs().max()) <= 2e-6 * max(scale, 1e-30), k

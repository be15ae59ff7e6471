"""Resident dataset, device-side collation and the zero-host-sync forward (pamnet_amd/store.py).
GPU tests: the collated batch equals the host-side collation bit for bit; the per-molecule size table equals what graph
construction finds; a forward on a batch that carries its sizes issues no synchronising call (torch's sync debug mode set
to 'error') and returns bitwise what the plain forward returns; sizes that do not belong to the batch -- too small or too
large -- are caught by the deferred device-side check without touching memory out of bounds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs an MI355X'
    from pamnet_amd import lib
    lib.load()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def store(dev):
    from pamnet_amd import store as S, synth
    mols = [synth.qm9_molecule(3, i) for i in range(300)]
    return S.MoleculeStore(mols, dev), mols


def _model(dev, small=False, dim=128, n_layer=2, cutoff_g=5.0):
    from models import Config, PAMNet, PAMNet_s
    torch.manual_seed(0)
    cls = PAMNet_s if small else PAMNet
    return cls(Config(dataset='QM9', dim=dim, n_layer=n_layer, cutoff_l=5.0, cutoff_g=cutoff_g)).to(dev)


def test_collate_matches_host_collation(dev, store):
    from pamnet_amd import synth
    st, mols = store
    idx = [17, 3, 250, 3, 0, 299, 120]                       # any order, repeats allowed
    b = st.collate(idx, with_sizes=False)
    ref = synth.collate([mols[i] for i in idx])
    assert torch.equal(b.x.cpu(), ref.x.float()) and torch.equal(b.pos.cpu(), ref.pos)
    assert torch.equal(b.batch.cpu().long(), ref.batch) and torch.equal(b.edge_index.cpu().long(), ref.edge_index)
    assert torch.equal(b.y.cpu(), ref.y) and b.num_graphs == len(idx)


@pytest.mark.parametrize('small', [False, True])
def test_size_table_matches_graph_construction(dev, store, small):
    from pamnet_amd import graph as G
    st, _ = store
    eg, el, tp = st.counts_for(_model(dev, small, dim=16, n_layer=1), chunk=128)      # several passes
    assert eg.shape == (300,) and (eg > 0).all() and (tp > 0).all() and (el == st.n_edges).all()
    for idx in ([5], [1, 2, 3], list(range(40, 168))):
        b = st.collate(idx, with_sizes=False)
        g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=len(idx),
                          need_grad=False, with_triplets=not small, n_types=5)
        assert g.glob.m == int(eg[idx].sum()) and g.tp.m == int(tp[idx].sum())


@pytest.mark.parametrize('small', [False, True])
def test_forward_without_host_sync(dev, store, small):
    st, _ = store
    model = _model(dev, small)
    st.prepare_for(model)
    idx = list(range(100, 228))
    with torch.no_grad():
        ref = model(st.collate(idx, with_sizes=False))           # sizes read back from the device
        model(st.collate(idx))                                   # warm every cache the first hinted call fills
        torch.cuda.synchronize()
        b = st.collate(idx)
        assert b.sizes
        torch.cuda.set_sync_debug_mode('error')                  # any synchronising torch call raises from here on
        try:
            out = model(b)
        finally:
            torch.cuda.set_sync_debug_mode('default')
    model.verify()                                               # clean flag words
    assert torch.equal(out, ref)
    # training forward + backward, same property
    b = st.collate(idx)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        loss = (model(b) - b.y).abs().mean()
        loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    model.verify()
    assert torch.isfinite(loss)


def test_wrong_sizes_are_caught(dev, store):
    from pamnet_amd.graph import GraphCheckError
    st, _ = store
    model = _model(dev)
    st.prepare_for(model)
    idx = list(range(64))
    from pamnet_amd.store import size_key
    key = size_key(model)
    for d_eg, d_tp in ((-7, 0), (+9, 0), (0, -5), (0, +11)):
        b = st.collate(idx)
        eg, el, tp = b.sizes[key]
        b.sizes = {key: (eg + d_eg, el, tp + d_tp)}
        with torch.no_grad():
            out = model(b)                                       # runs to completion, memory-safe, result invalid
        assert out.shape == (64,)
        with pytest.raises(GraphCheckError):
            model.verify()
        model.verify()                                           # the pending list is cleared by the raise
    # invalid index inputs surface through the same deferred check, as the reference's IndexError
    b = st.collate(idx)
    b.x = b.x.clone()
    b.x[3] = 7.0
    with torch.no_grad():
        model(b)
    with pytest.raises(IndexError):
        model.verify()


def test_trainer_and_predict_verify(dev, store):
    """Trainer.step on store batches: no host round trip in the steady state; a bad batch raises a few steps later."""
    from pamnet_amd import train
    from pamnet_amd.graph import GraphCheckError
    st, _ = store
    model = _model(dev, n_layer=1)
    st.prepare_for(model)
    tr = train.Trainer(model, lr=1e-4)
    batches = [st.collate(list(range(k * 32, k * 32 + 32))) for k in range(6)]
    losses = [tr.step(batches[k], next_data=batches[k + 1] if k + 1 < 6 else None) for k in range(6)]
    tr.drain()                                                   # every step's own flag words, read once it has completed
    assert all(torch.isfinite(l) for l in losses)
    outs = [o for _, o in train.predict(model, [st.collate(list(range(k * 50, k * 50 + 50))) for k in range(4)])]
    assert len(outs) == 4
    bad = st.collate(list(range(32)))
    key = next(iter(bad.sizes))
    bad.sizes = {key: (bad.sizes[key][0] - 3,) + tuple(bad.sizes[key][1:])}
    with pytest.raises(GraphCheckError):
        for k in range(5):
            tr.step(bad if k == 0 else st.collate(list(range(32))))
        tr.drain()
    tr.drain()                                                   # the raise left nothing behind


def test_evaluation_between_steps_keeps_flag_ownership(dev, store):
    """An evaluation pass between training steps has its own forwards and its own verify(): it must neither swallow nor
    mis-attribute the flag words of the training steps still in flight -- they travel with the steps' events
    (Trainer._throttle), whatever model.verify() is called in between."""
    from pamnet_amd import train
    from pamnet_amd.graph import GraphCheckError
    st, _ = store
    model = _model(dev, n_layer=1)
    st.prepare_for(model)
    tr = train.Trainer(model, lr=1e-4)
    ev = lambda: tr.evaluate([st.collate(list(range(k * 50, k * 50 + 50))) for k in range(2)])
    for k in range(4):
        tr.step(st.collate(list(range(32 * k, 32 * k + 32))))
    assert len(tr._inflight) == tr.MAX_STEPS_IN_FLIGHT and sum(len(f) for _, f in tr._inflight) >= 1
    mae = ev()                                                   # drains the steps in flight first: all clean
    assert np.isfinite(mae) and not tr._inflight and not model._pending_checks
    tr.step(st.collate(list(range(32))))
    worse = st.collate(list(range(32, 64)))
    key = next(iter(worse.sizes))
    worse.sizes = {key: (worse.sizes[key][0] + 5,) + tuple(worse.sizes[key][1:])}
    tr.step(worse)                                               # its flag word is still on the model ...
    tr.step(st.collate(list(range(64, 96))))                     # ... and now travels with this step's event
    with torch.no_grad():
        model(st.collate(list(range(10))))
    model.verify()                                               # a verify() of somebody else's forward: clean, and it
    with pytest.raises(GraphCheckError):                         # did not swallow the training step's word
        ev()
    tr.drain()


def test_store_strips_self_loops_and_keys_its_counts(dev):
    """remove_self_loops (models.py:63) happens once, at ingestion: a dataset whose bond lists contain self loops runs the
    zero-host-sync forward like its clean twin (same outputs, no deferred error).  The per-molecule size tables are keyed
    by everything the sizes depend on: two models with different cutoffs / layer kinds sharing a store get their own."""
    import copy
    from pamnet_amd import store as S, synth
    clean = [synth.qm9_molecule(9, i) for i in range(40)]
    dirty = copy.deepcopy(clean)
    for k in (0, 7, 39):
        n = dirty[k]['x'].shape[0]
        loops = np.array([[0, n - 1], [0, n - 1]], dtype=np.int64)
        dirty[k]['edge_index'] = np.concatenate([dirty[k]['edge_index'][:, :3], loops, dirty[k]['edge_index'][:, 3:]], axis=1)
    sc, sd_ = S.MoleculeStore(clean, dev), S.MoleculeStore(dirty, dev)
    assert (sc.n_edges == sd_.n_edges).all()
    model = _model(dev)
    other = _model(dev, cutoff_g=4.0)
    small = _model(dev, small=True)
    sc.prepare_for(model, other, small)
    sd_.prepare_for(model)
    assert len({S.size_key(m) for m in (model, other, small)}) == 3 and len(sc._counts) == 3
    idx = [39, 0, 12, 7]
    with torch.no_grad():
        ref = model(sc.collate(idx))
        out = model(sd_.collate(idx))
        model.verify()                                           # no self-loop flag, no size mismatch
        assert torch.equal(out, ref)
        b = sc.collate(idx)
        assert set(b.sizes) == {S.size_key(m) for m in (model, other, small)}
        o1, o2, o3 = model(b), other(sc.collate(idx)), small(sc.collate(idx))
        for m in (model, other, small):
            m.verify()                                           # each model found ITS sizes in the batch
        assert not torch.equal(o1, o2)


@pytest.mark.parametrize('kind', ['PDBbind', 'rna'])
def test_other_schemas_without_host_sync(dev, kind):
    """The PDBbind and RNA schemas (rows of coordinates + features, no bond list) through the same store: collation equals
    the host-side one, the forward on a batch that carries its sizes issues no synchronising call and returns bitwise what
    the plain forward returns, wrong sizes -- each of the three, either way -- are caught without a memory fault."""
    from models import Config, PAMNet
    from pamnet_amd import store as S, synth
    from pamnet_amd.graph import GraphCheckError
    if kind == 'PDBbind':
        graphs = [synth.pdbbind_complex(7, i) for i in range(6)]
        cfg = Config(dataset='PDBbind', dim=128, n_layer=1, cutoff_l=2.0, cutoff_g=6.0)
    else:
        graphs = [synth.rna_chain(7, i, n_nodes=300 + 40 * i) for i in range(6)]
        cfg = Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    torch.manual_seed(1)
    model = PAMNet(cfg).to(dev)
    st = S.MoleculeStore(graphs, dev).prepare_for(model)
    idx = [4, 1, 5, 0]
    ref_b = synth.collate([graphs[i] for i in idx])
    b0 = st.collate(idx, with_sizes=False)
    assert torch.equal(b0.x.cpu(), ref_b.x.float()) and torch.equal(b0.batch.cpu().long(), ref_b.batch)
    with torch.no_grad():
        ref = model(b0)
        model(st.collate(idx))
        torch.cuda.synchronize()
        b = st.collate(idx)
        torch.cuda.set_sync_debug_mode('error')
        try:
            out = model(b)
        finally:
            torch.cuda.set_sync_debug_mode('default')
    model.verify()
    assert torch.equal(out, ref)
    b = st.collate(idx)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        model(b).sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    model.verify()
    key = next(iter(b.sizes))
    for pos_ in range(3):
        for delta in (-5, +7):
            bad = st.collate(idx)
            v = list(bad.sizes[key])
            v[pos_] += delta
            bad.sizes = {key: tuple(v)}
            with torch.no_grad():
                model(bad)
            with pytest.raises(GraphCheckError):
                model.verify()


# ------------------------------------------------------------------------------------------------------------------------
# SURVEY 8f N2 against the REFERENCE, not against the step-by-step HIP path: resident store -> pamnet_collate_f32 ->
# pamnet_graph_plan / pamnet_graph_build_i32 (graph + basis as one engine call, QM9: the molecule-local builder) -> model,
# compared with the reference-run goldens under the same bounds as tests/test_hip_model.py.  What these calls replace:
# the reference's DataLoader collation (main_qm9.py:74-77) and models.py:104-177.
def _split(b):
    """Per-graph dicts (what a dataset holds) of a collated batch: node rows, positions, bonds with graph-local endpoints."""
    batch = b.batch.numpy()
    ng = int(batch.max()) + 1
    ptr = np.searchsorted(batch, np.arange(ng + 1))
    ei = b.edge_index.numpy() if getattr(b, 'edge_index', None) is not None else None
    out = []
    for k in range(ng):
        lo, hi = int(ptr[k]), int(ptr[k + 1])
        d = dict(x=b.x[lo:hi].numpy(), y=np.float32(b.y[k]))
        if getattr(b, 'pos', None) is not None:
            d['pos'] = b.pos[lo:hi].numpy()
        if ei is not None:
            sel = (ei[0] >= lo) & (ei[0] < hi)
            d['edge_index'] = ei[:, sel] - lo
        out.append(d)
    return out


def _through_store(model, graphs, dev, grad=False):
    """The batch of ALL `graphs` (in order) through the resident store; asserts that the engine built the graph and that the
    forward read nothing back."""
    from pamnet_amd import graph as G, store as S
    st = S.MoleculeStore(graphs, dev).prepare_for(model)
    idx = list(range(len(graphs)))
    with torch.set_grad_enabled(grad):
        model(st.collate(idx))                                   # warm every cache the first hinted call fills
    torch.cuda.synchronize()
    b = st.collate(idx)
    assert b.sizes
    torch.cuda.set_sync_debug_mode('error')
    try:
        with torch.set_grad_enabled(grad):
            out = model(b)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert isinstance(model._graph_cache, G.EngineGraph), 'the zero-host-sync engine call must be what ran'
    model.verify()
    return out, b, st


GOLDEN_INPUT_CASES = [('qm9_d32_l2', False), ('qm9s_d32_l2', True), ('pdbbind_d32_l2', False), ('qm9_ragged_d32_l2', False),
                      ('qm9s_ragged_d32_l2', True), ('pdbbind_d128_l3', False), ('qm9s_d128_l2', True), ('qm9_d128_l6', False),
                      ('qm9_basis_5x4_p6_d32_l2', False), ('qm9_basis_8x7_p4_d128_l2', False)]


@pytest.mark.parametrize('name,small', GOLDEN_INPUT_CASES)
def test_store_engine_path_vs_reference_golden(dev, golden, name, small):
    """Every reference-run fixture that carries its inputs, through store -> collate -> engine graph -> model: per-layer node
    features, pooled node values, graph outputs within the parity bound of the reference's fp64 run; integer sizes exact."""
    import models
    from oracle import pamnet_oracle as O
    from test_hip_model import _basis_args, _batch_from, _cfg_from, _ok
    from conftest import maxnorm_err
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    model = (models.PAMNet_s if small else models.PAMNet)(cfg, *_basis_args(cfg))
    model.load_state_dict(O.init_state_dict(cfg, seed=int(g['seed']), small=small), strict=True)
    model = model.to(dev)
    graphs = _split(_batch_from(g, 'cpu'))
    out, b, _ = _through_store(model, graphs, dev)
    out = out.cpu().numpy()
    ok, info = _ok(model._node_out.cpu().numpy(), g['node_out32'], g['node_out64'])
    assert ok, ('node_out', info)
    if 'x_layers64' in g.files:
        ok, info = _ok(torch.stack(list(model._x_layers)).cpu().numpy(), g['x_layers32'], g['x_layers64'])
        assert ok, ('x_layers', info)
    scale = None
    if cfg.dataset == 'PDBbind':
        scale = max(float(np.abs(g['node_out64'][g['in/batch'] == k]).sum()) for k in range(len(g['out64'])))
    ok, info = _ok(out, g['out32'], g['out64'], scale)
    assert ok, ('out', info)
    if scale is not None:
        raw, raw_floor = maxnorm_err(out, g['out64']), maxnorm_err(g['out32'], g['out64'])
        assert raw <= max(1e-5, 2 * raw_floor), (raw, raw_floor)
    gc = model._graph_cache
    assert gc.loc.m == int(g['num_edges_l']) and gc.n_pair == int(g['num_pairs'])
    if not small:
        assert gc.n_trip == int(g['num_triplets'])


BASELINE_CASES = ['baseline_qm9_b32', 'baseline_qm9_b128', 'baseline_pdbbind_b8', 'baseline_pdbbind_b32', 'baseline_rna_b8',
                  'baseline_qm9s_b128']


def _baseline_graphs(name):
    from pamnet_amd import synth
    kind, n = name.split('_')[1], int(name.rsplit('_b', 1)[1])
    if kind.startswith('qm9'):
        return [synth.qm9_molecule(0, i) for i in range(n)]
    if kind == 'pdbbind':
        return [synth.pdbbind_complex(1, i) for i in range(n)]
    return [synth.rna_chain(2, i) for i in range(n)]


@pytest.mark.parametrize('name', BASELINE_CASES)
def test_store_engine_path_at_baseline_sizes_vs_reference_runs(dev, golden, name):
    """The six runs of the REFERENCE ITSELF at the BASELINE.json batch sizes (tests/golden/gen/gen_golden.py --baseline-only),
    through the path bench.py's zero_host_sync / store_* fields time."""
    import models
    from oracle import pamnet_oracle as O
    from test_hip_model import _cfg_from, _ok
    from conftest import maxnorm_err
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    small = 'qm9s' in name
    sd = O.init_state_dict(cfg, seed=int(g['seed']), small=small)
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(g['weights_checksum'])) < 1e-6
    model = (models.PAMNet_s if small else models.PAMNet)(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    graphs = _baseline_graphs(name)
    out, b, st = _through_store(model, graphs, dev)
    assert b.x.size(0) == int(g['num_nodes']) and abs(float(b.x.double().abs().sum()) - float(g['x_checksum'])) < 1e-6
    out = out.cpu().numpy()
    ok, info_n = _ok(model._node_out.cpu().numpy(), g['node_out32'], g['node_out64'])
    assert ok, ('node_out', info_n)
    gc = model._graph_cache
    assert gc.loc.m == int(g['num_edges_l']) and gc.n_trip == int(g['num_triplets']) and gc.n_pair == int(g['num_pairs'])
    if cfg.dataset == 'PDBbind':
        batch = b.batch.cpu().numpy()
        scale = max(float(np.abs(g['node_out64'][batch == k]).sum()) for k in range(len(g['out64'])))
        ok, info = _ok(out, g['out32'], g['out64'], scale)
        assert ok, ('out', info)
        raw, raw_floor = maxnorm_err(out, g['out64']), maxnorm_err(g['out32'], g['out64'])
        assert raw <= max(1e-5, 2 * raw_floor), (raw, raw_floor)
    else:
        ok, info = _ok(out, g['out32'], g['out64'])
        assert ok, ('out', info)
    print('%s through the store: out %.2e (ref fp32 %.2e), node_out %.2e (ref fp32 %.2e)' % ((name,) + info + info_n))


@pytest.mark.parametrize('name', ['qm9_d32_l2', 'pdbbind_d32_l2', 'pdbbind_d128_l3', 'qm9_basis_5x4_p6_d32_l2',
                                  'qm9_basis_8x7_p4_d128_l2'])
def test_store_engine_path_gradients_vs_reference_golden(dev, golden, name):
    """d L1-loss / d params with the graph (and the transposed index lists of the backward) built by the engine call on a
    store batch, against the reference's fp64 autograd."""
    import models
    from oracle import pamnet_oracle as O
    from test_hip_model import _basis_args, _batch_from, _cfg_from
    from conftest import maxnorm_err
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    model = models.PAMNet(cfg, *_basis_args(cfg))
    model.load_state_dict(O.init_state_dict(cfg, seed=int(g['seed'])), strict=True)
    model = model.to(dev)
    out, b, _ = _through_store(model, _split(_batch_from(g, 'cpu')), dev, grad=True)
    model.zero_grad()
    loss = torch.nn.functional.l1_loss(out, b.y)
    loss.backward()
    assert abs(loss.item() - float(g['loss64'])) < 2e-5 * max(1.0, abs(float(g['loss64'])))
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(gn / float(g['grad_norm64']) - 1) < 1e-4
    sd = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('grad64/'):
            assert maxnorm_err(sd[k[7:]].grad.cpu().numpy(), g[k]) < 1e-4, k


def test_store_trainer_step_at_configs1_vs_reference_gradients(dev, golden):
    """Trainer.forward_backward on a STORE batch at BASELINE configs[1] (QM9 schema, d=128, L=6, B=128): loss, gradient norm,
    every parameter gradient's L2 norm and the committed full gradient tensors of the REFERENCE's fp64 autograd
    (tests/golden/baseline_qm9_b128.npz) -- and bitwise the gradients of the same step on plain tensors."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, store as S, synth, train
    from conftest import maxnorm_err
    g = golden('baseline_qm9_b128')
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    model = models.PAMNet(cfg)
    model.load_state_dict(O.init_state_dict(cfg, seed=0), strict=True)
    model = model.to(dev)
    tr = train.Trainer(model, lr=1e-4)
    st = S.MoleculeStore([synth.qm9_molecule(0, i) for i in range(128)], dev).prepare_for(model)
    tr.forward_backward(st.collate(list(range(128))))
    torch.cuda.synchronize()
    b = st.collate(list(range(128)))
    torch.cuda.set_sync_debug_mode('error')
    try:
        loss = tr.forward_backward(b)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert isinstance(model._graph_cache, G.EngineGraph) and model._one_node()
    model.verify()
    assert abs(float(loss) - float(g['loss64'])) < 2e-5 * max(1.0, abs(float(g['loss64'])))
    gn = float(torch.linalg.vector_norm(tr.fp.grad.double()))
    assert abs(gn / float(g['grad_norm64']) - 1) < 1e-4, (gn, float(g['grad_norm64']))
    grads = dict(zip(tr.fp.names, tr.fp.grad_views))
    for k, l2 in zip(g['grad_keys'].tolist(), g['grad_l2_64']):
        e = abs(float(grads[k].double().norm()) - float(l2)) / max(float(l2), 1e-300)
        assert e < 1e-4, (k, e)
    worst = 0.0
    for k in g.files:
        if k.startswith('grad64/'):
            e = maxnorm_err(grads[k[7:]].cpu().numpy(), g[k])
            assert e < 1e-4, (k, e)
            worst = max(worst, e)
    via_store = tr.fp.grad.clone()
    tr.forward_backward(synth.qm9_batch(0, 0, 128).to(dev))      # plain tensors: sizes read back, same kernels
    assert torch.equal(via_store, tr.fp.grad)
    print('Trainer.forward_backward through the store at configs[1]: worst full gradient tensor vs reference fp64 %.1e' % worst)


def test_store_engine_path_rna_checkpoint(dev, golden):
    """Shipped checkpoint + shipped RNA-Puzzles graphs (TU rows: xyz + type) as a resident store: per-graph scores against
    the reference's fp32 / fp64 runs, batched selection == each alone."""
    import models
    from test_hip_model import _ok
    g = golden('rna_native')
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    model = models.PAMNet(cfg)
    model.load_state_dict({k: torch.from_numpy(g['ckpt/' + k]) for k in g['ckpt_keys'].tolist()}, strict=True)
    model = model.to(dev).eval()
    gids = (6, 4, 17)
    graphs = [dict(x=g['g%d/x' % gid], y=np.float32(0)) for gid in gids]
    out, _, st = _through_store(model, graphs, dev)
    out = out.cpu().numpy()
    for k, gid in enumerate(gids):
        ok, info = _ok(out[k:k + 1], g['g%d/out32' % gid], g['g%d/out64' % gid])
        assert ok, (gid, info)
    with torch.no_grad():
        one = model(st.collate([1]))
    model.verify()
    gc = model._graph_cache
    assert gc.loc.m == int(g['g4/num_edges_l']) and gc.n_trip == int(g['g4/num_triplets']) and gc.n_pair == int(g['g4/num_pairs'])
    ok, info = _ok(one.cpu().numpy(), g['g4/out32'], g['g4/out64'])
    assert ok, info


@pytest.mark.parametrize('name', ['qm9_d32_l2', 'pdbbind_d32_l2', 'qm9_ragged_d32_l2', 'qm9_basis_5x4_p6_d32_l2'])
def test_store_engine_graph_indices_equal_the_reference_lists(dev, golden, name):
    """Integer parity of the ONE-call graph with the reference's own index lists (captured from its fp32 run: local edges,
    triplets (k, j, i), pairs (i, j, j'), models.py:68-98): the same edges, the same triplet / pair rows with the same
    multiplicities, rows grouped by target edge with triplets before pairs (local_message_passing.py:39)."""
    import collections
    import models
    from oracle import pamnet_oracle as O
    from test_hip_model import _basis_args, _batch_from, _cfg_from
    g = golden(name)
    cfg = _cfg_from(g, models.Config)
    model = models.PAMNet(cfg, *_basis_args(cfg))
    model.load_state_dict(O.init_state_dict(cfg, seed=int(g['seed'])), strict=True)
    model = model.to(dev)
    _through_store(model, _split(_batch_from(g, 'cpu')), dev)
    gc = model._graph_cache
    src, dst = gc.loc.col.cpu().long().numpy(), gc.loc.row_of.cpu().long().numpy()
    ref_e = g['ref/edge_index_l']
    assert sorted(zip(src.tolist(), dst.tolist())) == sorted(zip(ref_e[0].tolist(), ref_e[1].tolist()))
    e, e2, kind = gc.tp.row_of.cpu().long().numpy(), gc.tp.col.cpu().long().numpy(), gc.tp_kind.cpu().numpy()
    mine_t = collections.Counter((int(src[b]), int(src[a]), int(dst[a])) for a, b, k in zip(e, e2, kind) if k == 0)
    ref_t = collections.Counter(zip(g['ref/idx_k'].tolist(), g['ref/idx_j'].tolist(), g['ref/idx_i'].tolist()))
    assert mine_t == ref_t
    mine_p = collections.Counter((int(src[a]), int(dst[a]), int(src[b])) for a, b, k in zip(e, e2, kind) if k == 1)
    ref_p = collections.Counter(zip(g['ref/idx_i_pair'].tolist(), g['ref/idx_j1_pair'].tolist(), g['ref/idx_j2_pair'].tolist()))
    assert mine_p == ref_p
    assert (np.diff(e) >= 0).all()
    for a in np.unique(e):                                       # inside a target edge's group: triplets, then pairs
        ks = kind[e == a]
        assert (np.diff(ks) >= 0).all()

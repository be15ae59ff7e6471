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

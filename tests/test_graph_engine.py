"""Graph-construction engine (csrc/graph_engine.hip, pamnet_graph_build_i32): the whole parameter-independent front of
PAMNet.forward -- index ingestion, radius / kNN graphs, cutoff masks, CSRs, triplets / pairs, angles, transposed index
lists, spherical basis (reference models.py:62-98, 104-177, layers/basic.py:107-116) -- as ONE C call for batches whose
sizes the host knows.  Integer work: every array bit-identical to the step-by-step path (which tests/test_hip_kernels.py
pins against the oracle and the reference's star-graph golden); geometry: bit-identical floats; the model on top: bitwise
the same outputs and gradients; wrong sizes: caught by the deferred device-side check, memory-safe."""
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


def _case(kind, dev):
    from pamnet_amd import synth
    if kind == 'qm9':
        b = synth.qm9_batch(41, 0, 24)
        return b.to(dev), dict(dataset='QM9', cutoff_l=5.0, cutoff_g=5.0, flow='source_to_target', n_types=5)
    if kind == 'qm9_ragged':
        b = synth.collate([synth.qm9_molecule(3, 0),
                           dict(x=np.array([1], np.float32), pos=np.zeros((1, 3), np.float32),
                                edge_index=np.zeros((2, 0), np.int64), y=np.float32(0.1)),
                           synth.qm9_molecule(3, 1),
                           dict(x=np.array([0, 2], np.float32), pos=np.array([[0, 0, 0], [1.1, 0, 0]], np.float32),
                                edge_index=np.array([[0, 1], [1, 0]], np.int64), y=np.float32(0.2))])
        return b.to(dev), dict(dataset='QM9', cutoff_l=5.0, cutoff_g=5.0, flow='source_to_target', n_types=5)
    if kind == 'pdbbind':
        b = synth.pdbbind_batch(9, 0, 3, n_pocket=90, n_ligand=16)
        return b.to(dev), dict(dataset='PDBbind', cutoff_l=2.0, cutoff_g=6.0, flow='source_to_target', n_types=None)
    flow = 'target_to_source' if kind == 'rna_t2s' else 'source_to_target'
    b = synth.collate([synth.rna_chain(5, i, n_nodes=180 + 70 * i) for i in range(3)])
    return b.to(dev), dict(dataset='rna_x', cutoff_l=2.6, cutoff_g=20.0, flow=flow, n_types=3)


def _build(b, kw, need_grad, with_triplets, sizes=None, mol_local=None):
    from pamnet_amd import graph as G
    return G.build_graph(kw['dataset'], kw['cutoff_l'], kw['cutoff_g'], kw['flow'], b.x, b.batch, getattr(b, 'pos', None),
                         getattr(b, 'edge_index', None), num_graphs=b.num_graphs, need_grad=need_grad,
                         with_triplets=with_triplets, n_types=kw['n_types'], sizes=sizes, mol_local=mol_local)


def _transposes_are_the_counting_sorts(g):
    """The transposed lists, however a path builds them (reverse-edge index, inverse transposition, structural enumeration
    of the triplet rows), hold what the stable counting sort of the column returns -- same pointer, same rows; the kNN
    lists keep the original order inside a row, every other list is ascending (= bitwise the counting sort)."""
    from pamnet_amd import graph as G
    for csr, tr, exact in ((g.glob, g.glob_T, None), (g.loc, g.loc_T, None), (g.tp, g.tp_T, True)):
        want = G.Transpose(csr.col, tr.rows)
        assert torch.equal(want.ptr, tr.ptr)
        if exact or not isinstance(tr, G.InverseTranspose):
            assert torch.equal(want.perm, tr.perm)
        else:
            seg = torch.repeat_interleave(torch.arange(tr.rows, device=tr.ptr.device), (tr.ptr[1:] - tr.ptr[:-1]).long())
            key = seg.long() * (csr.m + 1)
            assert torch.equal(torch.sort(key + want.perm.long()).values, torch.sort(key + tr.perm.long()).values)


def _same_graph(a_g, c_g, need_grad, kw, tag):
    for name, sub in FIELDS:
        a, c = getattr(a_g, name), getattr(c_g, name)
        if sub is None:
            ok = torch.equal(a, c) if isinstance(a, torch.Tensor) else a == c
            assert ok, (tag, name)
            continue
        if name.endswith('_T') and not need_grad:
            assert a.ptr is None and c.ptr is None
            continue
        for f in sub:
            assert torch.equal(getattr(a, f), getattr(c, f)), (tag, name, f)
        if hasattr(a, 'm') and hasattr(c, 'm'):
            assert a.m == c.m and a.rows == c.rows
    assert a_g.n_trip == c_g.n_trip and a_g.n_pair == c_g.n_pair
    assert torch.equal(a_g.pos, c_g.pos)
    if kw['dataset'] == 'PDBbind':
        assert torch.equal(a_g.sign, c_g.sign)
    if kw['n_types'] is not None:
        assert torch.equal(a_g.types, c_g.types)


FIELDS = [('n', None), ('n_graphs', None), ('node_graph', None), ('gptr', None), ('dist_g', None), ('dist_l', None),
          ('tp_angle', None), ('tp_kind', None), ('glob', ('ptr', 'row_of', 'col')), ('loc', ('ptr', 'row_of', 'col')),
          ('tp', ('ptr', 'row_of', 'col')), ('glob_T', ('ptr', 'perm')), ('loc_T', ('ptr', 'perm')), ('tp_T', ('ptr', 'perm'))]


@pytest.mark.parametrize('kind', ['qm9', 'qm9_ragged', 'pdbbind', 'rna_t2s', 'rna_s2t'])
@pytest.mark.parametrize('need_grad', [True, False])
@pytest.mark.parametrize('with_triplets', [True, False])
def test_engine_graph_equals_step_by_step_graph(dev, kind, need_grad, with_triplets):
    from pamnet_amd import graph as G
    if not with_triplets and not kind.startswith('qm9'):
        pytest.skip('PAMNet_s is QM9 only (models.py:286)')
    b, kw = _case(kind, dev)
    ref = _build(b, kw, need_grad, with_triplets)                       # sizes read back from the device
    ref_sbf = G.spherical_basis(ref, kw['cutoff_l'])
    sizes = (ref.glob.m, ref.loc.m, ref.tp.m)
    if need_grad:
        _transposes_are_the_counting_sorts(ref)
    saved = G.ENGINE
    try:
        G.ENGINE = False
        old = _build(b, kw, need_grad, with_triplets, sizes)            # zero-sync, step by step (Python orchestration)
        G.ENGINE = True
        eng = _build(b, kw, need_grad, with_triplets, sizes)            # zero-sync, one engine call
    finally:
        G.ENGINE = saved
    assert isinstance(eng, G.EngineGraph) and not isinstance(old, G.EngineGraph)
    torch.cuda.synchronize()
    G.raise_for_flag(G.read_flags([eng.check]))
    for other in (ref, old):
        _same_graph(eng, other, need_grad, kw, kind)
    assert torch.equal(G.spherical_basis(eng, kw['cutoff_l']), ref_sbf)


@pytest.mark.parametrize('kind', ['qm9', 'pdbbind', 'rna_t2s'])
def test_model_on_engine_graph_is_bitwise_the_model_on_the_step_by_step_graph(dev, kind):
    """Forward outputs and every parameter gradient: the engine-built graph against the step-by-step one."""
    import models
    from pamnet_amd import graph as G
    b, kw = _case(kind, dev)
    torch.manual_seed(5)
    dim, L = (16, 1) if kind.startswith('rna') else (128, 2)
    model = models.PAMNet(models.Config(dataset=kw['dataset'], dim=dim, n_layer=L, cutoff_l=kw['cutoff_l'],
                                        cutoff_g=kw['cutoff_g'], flow=kw['flow'])).to(dev)
    ref = _build(b, kw, True, True)
    b.sizes = (ref.glob.m, ref.loc.m, ref.tp.m)
    res = []
    saved = G.ENGINE
    try:
        for on in (False, True):
            G.ENGINE = on
            model.zero_grad()
            out = model(b)
            assert isinstance(model._graph_cache, G.EngineGraph) == on
            (out * torch.arange(1, out.numel() + 1, device=dev)).sum().backward()
            model.verify()
            res.append((out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    finally:
        G.ENGINE = saved
    assert torch.equal(res[0][0], res[1][0])
    assert len(res[0][1]) == len(res[1][1]) and all(torch.equal(a, c) for a, c in zip(res[0][1], res[1][1]))


@pytest.mark.parametrize('kind', ['qm9', 'pdbbind', 'rna_t2s'])
def test_engine_catches_wrong_sizes_without_a_memory_fault(dev, kind):
    """Each of the sizes too small / too large: the call runs to completion inside its arena and the flag word names the
    mismatch; valid sizes leave it clean.  (QM9: the bond count is an input size, a mismatch there falls back to the
    step-by-step path, which flags it the same way.)"""
    from pamnet_amd import graph as G
    b, kw = _case(kind, dev)
    ref = _build(b, kw, True, True)
    good = [ref.glob.m, ref.loc.m, ref.tp.m]
    for pos_ in range(3):
        for delta in (-7, +9):
            sz = list(good)
            sz[pos_] += delta
            g = _build(b, kw, True, True, tuple(sz))
            if kind == 'qm9' and pos_ == 1:
                assert not isinstance(g, G.EngineGraph)
            else:
                assert isinstance(g, G.EngineGraph)
            torch.cuda.synchronize()
            bits = G.read_flags([g.check])
            assert bits & (2 << pos_), (kind, pos_, delta, bits)
            with pytest.raises(G.GraphCheckError):
                G.raise_for_flag(bits)
    g = _build(b, kw, True, True, tuple(good))
    torch.cuda.synchronize()
    assert G.read_flags([g.check]) == 0
    # invalid index inputs surface through the same word (bit 1), as the reference's IndexError
    if kind != 'pdbbind':
        import copy
        bad = copy.copy(b)
        bad.x = b.x.clone()
        if kind == 'qm9':
            bad.x[3] = 9.0
        else:
            bad.x[3, -1] = 5.0
        g = _build(bad, kw, True, True, tuple(good))
        torch.cuda.synchronize()
        with pytest.raises(IndexError):
            G.raise_for_flag(G.read_flags([g.check]))


def test_engine_forward_is_a_handful_of_host_calls(dev):
    """N2: with the sizes known, graph construction + basis is ONE library call (plus its plan), the layer stack ONE, and
    no tensor op of the calling convention remains in between: count the C-ABI calls of a whole RNA forward."""
    import models
    from pamnet_amd import lib, store as S, synth
    graphs = [synth.rna_chain(5, i, n_nodes=200 + 50 * i) for i in range(4)]
    torch.manual_seed(1)
    model = models.PAMNet(models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0,
                                        flow='target_to_source')).to(dev)
    st = S.MoleculeStore(graphs, dev).prepare_for(model)
    with torch.no_grad():
        model(st.collate([0, 1, 2, 3]))
        calls = []
        orig = lib.call
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            b = st.collate([0, 1, 2, 3])
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode('error')
            try:
                model(b)
            finally:
                torch.cuda.set_sync_debug_mode('default')
        finally:
            lib.call = orig
    model.verify()
    graph_calls = [c for c in calls if c.startswith('pamnet_graph_')]
    assert graph_calls == ['pamnet_graph_plan', 'pamnet_graph_build_i32'], calls
    assert len(calls) <= 12, calls                    # collate, plan, build, 3 embeddings, workspace, stack, fuse+pool, ...


# ---- the molecule-local builder (csrc/graph_mol.hip): QM9 schema, one wavefront per molecule, two launches -----------------
def _qm9_case(name, dev):
    from pamnet_amd import synth
    kw = dict(dataset='QM9', cutoff_l=5.0, cutoff_g=5.0, flow='source_to_target', n_types=5)
    if name == 'batch128':
        return synth.qm9_batch(0, 0, 128).to(dev), kw
    if name == 'ragged':
        return _case('qm9_ragged', dev)
    if name == 'tight_cutoff':                      # a global cutoff that leaves atoms without neighbours
        kw = dict(kw, cutoff_g=1.2)
        return synth.qm9_batch(7, 0, 16).to(dev), kw
    if name == 'asymmetric_bonds':                  # one direction of some bonds only, duplicates: user-supplied lists are arbitrary
        b = synth.qm9_batch(3, 0, 6)
        ei = b.edge_index
        keep = torch.ones(ei.size(1), dtype=torch.bool)
        keep[::5] = False
        ei = torch.cat([ei[:, keep], ei[:, :3]], 1)
        order = torch.argsort(b.batch[ei[1]], stable=True)          # grouped by molecule, any order inside
        b.edge_index = ei[:, order].contiguous()
        return b.to(dev), kw
    raise KeyError(name)


@pytest.mark.parametrize('name', ['batch128', 'ragged', 'tight_cutoff', 'asymmetric_bonds'])
@pytest.mark.parametrize('need_grad', [True, False])
@pytest.mark.parametrize('with_triplets', [True, False])
def test_molecule_local_builder_equals_step_by_step_graph(dev, name, need_grad, with_triplets):
    """Every index / geometry array bit for bit, plain tensors (one host round trip) and engine (none)."""
    from pamnet_amd import graph as G
    b, kw = _qm9_case(name, dev)
    saved = G.MOL_LOCAL
    try:
        G.MOL_LOCAL = False
        ref = _build(b, kw, need_grad, with_triplets)                       # step-by-step launches
        sizes = (ref.glob.m, ref.loc.m, ref.tp.m)
        if need_grad:
            _transposes_are_the_counting_sorts(ref)
        G.MOL_LOCAL = True
        calls = []
        from pamnet_amd import lib
        orig = lib.call
        lib.call = lambda nm, *a: (calls.append(nm), orig(nm, *a))[1]
        try:
            plain = _build(b, kw, need_grad, with_triplets)                 # count launch, round trip, fill launch
            eng = _build(b, kw, need_grad, with_triplets, sizes, mol_local=True)
        finally:
            lib.call = orig
    finally:
        G.MOL_LOCAL = saved
    assert calls.count('pamnet_mol_graph_fill_i32') == 1 and 'pamnet_radius_fill_i32' not in calls, calls
    assert isinstance(eng, G.EngineGraph) and not isinstance(plain, G.EngineGraph)
    torch.cuda.synchronize()
    G.raise_for_flag(G.read_flags([eng.check]))
    _same_graph(plain, ref, need_grad, kw, name + ':plain')
    _same_graph(eng, ref, need_grad, kw, name + ':engine')
    assert torch.equal(G.spherical_basis(eng, kw['cutoff_l']), G.spherical_basis(ref, kw['cutoff_l']))


def test_molecule_local_builder_declines_what_it_cannot_hold(dev):
    """Plain tensors: bonds not grouped by molecule, a bond across molecules, a molecule over 64 atoms -> the step-by-step
    launches build the graph (same arrays as with the builder switched off).  Engine with the caller's word for it: the
    batch is flagged (local-edge mismatch), nothing is written out of bounds."""
    from pamnet_amd import graph as G, synth
    kw = dict(dataset='QM9', cutoff_l=5.0, cutoff_g=5.0, flow='source_to_target', n_types=5)
    base = synth.qm9_batch(11, 0, 12)
    cases = {}
    sh = synth.qm9_batch(11, 0, 12)
    perm = torch.randperm(sh.edge_index.size(1), generator=torch.Generator().manual_seed(0))
    sh.edge_index = sh.edge_index[:, perm].contiguous()
    cases['shuffled'] = sh
    big = synth.collate([dict(x=np.zeros(70, np.float32), pos=np.random.RandomState(0).rand(70, 3).astype(np.float32) * 6,
                              edge_index=np.stack([np.arange(69), np.arange(1, 70)]).astype(np.int64),
                              y=np.float32(0.0)), synth.qm9_molecule(11, 1)])
    big.edge_index = torch.cat([big.edge_index, big.edge_index.flip(0)], 1)
    order = torch.argsort(big.batch[big.edge_index[1]], stable=True)
    big.edge_index = big.edge_index[:, order].contiguous()
    cases['big'] = big
    for name, b in cases.items():
        b = b.to(dev)
        saved = G.MOL_LOCAL
        try:
            G.MOL_LOCAL = False
            ref = _build(b, kw, True, True)
            G.MOL_LOCAL = True
            got = _build(b, kw, True, True)
            _same_graph(got, ref, True, kw, name)
            eng = _build(b, kw, True, True, (ref.glob.m, ref.loc.m, ref.tp.m), mol_local=True)
        finally:
            G.MOL_LOCAL = saved
        torch.cuda.synchronize()
        bits = G.read_flags([eng.check])
        assert bits & 4, (name, bits)
        with pytest.raises(G.GraphCheckError):
            G.raise_for_flag(bits)
    del base


def test_store_batches_take_the_molecule_local_builder(dev):
    """A QM9-schema store vouches for its molecules: the engine call of its batches runs the two-launch builder, the model's
    outputs and gradients are bitwise those of the step-by-step graph."""
    import models
    from pamnet_amd import graph as G, store as S, synth
    graphs = [synth.qm9_molecule(2, i) for i in range(48)]
    torch.manual_seed(3)
    model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
    st = S.MoleculeStore(graphs, dev).prepare_for(model)
    assert st._mol_local[S.size_key(model)] is True
    res = []
    saved = G.MOL_LOCAL
    try:
        for on in (False, True):
            G.MOL_LOCAL = on
            b = st.collate(list(range(5, 37)))
            model.zero_grad()
            out = model(b)
            (out * torch.arange(1, out.numel() + 1, device=dev)).sum().backward()
            model.verify()
            res.append((out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    finally:
        G.MOL_LOCAL = saved
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, c) for a, c in zip(res[0][1], res[1][1]))


@pytest.mark.parametrize('seed', list(range(8)))
def test_molecule_local_builder_random_molecules(dev, seed):
    """Seeded random batches up to the builder's limits: 1-64 atoms, random directed bond lists (duplicates, one-directional
    bonds, atoms without bonds), clustered positions (coincident atoms included), a cutoff drawn per batch."""
    from pamnet_amd import graph as G, synth
    rng = np.random.default_rng(100 + seed)
    mols = []
    for _ in range(int(rng.integers(1, 24))):
        na = int(rng.integers(1, 65)) if rng.random() < 0.8 else 64
        pos = (rng.normal(size=(na, 3)) * rng.uniform(0.5, 3.0)).astype(np.float32)
        if na > 2 and rng.random() < 0.3:
            pos[1] = pos[0]                                   # coincident atoms: zero distances, ties
        nb = int(rng.integers(0, min(256, 4 * na) + 1)) if na > 1 else 0
        s = rng.integers(0, na, nb)
        d = rng.integers(0, na, nb)
        keep = s != d
        ei = np.stack([s[keep], d[keep]]).astype(np.int64)
        mols.append(dict(x=rng.integers(0, 5, na).astype(np.float32), pos=pos, edge_index=ei, y=np.float32(0.0)))
    b = synth.collate(mols).to(dev)
    kw = dict(dataset='QM9', cutoff_l=5.0, cutoff_g=float(rng.uniform(0.8, 6.0)), flow='source_to_target', n_types=5)
    if b.edge_index.size(1) == 0:
        pytest.skip('no bonds drawn')
    saved = G.MOL_LOCAL
    try:
        G.MOL_LOCAL = False
        ref = _build(b, kw, True, True)
        G.MOL_LOCAL = True
        got = _build(b, kw, True, True, mol_local=True)
        if ref.glob.m > 0 and ref.tp.m > 0:
            eng = _build(b, kw, True, True, (ref.glob.m, ref.loc.m, ref.tp.m), mol_local=True)
        else:
            eng = None
    finally:
        G.MOL_LOCAL = saved
    _same_graph(got, ref, True, kw, 'random:plain')
    if eng is not None:
        torch.cuda.synchronize()
        G.raise_for_flag(G.read_flags([eng.check]))
        _same_graph(eng, ref, True, kw, 'random:engine')

"""Synthetic inputs with the reference datasets' *schema* (numpy only, deterministic per graph index).

No QM9 / PDBbind files exist offline, so benchmarks and parity tests run on generated graphs that
carry exactly the fields the reference datasets yield:

  QM9      (datasets/qm9_dataset.py:122,243,251):  x [N] float atom-type index in {0..4} (H,C,N,O,F), pos [N,3] float32,
           edge_index [2,E] int64 -- bonds in both directions sorted by row*N+col --, y [1] float32, batch [N] int64.
  PDBbind  (preprocess_pdbbind.py:126-139, utils/featurizer.py): x [N,21] = xyz + 18 features; the complex, the
           pocket (+100 A in x) and the ligand (+200 A in x) are three disconnected copies in one graph.

A molecule's content depends only on (seed, graph index), so any rank can regenerate any shard.
"""
import numpy as np

_VALENCE = {1: 4, 2: 3, 3: 2, 4: 1}            # type index -> valence (C,N,O,F); 0 = H


def _unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def _grow(rng, pos, anchor_pos, bond_lo, bond_hi, min_sep, others, tries=40):
    """A point at bond length from `anchor_pos` that keeps `min_sep` from every point in `others`."""
    for _ in range(tries):
        p = anchor_pos + _unit(rng) * rng.uniform(bond_lo, bond_hi)
        if others.shape[0] == 0 or np.min(np.linalg.norm(others - p, axis=1)) >= min_sep:
            return p
    return None


def qm9_molecule(seed, index):
    """One QM9-schema molecule: 7-9 heavy atoms grown as a random tree + hydrogens filling free valences."""
    rng = np.random.default_rng([seed, index])
    n_heavy = int(rng.integers(7, 10))
    types = [1]
    pos = [np.zeros(3)]
    free = [_VALENCE[types[0]]]
    bonds = []
    while len(pos) < n_heavy:
        cand = [a for a in range(len(pos)) if free[a] > 0]
        if not cand:
            break
        # favour recently added atoms -> extended, chain-like skeletons (realistic 5 A neighbour counts)
        w = np.array([1.0 + 2.0 * (a >= len(pos) - 2) for a in cand])
        a = int(rng.choice(cand, p=w / w.sum()))
        arr = np.array(pos)
        others = np.delete(arr, a, axis=0)
        p = _grow(rng, arr, arr[a], 1.30, 1.55, 2.15, others)
        if p is None:
            free[a] = 0
            continue
        t = int(rng.choice([1, 2, 3, 4], p=[0.74, 0.11, 0.14, 0.01]))
        bonds.append((a, len(pos)))
        pos.append(p)
        types.append(t)
        free[a] -= 1
        free.append(_VALENCE[t] - 1)
    # hydrogens
    n_h_target = int(rng.integers(8, 13))
    order = [a for a in range(len(pos)) for _ in range(free[a])]
    rng.shuffle(order)
    n_h = 0
    for a in order:
        if n_h >= n_h_target:
            break
        arr = np.array(pos)
        others = np.delete(arr, a, axis=0)
        p = _grow(rng, arr, arr[a], 1.07, 1.11, 1.55, others)
        if p is None:
            continue
        bonds.append((a, len(pos)))
        pos.append(p)
        types.append(0)
        n_h += 1
    n = len(pos)
    pos = np.asarray(pos, dtype=np.float64)
    pos -= pos.mean(0)
    pos = pos @ _random_rotation(rng)
    b = np.asarray(bonds, dtype=np.int64)
    row = np.concatenate([b[:, 0], b[:, 1]])
    col = np.concatenate([b[:, 1], b[:, 0]])
    perm = np.argsort(row * n + col, kind='stable')                 # datasets/qm9_dataset.py:243
    return dict(x=np.asarray(types, dtype=np.float32), pos=pos.astype(np.float32),
                edge_index=np.stack([row[perm], col[perm]]), y=np.float32(rng.normal()))


def _random_rotation(rng):
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _tree_cloud(rng, n, bond_lo=1.30, bond_hi=1.55, min_sep=2.05, max_deg=4):
    """n points grown as a random bonded tree with a non-bonded exclusion radius (protein-like packing)."""
    pos = np.zeros((n, 3))
    deg = np.zeros(n, dtype=np.int64)
    k = 1
    while k < n:
        cand = np.nonzero(deg[:k] < max_deg)[0]
        a = int(cand[rng.integers(max(0, cand.size - 24), cand.size)]) if rng.random() < 0.7 else int(rng.choice(cand))
        others = np.delete(pos[:k], a, axis=0)
        p = _grow(rng, pos[:k], pos[a], bond_lo, bond_hi, min_sep, others, tries=12)
        if p is None:
            deg[a] = max_deg
            if not np.any(deg[:k] < max_deg):
                deg[:k] = 0
            continue
        pos[k] = p
        deg[a] += 1
        deg[k] = 1
        k += 1
    return pos


def _pdb_features(rng, n):
    """18 features per atom (utils/featurizer.py:62-72,98-99,124-132): 9-way one-hot + 4 numeric + 5 binary."""
    f = np.zeros((n, 18), dtype=np.float32)
    f[np.arange(n), rng.integers(0, 9, size=n)] = 1.0
    f[:, 9] = rng.integers(1, 4, size=n)             # hybridisation
    f[:, 10] = rng.integers(1, 5, size=n)            # heavy valence
    f[:, 11] = rng.integers(0, 3, size=n)            # hetero valence
    f[:, 12] = rng.normal(scale=0.3, size=n)         # partial charge
    f[:, 13:] = rng.integers(0, 2, size=(n, 5))
    return f


def pdbbind_complex(seed, index, n_pocket=None, n_ligand=None):
    """One PDBbind-schema graph: [complex | pocket +100 A | ligand +200 A], x [N,21]."""
    rng = np.random.default_rng([seed, 7919, index])
    n_pocket = int(rng.integers(150, 351)) if n_pocket is None else n_pocket
    n_ligand = int(rng.integers(10, 71)) if n_ligand is None else n_ligand
    cloud = _tree_cloud(rng, n_pocket + n_ligand)
    cloud -= cloud.mean(0)
    # ligand = the n_ligand atoms nearest the centroid (a bound pose inside the pocket)
    order = np.argsort(np.linalg.norm(cloud, axis=1))
    lig, poc = cloud[order[:n_ligand]], cloud[order[n_ligand:]]
    f_lig, f_poc = _pdb_features(rng, n_ligand), _pdb_features(rng, n_pocket)
    f_lig[:, 17], f_poc[:, 17] = 1.0, -1.0           # molcode column
    shift = lambda p, dx: p + np.array([dx, 0.0, 0.0])
    pos = np.concatenate([lig, poc, shift(poc, 100.0), shift(lig, 200.0)]).astype(np.float32)
    feat = np.concatenate([f_lig, f_poc, f_poc, f_lig])
    return dict(x=np.concatenate([pos, feat], axis=1).astype(np.float32), y=np.float32(rng.normal(loc=6.0, scale=2.0)))


def rna_chain(seed, index, n_nodes=None):
    """RNA-schema graph (datasets/tu_dataset.py:104-122): x [N,4] = xyz + label in {0,1,2} (C,N,O)."""
    rng = np.random.default_rng([seed, 104729, index])
    n = int(rng.integers(800, 3900)) if n_nodes is None else n_nodes
    pos = _tree_cloud(rng, n, 1.33, 1.52, 2.2, max_deg=3)
    lab = rng.choice([0, 1, 2], size=n, p=[0.55, 0.2, 0.25]).astype(np.float32)
    return dict(x=np.concatenate([np.round(pos, 3), lab[:, None]], axis=1).astype(np.float32),
                y=np.float32(rng.uniform(0, 20)))


class Batch(object):
    """Duck-typed stand-in for PyG's Batch: the attributes PAMNet.forward reads (models.py:101-106)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device, non_blocking=False):
        """non_blocking=True: asynchronous copies on the current stream; the returned batch carries `inputs_ready`, an
        event recorded behind them, which the trainer's side-stream graph construction waits for (train.Trainer.prefetch)."""
        import torch
        b = Batch(**{k: (v.to(device, non_blocking=non_blocking) if isinstance(v, torch.Tensor) else v)
                     for k, v in self.__dict__.items() if k != 'inputs_ready'})
        if non_blocking and torch.device(device).type == 'cuda':
            b.inputs_ready = torch.cuda.Event()
            b.inputs_ready.record(torch.cuda.current_stream(torch.device(device)))
        return b

    def keys(self):
        return list(self.__dict__.keys())


def collate(graphs):
    """Concatenate per-graph dicts into one batch (what PyG's Batch.from_data_list does for these fields):
    node tensors concatenated, edge_index offset by the running node count, `batch` = graph id per node."""
    import torch
    xs, poss, eis, ys, bs = [], [], [], [], []
    off = 0
    for g, d in enumerate(graphs):
        n = d['x'].shape[0]
        xs.append(d['x'])
        bs.append(np.full(n, g, dtype=np.int64))
        ys.append(d['y'])
        if 'pos' in d:
            poss.append(d['pos'])
        if 'edge_index' in d:
            eis.append(d['edge_index'] + off)
        off += n
    kw = dict(x=torch.from_numpy(np.concatenate(xs)), batch=torch.from_numpy(np.concatenate(bs)),
              y=torch.from_numpy(np.asarray(ys, dtype=np.float32)), num_graphs=len(graphs))
    if poss:
        kw['pos'] = torch.from_numpy(np.concatenate(poss))
    if eis:
        kw['edge_index'] = torch.from_numpy(np.concatenate(eis, axis=1))
    return Batch(**kw)


def qm9_batch(seed, start, count, target=None):
    """target: None -> the molecule's scalar label; 0..11 -> that column of the synthetic 16-column label table with the
    reference's column map (main_qm9.py:60-66: targets 7-10 read columns 12-15)."""
    b = collate([qm9_molecule(seed, start + i) for i in range(count)])
    if target is not None:
        import torch
        col = target + 5 if target in (7, 8, 9, 10) else target
        b.y = torch.from_numpy(qm9_label_table(seed, start, count)[:, col].copy())
    return b


def qm9_label_table(seed, start, count):
    """[count, 16] synthetic label table (QM9's `data.y` has 16+ columns, qm9_dataset.py; values ~N(0,1) per molecule,
    reproducible from the molecule index alone so that shards agree across ranks)."""
    return np.stack([np.random.default_rng([seed, 7919, start + i]).standard_normal(16).astype(np.float32)
                     for i in range(count)])


def pdbbind_batch(seed, start, count, **kw):
    return collate([pdbbind_complex(seed, start + i, **kw) for i in range(count)])


def rna_batch(seed, start, count, **kw):
    return collate([rna_chain(seed, start + i, **kw) for i in range(count)])


def ragged_qm9_batch(seed=21):
    """Degenerate / ragged molecules in one QM9-schema batch (edge cases of the graph code, models.py:62-98,104-113):
    a single atom (no edges of either kind), a single bond (no triplets; self-pairs only), three atoms farther apart than
    any cutoff, a 3-chain (the smallest graph with triplets), between two ordinary molecules."""
    def mol(x, pos, bonds, y):
        ei = np.array([[a, b] for a, b in bonds] + [[b, a] for a, b in bonds], dtype=np.int64).reshape(-1, 2).T
        if ei.size:
            ei = ei[:, np.lexsort((ei[1], ei[0]))]
        return dict(x=np.asarray(x, np.float32), pos=np.asarray(pos, np.float32), edge_index=ei.reshape(2, -1),
                    y=np.float32(y))
    return collate([mol([1], [[0, 0, 0]], [], 0.3),
                    mol([0, 2], [[0, 0, 0], [1.1, 0, 0]], [(0, 1)], -0.2),
                    qm9_molecule(seed, 0),
                    mol([3, 3, 3], [[0, 0, 0], [9, 0, 0], [0, 9, 0]], [], 1.0),
                    mol([0, 1, 4], [[0, 0, 0], [1.2, 0, 0], [2.0, 0.9, 0]], [(0, 1), (1, 2)], 0.5),
                    qm9_molecule(seed, 1)])

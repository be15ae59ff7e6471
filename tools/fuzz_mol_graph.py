"""Seeded fuzzing of the molecule-local graph builder (csrc/graph_mol.hip) against the step-by-step launches: 300 random
batches (1-64 atoms, arbitrary directed bond lists, coincident atoms, lattice positions, random cutoff / layer kind /
transposes), plain-tensor path and engine, every array bit for bit.  usage (GPU box): python tools/fuzz_mol_graph.py"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p_)
import numpy as np, torch
import test_graph_engine as T
from pamnet_amd import graph as G, synth, lib
lib.load()
dev = torch.device('cuda:0')
bad = done = 0
for seed in range(300):
    rng = np.random.default_rng(1000 + seed)
    mols = []
    for _ in range(int(rng.integers(1, 40))):
        na = int(rng.integers(1, 65))
        pos = (rng.normal(size=(na, 3)) * rng.uniform(0.3, 3.0)).astype(np.float32)
        if na > 3 and rng.random() < 0.3:
            pos[2] = pos[0]; pos[1] = pos[0]
        if rng.random() < 0.2:
            pos = np.round(pos)                      # lattice: many equal distances
        nb = int(rng.integers(0, min(256, 5 * na) + 1)) if na > 1 else 0
        s = rng.integers(0, na, nb); d = rng.integers(0, na, nb)
        keep = s != d
        mols.append(dict(x=rng.integers(0, 5, na).astype(np.float32), pos=pos, edge_index=np.stack([s[keep], d[keep]]).astype(np.int64), y=np.float32(0)))
    b = synth.collate(mols).to(dev)
    if b.edge_index.size(1) == 0: continue
    kw = dict(dataset='QM9', cutoff_l=5.0, cutoff_g=float(rng.uniform(0.5, 6.0)), flow='source_to_target', n_types=5)
    wt = bool(rng.integers(0, 2)); ng_ = bool(rng.integers(0, 2))
    try:
        G.MOL_LOCAL = False
        ref = T._build(b, kw, ng_, wt)
        G.MOL_LOCAL = True
        got = T._build(b, kw, ng_, wt, mol_local=True)
        T._same_graph(got, ref, ng_, kw, 'fuzz')
        done += 1
        if ref.glob.m > 0 and ref.tp.m > 0:
            eng = T._build(b, kw, ng_, wt, (ref.glob.m, ref.loc.m, ref.tp.m), mol_local=True)
            torch.cuda.synchronize(); G.raise_for_flag(G.read_flags([eng.check]))
            T._same_graph(eng, ref, ng_, kw, 'fuzz-eng')
    except Exception as e:
        bad += 1; print('seed', seed, type(e).__name__, str(e)[:200])
print('compared', done, 'batches; failures:', bad)

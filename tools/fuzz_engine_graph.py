"""Seeded fuzzing of the RNA-schema graph construction: engine (one C call: fused kNN cut, structural triplet transposition,
counting sorts with arrival order) against the step-by-step launches with and without host-side sizes -- random point clouds
incl. lattices (many equal distances: the tie rules) and coincident nodes, both flows.  usage (GPU box): python tools/fuzz_engine_graph.py"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p_)
import numpy as np
import torch
import test_graph_engine as T
from pamnet_amd import graph as G, synth, lib
lib.load()
dev = torch.device('cuda:0')
bad = done = 0
for seed in range(120):
    rng = np.random.default_rng(5000 + seed)
    graphs = []
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.integers(3, 400))
        pos = rng.normal(size=(n, 3)) * rng.uniform(2.0, 12.0)
        if rng.random() < 0.4:
            pos = np.round(pos / 1.5) * 1.5                   # lattice: ties
        if n > 5 and rng.random() < 0.3:
            pos[3] = pos[1]
        x = np.concatenate([pos, rng.integers(0, 3, (n, 1))], 1).astype(np.float32)
        graphs.append(dict(x=x, y=np.float32(0)))
    b = synth.collate(graphs).to(dev)
    flow = 'target_to_source' if rng.random() < 0.6 else 'source_to_target'
    kw = dict(dataset='rna_x', cutoff_l=float(rng.uniform(1.5, 4.0)), cutoff_g=float(rng.uniform(5.0, 25.0)), flow=flow, n_types=3)
    ng_ = bool(rng.integers(0, 2))
    try:
        ref = T._build(b, kw, ng_, True)
        if min(ref.glob.m, ref.loc.m, ref.tp.m) < 1:
            continue
        if ng_:
            T._transposes_are_the_counting_sorts(ref)
        sizes = (ref.glob.m, ref.loc.m, ref.tp.m)
        G.ENGINE = False
        old = T._build(b, kw, ng_, True, sizes)
        G.ENGINE = True
        eng = T._build(b, kw, ng_, True, sizes)
        assert isinstance(eng, G.EngineGraph)
        torch.cuda.synchronize()
        G.raise_for_flag(G.read_flags([eng.check]))
        T._same_graph(eng, ref, ng_, kw, 'ref')
        T._same_graph(eng, old, ng_, kw, 'old')
        done += 1
    except Exception as e:
        bad += 1
        print('seed', seed, type(e).__name__, str(e)[:300])
    finally:
        G.ENGINE = True
print('RNA schema: compared', done, 'batches; failures:', bad)

# PDBbind schema: radius graphs at both cutoffs, reverse-edge transposes
bad = done = 0
for seed in range(60):
    rng = np.random.default_rng(9000 + seed)
    graphs = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(2, 500))
        pos = rng.normal(size=(n, 3)) * rng.uniform(2.0, 8.0) + 38.0      # around the pocket / ligand sign threshold (x > 40)
        if rng.random() < 0.4:
            pos = np.round(pos)
        x = np.concatenate([pos, rng.normal(size=(n, 18))], 1).astype(np.float32)
        graphs.append(dict(x=x, y=np.float32(0)))
    b = synth.collate(graphs).to(dev)
    kw = dict(dataset='PDBbind', cutoff_l=float(rng.uniform(1.0, 2.5)), cutoff_g=float(rng.uniform(3.0, 7.0)),
              flow='source_to_target', n_types=None)
    ng_ = bool(rng.integers(0, 2))
    try:
        ref = T._build(b, kw, ng_, True)
        if min(ref.glob.m, ref.loc.m, ref.tp.m) < 1:
            continue
        if ng_:
            T._transposes_are_the_counting_sorts(ref)
        sizes = (ref.glob.m, ref.loc.m, ref.tp.m)
        G.ENGINE = False
        old = T._build(b, kw, ng_, True, sizes)
        G.ENGINE = True
        eng = T._build(b, kw, ng_, True, sizes)
        assert isinstance(eng, G.EngineGraph)
        torch.cuda.synchronize()
        G.raise_for_flag(G.read_flags([eng.check]))
        T._same_graph(eng, ref, ng_, kw, 'ref')
        T._same_graph(eng, old, ng_, kw, 'old')
        done += 1
    except Exception as e:
        bad += 1
        print('seed', seed, type(e).__name__, str(e)[:300])
    finally:
        G.ENGINE = True
print('PDBbind schema: compared', done, 'batches; failures:', bad)

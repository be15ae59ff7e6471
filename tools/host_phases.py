#!/usr/bin/env python
"""Host enqueue time of a training step / a forward by phase (wrapper timers, no synchronisation added), through the
zero-host-sync store path.  Usage on the GPU box:  python tools/host_phases.py [qm9|rna|pdbbind]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import fused, graph as G, narrow, ops, store as S, synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'rna'
dev = torch.device('cuda:0')
torch.manual_seed(0)
if kind == 'qm9':
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    graphs = [synth.qm9_molecule(0, i) for i in range(512)]
    idx = [list(range(128 * k, 128 * k + 128)) for k in range(4)]
elif kind == 'pdbbind':
    cfg = models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    graphs = [synth.pdbbind_complex(1, i) for i in range(64)]
    idx = [list(range(32 * (k % 2), 32 * (k % 2) + 32)) for k in range(4)]
else:
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    graphs = [synth.rna_chain(2, i) for i in range(8)]
    idx = [[(i + 2 * k) % 8 for i in range(8)] for k in range(4)]
model = models.PAMNet(cfg).to(dev)
st = S.MoleculeStore(graphs, dev).prepare_for(model)
tr = Trainer(model, lr=1e-4)

acc = {}
T = time.perf_counter


def wrap(obj, name, tag):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = T()
        try:
            return fn(*a, **k)
        finally:
            acc[tag] = acc.get(tag, 0.0) + (T() - t0)
    setattr(obj, name, w)


wrap(st, 'collate', 'collate')
wrap(G, 'build_graph', 'graph')
wrap(G, 'spherical_basis', 'sbf')
wrap(model, '_input_stage', 'input_stage')
wrap(model, '_embed', 'embed_x')
wrap(model, '_edge_embeddings', 'edge_embed')
wrap(fused, 'layer_stack', 'stack_fwd')
wrap(narrow, 'layer_stack', 'stack_fwd')
wrap(ops, 'fuse_pool', 'fuse_pool')
wrap(ops, 'l1_loss_with_grad', 'loss')
wrap(ops, 'backward_whole', 'backward')
wrap(tr, 'native_update', 'update')
wrap(tr, '_throttle', 'throttle')
wrap(tr, 'prefetch', 'prefetch(total)')


plain = len(sys.argv) > 2 and sys.argv[2] == 'plain'          # plain tensors (the reference calling convention), resident
if plain:
    plain_batches = [synth.collate([graphs[j] for j in ix]).to(dev) for ix in idx]


def steps(n):
    if plain:
        for i in range(n):
            tr.step(plain_batches[i % 4], next_data=plain_batches[(i + 1) % 4])
        return
    nxt = st.collate(idx[0])
    for i in range(n):
        cur, nxt = nxt, st.collate(idx[(i + 1) % 4])
        tr.step(cur, next_data=nxt)


steps(8)
torch.cuda.synchronize()
acc.clear()
n = 50
t0 = T()
steps(n)
host = (T() - t0) / n * 1e3
torch.cuda.synchronize()
wall = (T() - t0) / n * 1e3
print('%s training step %s: host loop %.3f ms/step, wall %.3f ms/step' % (kind, 'on plain tensors' if plain else 'through the store', host, wall))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('   %-18s %7.3f ms/step' % (k, v / n * 1e3))
print('   (graph / sbf / the first collate are inside prefetch(total); "backward" includes the input stage backward)')

if plain:
    sys.exit(0)
# forward only, un-pipelined
with torch.no_grad():
    for i in range(5):
        model(st.collate(idx[i % 4]))
    torch.cuda.synchronize()
    acc.clear()
    t0 = T()
    for i in range(n):
        model(st.collate(idx[i % 4]))
    host = (T() - t0) / n * 1e3
    torch.cuda.synchronize()
    wall = (T() - t0) / n * 1e3
model.verify()
print('%s forward through the store: host loop %.3f ms, wall %.3f ms' % (kind, host, wall))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('   %-18s %7.3f ms' % (k, v / n * 1e3))

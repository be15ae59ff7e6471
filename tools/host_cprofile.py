#!/usr/bin/env python
"""Where the HOST spends a training step (cProfile over the enqueue loop; the GPU runs behind):
    python tools/host_cprofile.py [rna|qm9|pdbbind] [steps]"""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import store as S, synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'rna'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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


def run(n):
    nxt = st.collate(idx[0])
    for i in range(n):
        cur, nxt = nxt, st.collate(idx[(i + 1) % 4])
        tr.step(cur, next_data=nxt)


run(20)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats('cumulative')
print('%s: %d steps; per-step microseconds = table seconds * 1e6 / %d' % (kind, steps, steps))
ps.print_stats(45)
ps.sort_stats('tottime')
ps.print_stats(30)
tr.drain()

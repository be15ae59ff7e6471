"""Training step with the input pipeline (next batch's graph built on the side stream while this step runs) against the
same steps on graphs prepared once and reused (no graph construction at all): the difference is what graph construction
costs the step although it runs beside it.  usage: noprefetch_bound.py [qm9|pdbbind|rna]"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch
import models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
kind = sys.argv[1] if len(sys.argv) > 1 else 'qm9'
dev = torch.device('cuda:0')
torch.manual_seed(1234)
if kind == 'qm9':
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    bs = [synth.qm9_batch(0, k * 128, 128).to(dev) for k in range(4)]
elif kind == 'pdbbind':
    cfg = models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    bs = [synth.collate([synth.pdbbind_complex(1, 32 * k + i) for i in range(32)]).to(dev) for k in range(2)] * 2
else:
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    bs = [synth.collate([synth.rna_chain(2, (i + 2 * k) % 8) for i in range(8)]).to(dev) for k in range(4)]
model = models.PAMNet(cfg).to(dev)
tr = Trainer(model, lr=1e-4)


def run(prefetch, n=60):
    for i in range(5):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4] if prefetch else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4] if prefetch else None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('%s with the input pipeline                 %.3f ms/step' % (kind, run(True)))
print('%s graph built in line (main stream)       %.3f ms/step' % (kind, run(False)))
keep = {}
for b in bs:
    model.prepare(b)
    keep[id(b)] = b._pamnet_prepared
    b._pamnet_prepared = None
model._graph = lambda data: keep[id(data)]
print('%s graphs cached (no graph construction)   %.3f ms/step' % (kind, run(False)))

"""PDBbind / RNA schema training steps and forwards through the resident store (device-side collation, sizes carried by the
batch: no device->host read) beside the plain-tensor loop of tools/scale_pipelined.py.  Run on the GPU box."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.store import MoleculeStore
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0')


def run(name, cfg, graphs, B, steps=20):
    torch.manual_seed(0)
    model = models.PAMNet(cfg).to(dev)
    store = MoleculeStore(graphs, dev).prepare_for(model)
    idx = [list(range(k * B, (k + 1) * B)) for k in range(len(graphs) // B)]
    nb = len(idx)
    tr = Trainer(model, lr=1e-4)
    nxt = store.collate(idx[0])
    for i in range(6):
        cur, nxt = nxt, store.collate(idx[(i + 1) % nb])
        tr.step(cur, next_data=nxt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        cur, nxt = nxt, store.collate(idx[(i + 1) % nb])
        tr.step(cur, next_data=nxt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    model.verify()
    with torch.no_grad():
        for i in range(3):
            model(store.collate(idx[i % nb]))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            model(store.collate(idx[i % nb]))
        torch.cuda.synchronize(); df = (time.perf_counter() - t0) / steps * 1e3
    model.verify()
    print('%-10s store path (collation inside the loop, no device->host read): %.2f ms/step, %.2f ms/forward un-pipelined' % (name, dt, df))


only = sys.argv[1] if len(sys.argv) > 1 else ''
if only in ('', 'pdbbind'):
    run('pdbbind', models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0),
        [synth.pdbbind_complex(1, i) for i in range(128)], 32)
if only in ('', 'rna'):
    run('rna', models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'),
        [synth.rna_chain(2, i) for i in range(32)], 8)
if only in ('', 'rna_d64'):
    run('rna_d64', models.Config(dataset='rna_train', dim=64, n_layer=2, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'),
        [synth.rna_chain(2, i) for i in range(32)], 8)

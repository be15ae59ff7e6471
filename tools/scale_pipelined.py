"""PDBbind / RNA schema training steps WITH the input pipeline (next batch's graph on the side stream), 4 distinct
resident batches -- the product's intended loop; tools/scale_smoke.py is the un-pipelined single-batch variant."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0')


def run(name, cfg, mk, steps=20):
    torch.manual_seed(0)
    model = models.PAMNet(cfg).to(dev)
    tr = Trainer(model, lr=1e-4)
    bs = [mk(k).to(dev) for k in range(4)]
    for i in range(6):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    # host-only time of a step: enqueue without waiting (the queue is deep enough for a few steps)
    t0 = time.perf_counter()
    for i in range(3):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    th = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    print('%-10s %.2f ms/step pipelined (host enqueue ~%.2f ms/step)' % (name, dt, th))


only = sys.argv[1] if len(sys.argv) > 1 else ''
if only in ('', 'pdbbind'):
    run('pdbbind', models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0), lambda k: synth.pdbbind_batch(1, 32 * k, 32))
if only in ('', 'rna'):
    run('rna', models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'), lambda k: synth.rna_batch(2, 8 * k, 8))
if only in ('', 'rna_d64'):
    run('rna_d64', models.Config(dataset='rna_train', dim=64, n_layer=2, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'), lambda k: synth.rna_batch(2, 8 * k, 8))

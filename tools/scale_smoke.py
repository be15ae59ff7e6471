import os, sys, time
REPO='/root/repo'
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev=torch.device('cuda:0')
def run(name, cfg, batch, steps=20):
    torch.manual_seed(0)
    model=models.PAMNet(cfg).to(dev)
    tr=Trainer(model, lr=1e-4)
    b=batch.to(dev)
    for _ in range(5): loss=tr.step(b)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(steps): loss=tr.step(b)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/steps*1e3
    g=model._graph_cache
    print('%-10s N=%d E_g=%d E_l=%d TP=%d  loss=%.4f finite=%s  %.2f ms/step (graph rebuilt every step)'%(name,g.n,g.glob.m,g.loc.m,g.tp.m,float(loss.detach()),bool(torch.isfinite(loss)),dt))
only=sys.argv[1] if len(sys.argv)>1 else ''
if only in ('','pdbbind'): run('pdbbind', models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0), synth.pdbbind_batch(1, 0, 32))
if only in ('','rna'): run('rna', models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'), synth.rna_batch(2, 0, 8))
if only in ('','rna_d64'): run('rna_d64', models.Config(dataset='rna_train', dim=64, n_layer=2, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'), synth.rna_batch(2, 0, 8))

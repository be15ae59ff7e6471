import os, sys
REPO='/root/repo'
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,'physics-aware-multiplex-gnn_amd'))
import torch, models
from pamnet_amd import synth, train
dev=torch.device('cuda:0'); torch.manual_seed(0)
for ds,dim in (('QM9',256),('PDBbind',160)):
    if ds=='QM9':
        cfg=models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0); bs=[synth.qm9_batch(0, 64*k, 64).to(dev) for k in range(4)]
    else:
        cfg=models.Config(dataset='PDBbind', dim=dim, n_layer=2, cutoff_l=2.0, cutoff_g=6.0); bs=[synth.pdbbind_batch(3, 4*k, 4).to(dev) for k in range(4)]
    model=models.PAMNet(cfg).to(dev)
    tr=train.Trainer(model, lr=2e-4, loss='l1' if ds=='QM9' else 'mse')
    ls=[]
    for i in range(300):
        ls.append(tr.step(bs[i%4]))
    tr.drain()
    v=[float(l) for l in ls]
    assert all(x==x and abs(x)<1e9 for x in v)
    print('%s dim=%d: loss first 4 %s ... last 4 %s' % (ds, dim, ['%.4f'%x for x in v[:4]], ['%.4f'%x for x in v[-4:]]))
    print('   parameters finite:', all(bool(torch.isfinite(p).all()) for p in model.parameters()), ' peak memory %.2f GB' % (torch.cuda.max_memory_allocated()/2**30))

import os, sys, time
REPO='/root/repo'
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import store as S, synth
from pamnet_amd.train import predict
dev = torch.device('cuda:0'); torch.manual_seed(0)
kind = sys.argv[1]
if kind == 'qm9':
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    graphs = [synth.qm9_molecule(0, i) for i in range(512)]; idx = [list(range(128*k, 128*k+128)) for k in range(4)]
else:
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    graphs = [synth.rna_chain(2, i) for i in range(8)]; idx = [[(i + 2*k) % 8 for i in range(8)] for k in range(4)]
model = models.PAMNet(cfg).to(dev)
st = S.MoleculeStore(graphs, dev).prepare_for(model)
n = 200
with torch.no_grad():
    for _ in predict(model, (st.collate(idx[i % 4]) for i in range(8))): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in predict(model, (st.collate(idx[i % 4]) for i in range(n))): pass
    torch.cuda.synchronize(); print(kind, 'store + predict (pipelined): %.3f ms/forward' % ((time.perf_counter() - t0) / n * 1e3))
    for i in range(8): model(st.collate(idx[i % 4]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): model(st.collate(idx[i % 4]))
    torch.cuda.synchronize(); print(kind, 'store, unpipelined: %.3f ms/forward' % ((time.perf_counter() - t0) / n * 1e3))
    bs = [st.collate(idx[i % 4]) for i in range(4)]
    gs = [model.prepare(b, need_grad=False)._pamnet_prepared for b in bs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        bs[i % 4]._pamnet_prepared = gs[i % 4]
        model(bs[i % 4])
    torch.cuda.synchronize(); print(kind, 'forward on a prepared graph (no graph construction at all): %.3f ms' % ((time.perf_counter() - t0) / n * 1e3))

"""QM9 B=128 forward-only: where the time goes (graph + basis construction vs the engine forward)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev).eval()
b = synth.qm9_batch(0, 0, 128).to(dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    tp = t(lambda: model.prepare(b, need_grad=False))
    gp = model.prepare(b, need_grad=False)._pamnet_prepared
    def fwd():
        b._pamnet_prepared = gp       # re-attach: a prepared graph is consumed by the forward that uses it
        model(b)
    tf = t(fwd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fwd()
    th = (time.perf_counter() - t0) / 30 * 1e3
    torch.cuda.synchronize()
    b._pamnet_prepared = None
    tall = t(lambda: model(b))
print('QM9 B=128 forward-only: prepare (graph+basis) %.3f ms | engine forward on a prepared graph %.3f ms (host enqueue %.3f) | end to end %.3f ms' % (tp, tf, th, tall))

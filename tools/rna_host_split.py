import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')).to(dev).eval()
b = synth.rna_batch(2, 0, 8).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    tp = t(lambda: model.prepare(b, need_grad=False))
    model.prepare(b, need_grad=False)
    def fwd():
        model(b)            # picks up the prepared graph
    tf = t(fwd)
    def fwd_host():
        model(b)
    # host-only enqueue time of the forward (no sync inside the loop)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): model(b)
    th = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
print('prepare (graph + basis, syncs inside): %.2f ms   forward on prepared graph: %.2f ms (host enqueue %.2f ms)' % (tp, tf, th))

"""Forward-only (inference_rna_puzzles.py shape: dim=16 n_layer=1, 8 graphs) timing on synthetic RNA-schema graphs."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
dev = torch.device('cuda:0')
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='rna_native', dim=dim, n_layer=nl, cutoff_l=2.6, cutoff_g=20.0,
                                    flow='target_to_source')).to(dev).eval()
b = synth.rna_batch(2, 0, 8).to(dev)
with torch.no_grad():
    for _ in range(3): out = model(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): out = model(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
g = model._graph_cache
# same loop with the input pipeline (graph of batch i+1 built on a side stream while batch i runs)
from pamnet_amd.train import predict
bs = [synth.rna_batch(2, 8 * k, 8).to(dev) for k in range(4)]
for _ in predict(model, bs * 2): pass
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 0
for _d, o in predict(model, bs * (steps // 2)): n += 1
torch.cuda.synchronize(); dtp = (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    for k in range(4): model(bs[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): model(bs[k % 4])
    torch.cuda.synchronize(); dts = (time.perf_counter() - t0) / steps * 1e3
print('pipelined: %.2f ms/forward (%.0f graphs/s) over %d distinct batches; same batches one after the other: %.2f ms/forward'
      % (dtp, 8e3 / dtp, len(bs), dts))
print('rna infer d=%d L=%d  N=%d E_g=%d E_l=%d TP=%d  %.2f ms/forward  (%.0f graphs/s)  out[0]=%.6f' % (
    dim, nl, g.n, g.glob.m, g.loc.m, g.tp.m, dt, 8e3 / dt, float(out[0])))

"""Host-side profile of the training step (cProfile) + GPU busy time, to separate launch overhead from kernel time."""
import cProfile, pstats, sys, os, time, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
import models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
tr = Trainer(model)
bs = [synth.qm9_batch(0, 128 * k, 128).to(dev) for k in range(4)]
for i in range(5):
    tr.step(bs[i % 4])
torch.cuda.synchronize()
# enqueue-only time vs synchronized time
t0 = time.perf_counter()
for i in range(20):
    tr.step(bs[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step, incl. drain %.2f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
# phases
def timed(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    print('graph build   %.3f ms' % timed(lambda i: model._graph(bs[i % 4])))
    print('forward       %.3f ms' % timed(lambda i: model(bs[i % 4])))
print('fwd+bwd       %.3f ms' % timed(lambda i: tr.forward_backward(bs[i % 4])))
print('clip+adam+ema %.3f ms' % timed(lambda i: tr.native_update()))
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    tr.step(bs[i % 4])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])

"""What about side-stream work slows the main-stream step: the number of launches, or the CU time they take?  (GPU box)
Graphs are prepared once and reused (no real side-stream work); each step then enqueues K trivial one-workgroup
launches on a side stream (K = 0 .. 80), or the real graph construction."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0'); torch.manual_seed(1234)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
tr = Trainer(model, lr=1e-4)
bs = [synth.qm9_batch(0, k * 128, 128).to(dev) for k in range(4)]
side = torch.cuda.Stream(device=dev)
tiny = torch.zeros(64, device=dev)
big = torch.zeros(1 << 22, device=dev)
def run(extra, n=100):
    for i in range(5):
        tr.step(bs[i % 4]); extra()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        extra()
        tr.step(bs[i % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def real(n=100):
    for i in range(5):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('real graph construction on the side stream   %.3f ms' % real())
for b in bs:
    model.prepare(b)
keep = {id(b): b._pamnet_prepared for b in bs}
model._graph = lambda data: keep[id(data)]
def launches(k, t):
    def f():
        with torch.cuda.stream(side):
            for _ in range(k):
                t.add_(1.0)
    return f
for rep in range(2):
    for k in (0, 10, 20, 40, 80):
        print('cached graphs + %2d one-workgroup launches      %.3f ms' % (k, run(launches(k, tiny))))
    for k in (1, 4, 16):
        print('cached graphs + %2d 16 MB elementwise launches  %.3f ms' % (k, run(launches(k, big))))

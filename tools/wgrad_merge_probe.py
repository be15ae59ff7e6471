"""What merging a layer pair's two weight-gradient launches would buy (QM9 B=128 shapes): the local layer's own batch (5
node-level + 4 local-edge + 2 triplet/pair jobs) and the global layer's (3 node-level + 2 global-edge jobs) as two deferred
launches vs one."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
from pamnet_amd import fused, lib
lib.load()
dev = torch.device('cuda:0')
n, eg, el, tp, d = 2286, 32888, 4316, 17640, 128
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
keep = []
def jobs(spec):
    out = []
    for rows, cnt in spec:
        for _ in range(cnt):
            dz, a, dw = rnd(rows, d), rnd(rows, d), torch.empty(d, d, device=dev)
            keep.extend([dz, a, dw])
            out.append((dz, d, a, d, 0, rows, dw, d, None))
    return out
L = jobs(((n, 5), (el, 4), (tp, 2)))
G = jobs(((n, 3), (eg, 2)))
def timeit(fn, reps=40, groups=5):
    vals = []
    for _ in range(groups):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); e.synchronize()
        vals.append(s.elapsed_time(e) / reps * 1e3)
    return sorted(vals)[len(vals) // 2]
w = fused.DeferredWgrad(keep[0])
def two():
    w.launch(L); w.launch(G)
def one():
    w.launch(L + G)
for f in (two, one):
    f(); f()
torch.cuda.synchronize()
print('two launches (L then G): %.1f us per pair' % timeit(two))
print('one launch   (L + G)   : %.1f us per pair' % timeit(one))
w.launch(L); print('L alone: %.1f us' % timeit(lambda: w.launch(L)))
print('G alone: %.1f us' % timeit(lambda: w.launch(G)))
w.flush()

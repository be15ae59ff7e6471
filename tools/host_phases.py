#!/usr/bin/env python
"""Host-side time of each phase of Trainer.step (no GPU syncs added): is the step enqueue-bound?"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(1234)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
tr = Trainer(model, lr=1e-4)
bs = [synth.qm9_batch(0, k * 128, 128).to(dev) for k in range(4)]
for i in range(5):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
n = 60
acc = {k: [] for k in ('wait', 'fwd', 'bwd', 'update', 'prefetch', 'total')}
T = time.perf_counter
t_all = T()
for i in range(n):
    d, nx = bs[i % 4], bs[(i + 1) % 4]
    t0 = T()
    tr._wait_prepared(d)
    t1 = T()
    if not tr._grad_clean:
        tr.fp.zero_grad()
    tr._grad_clean = False
    out = model(d)
    loss = torch.nn.functional.l1_loss(out, d.y)
    t2 = T()
    loss.backward()
    t3 = T()
    tr.sync_gradients()
    tr.native_update()
    t4 = T()
    tr.prefetch(nx)
    t5 = T()
    for k, v in zip(('wait', 'fwd', 'bwd', 'update', 'prefetch', 'total'), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
        acc[k].append(v * 1e3)
torch.cuda.synchronize()
print('wall %.3f ms/step' % ((T() - t_all) / n * 1e3))
for k, v in acc.items():
    s = sorted(v)
    print('%-9s mean %.3f  median %.3f  p90 %.3f  max %.3f ms' % (k, sum(v) / n, s[n // 2], s[int(.9 * n)], s[-1]))

# pure enqueue cost of each phase with an idle GPU in front (sync before, no sync after)
def pure(fn, reps=20):
    ts = []
    for i in range(reps):
        torch.cuda.synchronize()
        t0 = T()
        fn(i)
        ts.append((T() - t0) * 1e3)
    torch.cuda.synchronize()
    ts.sort()
    return ts[len(ts) // 2]

state = {}
def fwd(i):
    d = bs[i % 4]
    if not hasattr(d, '_pamnet_prepared') or d._pamnet_prepared is None:
        pass
    state['loss'] = torch.nn.functional.l1_loss(model(d), d.y)
def bwd(i):
    state['loss'].backward()
for i in range(3):
    model.prepare(bs[i % 4]); fwd(i); bwd(i)
print('pure enqueue (GPU idle in front), median of 20:')
t_prep = pure(lambda i: model.prepare(bs[i % 4]))
print('  prepare (graph build, incl. its host syncs) %.3f ms' % t_prep)
def fb(i):
    model.prepare(bs[i % 4])
    torch.cuda.synchronize()
    t0 = T(); fwd(i); t1 = T(); bwd(i); t2 = T()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
r = [fb(i) for i in range(20)]
print('  forward enqueue %.3f ms   backward enqueue %.3f ms' % (sorted(a for a, b in r)[10], sorted(b for a, b in r)[10]))
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:4500])

#!/usr/bin/env python
"""Host time spent inside each C-ABI call vs around them (monkeypatched lib.call; thread-agnostic, so the autograd
worker thread is covered)."""
import collections
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import fused, lib, synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

T = time.perf_counter
acc = collections.defaultdict(lambda: [0, 0.0])
orig = lib.call


def timed_call(name, *a):
    t0 = T()
    r = orig(name, *a)
    e = acc[name]
    e[0] += 1
    e[1] += T() - t0
    return r


lib.call = timed_call
for mod in (fused, sys.modules['pamnet_amd.ops'], sys.modules['pamnet_amd.graph']):
    if getattr(mod, 'lib', None) is lib:
        pass                      # modules call lib.call(...) through the module attribute: patched above

for name in ('forward', 'backward'):
    for cls in (fused._Stack, fused._Embed):
        f = getattr(cls, name)

        def wrap(f, label):
            def g(*a, **k):
                t0 = T()
                r = f(*a, **k)
                e = acc[label]
                e[0] += 1
                e[1] += T() - t0
                return r
            return staticmethod(g)
        setattr(cls, name, wrap(f, '%s.%s (python total)' % (cls.__name__, name)))

dev = torch.device('cuda:0')
torch.manual_seed(1234)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
tr = Trainer(model, lr=1e-4)
bs = [synth.qm9_batch(0, k * 128, 128).to(dev) for k in range(4)]
for i in range(5):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
acc.clear()
n = 40
t0 = T()
for i in range(n):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
print('wall %.3f ms/step' % ((T() - t0) / n * 1e3))
tot = 0.0
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-42s %6.1f calls/step %8.3f ms/step' % (k, c / n, t / n * 1e3))

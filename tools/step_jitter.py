#!/usr/bin/env python
"""Per-step GPU time of the bench loop (event per step, no host syncs inside the loop): exposes jitter / outliers."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
model = models.PAMNet(cfg).to(dev)
tr = Trainer(model, lr=1e-4)
batches = [synth.qm9_batch(0, k * 128, 128).to(dev) for k in range(4)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for rep in range(3):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(5):
        tr.step(batches[i % 4], next_data=batches[(i + 1) % 4])
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(n):
        tr.step(batches[i % 4], next_data=batches[(i + 1) % 4])
        evs[i + 1].record()
    torch.cuda.synchronize()
    d = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    s = sorted(d)
    print('rep %d: mean %.3f  median %.3f  min %.3f  p90 %.3f  max %.3f ms   by batch: %s' % (
        rep, sum(d) / n, s[n // 2], s[0], s[int(0.9 * n)], s[-1],
        ' '.join('%.2f' % (sum(d[b::4]) / len(d[b::4])) for b in range(4))))

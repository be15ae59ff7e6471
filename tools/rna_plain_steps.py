#!/usr/bin/env python
"""RNA B=8 d=16 L=1 training steps on PLAIN tensors (the reference calling convention: sizes come back from the device) as
bench.other_configs runs them, for same-box A/B runs of environment switches:  PAMNET_KNN_TP_TOTAL=0 python tools/rna_plain_steps.py"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
rna = [synth.rna_chain(2, i) for i in range(8)]
torch.manual_seed(7)
model = models.PAMNet(cfg).to(dev)
tr = Trainer(model, loss='smooth_l1', max_grad_norm=None, ema_decay=None, lr=1e-4)
bs = [synth.collate([rna[(i + 2 * k) % 8] for i in range(8)]).to(dev) for k in range(4)]
for i in range(20):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
print('rna plain tensors: %.3f ms/step (%d steps), PAMNET_KNN_TP_TOTAL=%s' % ((time.perf_counter() - t0) / steps * 1e3, steps,
                                                                            os.environ.get('PAMNET_KNN_TP_TOTAL', '0')))
tr.drain()

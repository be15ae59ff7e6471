#!/usr/bin/env python
"""PDBbind B=32 d=128 L=3 training steps as bench.other_configs runs them (plain tensors + side-stream input pipeline, the
driver's MSE step), for same-box A/B runs of environment switches:  PAMNET_EDGE_WGRAD=0 python tools/pdbbind_steps.py"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda:0')
torch.manual_seed(7)
cfg = models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
pdb = [synth.pdbbind_complex(1, i) for i in range(128)]
model = models.PAMNet(cfg).to(dev)
tr = Trainer(model, loss='mse', max_grad_norm=None, ema_decay=None, lr=1e-3)
bs = [synth.collate([pdb[i] for i in range(32 * k, 32 * k + 32)]).to(dev) for k in range(4)]
for i in range(20):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
print('pdbbind plain tensors: %.3f ms/step (%d steps), PAMNET_EDGE_WGRAD=%s PAMNET_AGG_PP=%s PAMNET_EDGE_RECOMPUTE=%s' % ((time.perf_counter() - t0) / steps * 1e3, steps,
                                                                              os.environ.get('PAMNET_EDGE_WGRAD', 'auto'), os.environ.get('PAMNET_AGG_PP', 'auto'), os.environ.get('PAMNET_EDGE_RECOMPUTE', '0')))
tr.drain()

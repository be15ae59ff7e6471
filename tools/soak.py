#!/usr/bin/env python
"""Soak of the three training configurations: thousands of steps each through Trainer (input pipeline on, graphs rebuilt every
step), watching step time, device memory and the loss staying finite; store batches for half of the run (deferred checks)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
import models
from pamnet_amd import synth
from pamnet_amd.store import MoleculeStore
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0')
for tag, cfg, graphs, per, steps, kw in (
        ('qm9 d128 L6 B128', models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0),
         [synth.qm9_molecule(0, i) for i in range(1024)], 128, 4000, {}),
        ('qm9 d96 L3 B64 (zero-padded width)', models.Config(dataset='QM9', dim=96, n_layer=3, cutoff_l=5.0, cutoff_g=5.0),
         [synth.qm9_molecule(3, i) for i in range(512)], 64, 1500, {}),
        ('rna d16 L1 B8', models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source'),
         [synth.rna_chain(2, i) for i in range(16)], 8, 3000, dict(loss='smooth_l1', max_grad_norm=None, ema_decay=None)),
        ('pdbbind d128 L3 B32', models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0),
         [synth.pdbbind_complex(1, i) for i in range(64)], 32, 300, dict(loss='mse', max_grad_norm=None, ema_decay=None))):
    torch.manual_seed(0)
    model = models.PAMNet(cfg).to(dev)
    tr = Trainer(model, lr=1e-4, **kw)
    nb = len(graphs) // per
    plain = [synth.collate(graphs[k * per:(k + 1) * per]).to(dev) for k in range(nb)]
    st = MoleculeStore(graphs, dev).prepare_for(model)
    sel = [list(range(k * per, (k + 1) * per)) for k in range(nb)]
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    mem0 = torch.cuda.memory_reserved()
    t0 = time.perf_counter()
    loss = None
    for i in range(steps // 2):
        loss = tr.step(plain[i % nb], next_data=plain[(i + 1) % nb])
    nxt = st.collate(sel[0])
    for i in range(steps - steps // 2):
        cur, nxt = nxt, st.collate(sel[(i + 1) % nb])
        loss = tr.step(cur, next_data=nxt)
    tr.close()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss), tag
    print('%-38s %5d steps  %.3f ms/step  final loss %.4f  reserved %.0f -> %.0f MB (peak allocated %.0f MB)' % (
        tag, steps, dt / steps * 1e3, float(loss), mem0 / 2**20, torch.cuda.memory_reserved() / 2**20,
        torch.cuda.max_memory_allocated() / 2**20))
    del tr, model, plain, st
    torch.cuda.empty_cache()

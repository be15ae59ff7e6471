#!/usr/bin/env python
"""Round 6: would the weight-gradient launches overlap with the backward's dependent chain if they ran on a stream of their own?
Nothing on the critical path consumes them before the optimiser.  Emulation on the real step: EXTRA copies of a layer pair's
weight-gradient launch (the same 16 jobs at the QM9 batch's row counts: bench.py step_kernel_rooflines) are issued on a side stream,
six per step, while the step runs unchanged.  If the step slows down by about the copies' own time (6 x ~41 us), concurrency buys
nothing; if by much less, moving the real launches off the main stream would hide that much.
GPU box:  python tools/wgrad_overlap_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import fused, store as S, synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
graphs = [synth.qm9_molecule(0, i) for i in range(512)]
idx = [list(range(128 * k, 128 * k + 128)) for k in range(4)]
model = models.PAMNet(cfg).to(dev)
st = S.MoleculeStore(graphs, dev).prepare_for(model)
tr = Trainer(model, lr=1e-4)
n, eg, el, tp, d = 2286, 32888, 4316, 17640, 128
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    keep, jobs = [], []
    for rows, cnt in ((n, 5), (el, 4), (tp, 2), (n, 3), (eg, 2)):
        for _ in range(cnt):
            dz, a, dw = rnd(rows, d), rnd(rows, d), torch.empty(d, d, device=dev)
            keep += [dz, a, dw]
            jobs.append((dz, d, a, d, 0, rows, dw, d, None))
    dw_ = fused.DeferredWgrad(keep[0])
    for _ in range(3):
        dw_.launch(jobs)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    s.record()
    for _ in range(30):
        dw_.launch(jobs)
    e.record()
e.synchronize()
alone = s.elapsed_time(e) / 30 * 1e3
print('one weight-gradient launch of a layer pair, alone: %.1f us' % alone)


def run(steps, copies):
    nxt = st.collate(idx[0])
    for i in range(steps):
        cur, nxt = nxt, st.collate(idx[(i + 1) % 4])
        if copies:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                for _ in range(copies):
                    dw_.launch(jobs)
        tr.step(cur, next_data=nxt)
    torch.cuda.current_stream().wait_stream(side)


def timed(copies, steps=200):
    run(10, copies)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, copies)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    base = timed(0)
    for c in (3, 6, 12):
        t = timed(c)
        print('step %.3f ms;  + %2d extra weight-gradient launches on a side stream (%.0f us of work alone): %.3f ms  (%+.0f us = %.0f %% of it)'
              % (base, c, c * alone, t, (t - base) * 1e3, (t - base) * 1e3 / (c * alone) * 100))
dw_.flush()
tr.drain()

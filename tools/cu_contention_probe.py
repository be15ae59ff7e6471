#!/usr/bin/env python
"""Round 6: what do workgroups of ANOTHER kernel cost the training step when they hold some CUs during the backward -- the
situation at N > 1, where RCCL's all-reduce kernels (one workgroup per channel) run beside the backward on the communication
stream?  The step's big kernels launch one 8-wave, 256-register workgroup per CU (256 of them): a CU held by anybody else means a
second round for one workgroup.  A squatter kernel (K workgroups x D microseconds, tools/probes/squatter.hip) is started on a side
stream when the backward starts.  GPU box:  python tools/cu_contention_probe.py [qm9|pdbbind]"""
import ctypes
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import store as S, synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

so = '/tmp/libsquat.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared',
                       os.path.join(REPO, 'tools', 'probes', 'squatter.hip'), '-o', so])
kind = sys.argv[1] if len(sys.argv) > 1 else 'qm9'
dev = torch.device('cuda:0')
torch.manual_seed(0)
if kind == 'qm9':
    cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    graphs = [synth.qm9_molecule(0, i) for i in range(512)]
    idx = [list(range(128 * k, 128 * k + 128)) for k in range(4)]
    loop = dict(lr=1e-4)
else:
    cfg = models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    graphs = [synth.pdbbind_complex(1, i) for i in range(64)]
    idx = [list(range(32 * (k % 2), 32 * (k % 2) + 32)) for k in range(4)]
    loop = dict(loss='mse', max_grad_norm=None, ema_decay=None, lr=1e-3)
model = models.PAMNet(cfg).to(dev)
st = S.MoleculeStore(graphs, dev).prepare_for(model)
tr = Trainer(model, **loop)
lib = ctypes.CDLL(so)
lib.squat.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def run(n, wgs, micros, pieces):
    nxt = st.collate(idx[0])
    for i in range(n):
        cur, nxt = nxt, st.collate(idx[(i + 1) % 4])
        if wgs:
            # the squatters start with the step (the forward is ~1/3 of it) and come in `pieces` launches, like buckets
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            for _ in range(pieces):
                lib.squat(wgs, micros / pieces, sink.data_ptr(), side.cuda_stream)
        tr.step(cur, next_data=nxt)
    torch.cuda.current_stream().wait_stream(side)


def timed(wgs, micros, pieces=3, n=200):
    run(10, wgs, micros, pieces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n, wgs, micros, pieces)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


n = 200 if kind == 'qm9' else 60
base = timed(0, 0, n=n)
print('%s step through the store, nobody else on the device:           %.3f ms' % (kind, base))
for wgs, micros in ((8, 300), (32, 300), (32, 600), (64, 300)):
    t = timed(wgs, micros, n=n)
    print('  + %2d squatting workgroups for %3d us per step (3 launches):   %.3f ms  (%+.0f us)' % (wgs, micros, t, (t - base) * 1e3))
tr.drain()

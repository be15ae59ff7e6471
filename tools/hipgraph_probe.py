#!/usr/bin/env python
"""Round 6: what would HIP-graph capture of the step buy?  (SURVEY 8f N2 asked for "HIP-graph capture of the L-layer loop"; the
design replaced it by one C engine call per direction.)  One QM9 B=128 d=128 L=6 training step -- graph construction + basis,
forward, loss, backward, clip + Adam + EMA: every launch of it -- on ONE fixed batch collated by the store (sizes on the host: no
read-back), eager against a captured hipGraph replayed (torch.cuda.CUDAGraph = hipGraph on ROCm).  A fixed batch is the best case
for a graph (a real loop needs one graph per batch shape); the question is only what the replay saves.
GPU box:  python tools/hipgraph_probe.py [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from pamnet_amd import store as S, synth  # noqa: E402
from pamnet_amd.train import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
graphs = [synth.qm9_molecule(0, i) for i in range(128)]
model = models.PAMNet(cfg).to(dev)
st = S.MoleculeStore(graphs, dev).prepare_for(model)
tr = Trainer(model, lr=1e-4)
batch = st.collate(list(range(128)))


def step():
    # Trainer.step without its host-side throttle (an event wait: not capturable, and beside the point here): graph construction in
    # line on the calling stream, forward, loss, backward, clip + Adam + EMA + zero_grad
    tr.forward_backward(batch)
    tr.sync_gradients()
    tr.native_update(None)


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timed(step, steps)
print('eager, one fixed batch, graph construction in line: %.3f ms/step' % eager)
tr.drain()
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    replay = timed(g.replay, steps)
    print('captured hipGraph of the same step, replayed:       %.3f ms/step  (%+.1f %%)' % (replay, (replay / eager - 1) * 100))
except Exception as e:                                          # noqa: BLE001 -- a probe: report what stops the capture
    print('capture failed: %s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else ''))

"""RNA d=16 training step under the four loop shapes (same / distinct batches x with / without the input pipeline):
ms per step and the number of device segments the caching allocator had to create in the timed region."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0')
dim, nl = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 1)
cfg = models.Config(dataset='rna_native', dim=dim, n_layer=nl, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')


def run(distinct, pipelined, steps=30):
    torch.manual_seed(0)
    model = models.PAMNet(cfg).to(dev)
    tr = Trainer(model, lr=1e-4)
    bs = [synth.rna_batch(2, 8 * (k if distinct else 0), 8).to(dev) for k in range(4)]
    nxt = (lambda i: bs[(i + 1) % 4]) if pipelined else (lambda i: None)
    for i in range(8):
        tr.step(bs[i % 4], next_data=nxt(i))
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(bs[i % 4], next_data=nxt(i))
    th = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    st1 = torch.cuda.memory_stats()
    seg = st1['segment.all.allocated'] - st0['segment.all.allocated']
    print('distinct=%d pipelined=%d  %.2f ms/step (host loop %.2f)  new segments %d  reserved %.2f GB' % (
        distinct, pipelined, dt, th, seg, st1['reserved_bytes.all.current'] / 2**30), flush=True)
    del tr, model


for d in (0, 1):
    for p in (0, 1):
        run(d, p)

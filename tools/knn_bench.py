import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch
from pamnet_amd import graph as G, synth, lib
dev = torch.device('cuda:0')
b = synth.rna_batch(2, 0, 8)
pos = b.x[:, :3].contiguous().to(dev)
nodeg = b.batch.to(torch.int32).to(dev)
gptr, _ = G.csr_from_keys(nodeg, 8)
print('graph sizes', (gptr[1:] - gptr[:-1]).tolist())
def run(): return G.knn_table(pos, nodeg, gptr, 50, 20.0)
for _ in range(3): run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print('knn_table %.1f us per call (n=%d)' % (a.elapsed_time(e) / 20 * 1e3, pos.size(0)))

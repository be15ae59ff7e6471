"""sbf_radial launch time at the RNA B=8 and QM9 B=128 local-edge counts (GPU box)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch
from pamnet_amd import lib
dev = torch.device('cuda:0')
for name, m, c in (('RNA B=8', 75760, 16.0), ('QM9 B=128', 4900, 5.0)):
    dist = torch.rand(m, device=dev) * c * 0.98 + 0.02 * c
    rad = torch.empty(m * 42, device=dev)
    st = lib.stream_of(dist)
    for _ in range(5):
        lib.call('pamnet_sbf_radial_f32', lib.ptr(dist), c, m, lib.ptr(rad), st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        lib.call('pamnet_sbf_radial_f32', lib.ptr(dist), c, m, lib.ptr(rad), st)
    b.record(); torch.cuda.synchronize()
    print('%-10s %6d edges: %.1f us / launch' % (name, m, a.elapsed_time(b) * 1e3 / 200))

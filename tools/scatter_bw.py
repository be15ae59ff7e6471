import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
from pamnet_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
R, D = 272034, 128
lens = torch.randint(8, 22, (R,), generator=g)
ptr = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(torch.int32).to(dev)
M = int(ptr[-1])
src = torch.randn(M, D, device=dev)
out = torch.empty(R, D, device=dev)
by = 4 * D * M + 4 * (R + 1) + 4 * D * R
for rep in range(4):
    for _ in range(3):
        ops.segment_sum_raw(out, None, src, None, None, None, None, ptr, R, D)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.segment_sum_raw(out, None, src, None, None, None, None, ptr, R, D)
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / 10
    print('rep %d: %.3f ms  %.0f GB/s' % (rep, ms, by / ms / 1e6))

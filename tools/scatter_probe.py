"""Streamed scatter-add probe (pamnet_segment_sum_f32 on ~2 GB) for rocprofv3 PMC passes."""
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
for _ in range(5):
    ops.segment_sum_raw(out, None, src, None, None, None, None, ptr, R, D)
torch.cuda.synchronize()
print('rows_in', M, 'rows_out', R, 'algorithmic bytes', 4 * D * M + 4 * (R + 1) + 4 * D * R)

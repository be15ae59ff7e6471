"""Gather-form (transposed CSR, `perm`) scatter-add probe at the PDBbind shape for rocprofv3 PMC passes: the source-side
reduction d P_j[j] = sum over the edges leaving j of d z[e] (csrc/engine.hip: pamnet_segment_sum_f32 with gT_perm / gT_ptr)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
import models
from pamnet_amd import ops, synth
dev = torch.device('cuda:0')
model = models.PAMNet(models.Config(dataset='PDBbind', dim=128, n_layer=1, cutoff_l=2.0, cutoff_g=6.0)).to(dev)
b = synth.collate([synth.pdbbind_complex(1, i) for i in range(32)]).to(dev)
model.prepare(b, need_grad=True)
g = b._pamnet_prepared
n, m, D = g.n, g.glob.m, 128
src, out = torch.randn(m, D, device=dev), torch.empty(n, D, device=dev)
perm, ptr = g.glob_T.perm, g.glob_T.ptr
for _ in range(25):
    ops.segment_sum_raw(out, None, src, None, None, None, perm, ptr, n, D)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    ops.segment_sum_raw(out, None, src, None, None, None, perm, ptr, n, D)
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / 50
print('%.1f us per launch back to back (the source rows partly stay in the 256 MB Infinity Cache), %.0f GB/s of the algorithmic bytes' % (
    us, (4 * D * m + 4 * m + 4 * (n + 1) + 4 * D * n) / us * 1e-3))
print('rows_in', m, 'rows_out', n, 'algorithmic bytes', 4 * D * m + 4 * m + 4 * (n + 1) + 4 * D * n)

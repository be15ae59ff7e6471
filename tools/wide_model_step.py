"""A training step of PAMNet at a hidden size above 128 (default 256: QM9 schema, 128 molecules, 2 layers) for a rocprofv3
kernel trace: every GEMM of it must be csrc/dense.hip's dense_gemm_kernel -- no rocBLAS / hipBLASLt / Tensile (Cijk_*) kernel
may appear (tools/prof_wide.sh greps the statistics for them)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
import models
from pamnet_amd import synth
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='QM9', dim=dim, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
b = synth.qm9_batch(0, 0, 128).to(dev)
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
    opt.zero_grad()
    torch.nn.functional.l1_loss(model(b), b.y).backward()
    opt.step()
t1 = torch.cuda.Event(enable_timing=True); t1.record(); torch.cuda.synchronize()
print('PAMNet dim=%d n_layer=2, 128 molecules: %.2f ms per step (plain autograd loop, layer-by-layer path)' % (dim, t0.elapsed_time(t1) / steps))

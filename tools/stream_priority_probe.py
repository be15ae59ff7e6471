"""Does running the step on a high-priority stream shield it from the side-stream graph construction?  (GPU box)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0'); torch.manual_seed(1234)
print('priority range (least, greatest):', torch.cuda.Stream.priority_range())
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
tr = Trainer(model, lr=1e-4)
bs = [synth.qm9_batch(0, k * 128, 128).to(dev) for k in range(4)]
def run(n=100):
    for i in range(5):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('default stream            %.3f ms' % run())
hp = torch.cuda.Stream(device=dev, priority=-1)
torch.cuda.synchronize()
with torch.cuda.stream(hp):
    print('high-priority main stream %.3f ms' % run())
torch.cuda.synchronize()
print('default stream again      %.3f ms' % run())

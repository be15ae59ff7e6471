"""Forward-only loop (train.predict) on the QM9 B=128 workload, for rocprofv3 kernel traces of the inference path."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import predict
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev).eval()
bs = [synth.qm9_batch(0, 128 * k, 128).to(dev) for k in range(4)]
with torch.no_grad():
    for _ in predict(model, (bs[i % 4] for i in range(60))):
        pass
torch.cuda.synchronize()

"""Host enqueue time of a QM9 B=128 training step vs its GPU time: steps are enqueued back to back without waiting (the
queue holds several steps), so the loop time of a short burst is the host's cost per step."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch, models
from pamnet_amd import synth
from pamnet_amd.train import Trainer
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
tr = Trainer(model, lr=1e-4)
bs = [synth.qm9_batch(0, 128 * k, 128).to(dev) for k in range(4)]
for i in range(20):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('200 steps: host loop returned after %.3f ms/step, GPU done after %.3f ms/step' % (t_host / 200 * 1e3, t_all / 200 * 1e3))
# the prefetch's size round trip makes the host wait for the side stream once per step: time the pieces
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(22)
st.print_stats('host_ints|tolist|synchronize')

"""Pure host cost of a QM9 B=128 training step: the lead bound lifted, a burst of steps enqueued without waiting.
The loop time of the burst is what the host needs per step (including the prefetch's blocking size round trip on the
side stream); compare with the GPU's ms/step.  (GPU box)"""
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
for lead in (2, 10 ** 6):
    tr.MAX_STEPS_IN_FLIGHT = lead
    tr.__dict__['_inflight'] = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('lead bound %7d: host loop returned after %.3f ms/step, GPU done after %.3f ms/step' % (lead, t_host / 60 * 1e3, t_all / 60 * 1e3))
tr.MAX_STEPS_IN_FLIGHT = 2
tr.__dict__['_inflight'] = []
# pieces of the host's step
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(30)

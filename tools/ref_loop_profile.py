"""Where the host time of the reference's loop body goes on the HIP model (bench.py `reference_loop_unchanged`): cProfile of the
steady-state step (main_qm9.py:103-118 verbatim), split by phase with synchronised wall clocks."""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
import models
from torch.nn.utils import clip_grad_norm_
from utils import EMA
from pamnet_amd import synth
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, amsgrad=False)
ema = EMA(model, decay=0.999)
batches = [synth.qm9_batch(0, 128 * k, 128).to(dev) for k in range(4)]
model.train()
phase = {}
def tick(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    phase[name] = phase.get(name, 0.0) + t - t0
    return t
def step(data, timed=False):
    t = time.perf_counter()
    optimizer.zero_grad()
    if timed: t = tick('zero_grad', t)
    output = model(data)
    if timed: t = tick('forward', t)
    loss = torch.nn.functional.l1_loss(output, data.y)
    li = loss.item()
    if timed: t = tick('loss + item', t)
    loss.backward()
    if timed: t = tick('backward', t)
    clip_grad_norm_(model.parameters(), max_norm=1000, norm_type=2)
    if timed: t = tick('clip_grad_norm_', t)
    optimizer.step()
    if timed: t = tick('optimizer.step', t)
    ema(model)
    if timed: t = tick('ema', t)
for i in range(5):
    step(batches[i % 4])
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for i in range(n):
    step(batches[i % 4])
torch.cuda.synchronize()
print('loop: %.2f ms/step' % ((time.perf_counter() - t0) / n * 1e3))
for i in range(n):
    step(batches[i % 4], timed=True)
print('phases (synchronised after each, ms/step):', {k: round(v / n * 1e3, 2) for k, v in phase.items()})
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    step(batches[i % 4])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
st.print_callers('named_parameters')
st.print_callers('_named_members')

import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch
from pamnet_amd import lib, narrow
dev = torch.device('cuda:0'); lib.load()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
d, m = 64, 669280
x = torch.randn(m, d, device=dev); g = torch.randn(m, d, device=dev)
w1 = torch.randn(d, d, device=dev) * 0.3; b1 = torch.randn(d, device=dev)
dx = torch.empty_like(x); partial = torch.empty(256, 2 * d * d + 2 * d, device=dev)
dw, db = torch.empty(d, d, device=dev), torch.empty(d, device=dev)
st = lib.stream_of(x); P = lib.ptr
for act in (1, 0):
    for use_dx in (1, 0):
        t = timeit(lambda: lib.call('pamnet_narrow_linear_bwd_f32', P(x), m, d, P(w1), d, P(b1), act, P(g), d, P(dx) if use_dx else None, 0, P(partial), P(dw), P(db), st))
        print('linear_bwd d=64 m=%d act=%d dx=%d : %.1f us' % (m, act, use_dx, t))

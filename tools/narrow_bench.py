"""Per-call time of the narrow row kernels (HIP events around repeated C-ABI calls)."""
import os, sys, ctypes
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch
from pamnet_amd import lib, narrow
dev = torch.device('cuda:0')
lib.load()

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for d in (16, 64):
    for m in (17700, 75760, 669280):
        x = torch.randn(m, d, device=dev); g = torch.randn(m, d, device=dev)
        w1, w2 = torch.randn(d, d, device=dev) * 0.3, torch.randn(d, d, device=dev) * 0.3
        b1, b2 = torch.randn(d, device=dev), torch.randn(d, device=dev)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        nb = narrow._blocks(m)
        partial = torch.empty(nb, 2 * d * d + 2 * d, device=dev)
        dw, db = torch.empty(2, d, d, device=dev), torch.empty(2, d, device=dev)
        st = lib.stream_of(x)
        P = lib.ptr
        t_lf = timeit(lambda: lib.call('pamnet_narrow_linear_fwd_f32', P(x), m, d, P(w1), d, P(b1), 1, P(y), d, st))
        t_lb = timeit(lambda: lib.call('pamnet_narrow_linear_bwd_f32', P(x), m, d, P(w1), d, P(b1), 1, P(g), d, P(dx), 0, P(partial), P(dw[0]), P(db[0]), st))
        t_mf = timeit(lambda: lib.call('pamnet_narrow_mlp2_fwd_f32', P(x), m, d, P(w1), P(b1), P(w2), P(b2), 1, None, P(y), st))
        t_mb = timeit(lambda: lib.call('pamnet_narrow_mlp2_bwd_f32', P(x), m, d, P(w1), P(b1), P(w2), P(b2), P(g), 1, P(dx), P(partial), P(dw), P(db), st))
        gb = m * d * 4 / 1e9
        print('d=%2d m=%7d  linear fwd %7.1f us (%5.0f GB/s)  bwd %7.1f us | mlp2 fwd %7.1f us  bwd %7.1f us (%5.0f GB/s min traffic)' % (
            d, m, t_lf, 2 * gb / t_lf * 1e6, t_lb, t_mf, t_mb, 3 * gb / t_mb * 1e6))

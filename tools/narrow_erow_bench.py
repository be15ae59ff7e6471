"""Per-call time of the edge / triplet-row kernels of the narrow path at the RNA B=8 shape (HIP events around repeated
C-ABI calls): global message fwd / bwd (E_g rows), mlp_sbf fwd / bwd (T+P rows)."""
import os, sys, ctypes
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch
from pamnet_amd import lib, narrow
dev = torch.device('cuda:0')
lib.load()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, eg, tp = 17700, 867252, 669280


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


torch.manual_seed(0)
P = lib.ptr
tgt = torch.sort(torch.randint(0, n, (eg,), device=dev)).values.to(torch.int32)
src = (tgt + torch.randint(-40, 40, (eg,), device=dev).to(torch.int32)).clamp_(0, n - 1)
e = torch.randn(eg, d, device=dev); Pn = torch.randn(n, 2 * d, device=dev)
wm = torch.randn(d, 3 * d, device=dev) * 0.2; bm = torch.randn(d, device=dev); wea = torch.randn(d, d, device=dev) * 0.2
msg, dz, de = torch.empty_like(e), torch.empty_like(e), torch.empty_like(e)
dagg = torch.randn(n, d, device=dev)
partial = torch.empty(256, 2 * d * d + 2 * d, device=dev)
dwe, dwea, db = torch.empty(d, d, device=dev), torch.empty(d, d, device=dev), torch.empty(d, device=dev)
st = lib.stream_of(e)
we = wm.data_ptr() + 4 * 2 * d
t_gf = timeit(lambda: lib.call('pamnet_narrow_global_fwd_f32', P(e), eg, d, P(tgt), P(src), P(Pn), we, 3 * d, P(bm), P(wea), d, P(msg), st))
t_gb = timeit(lambda: lib.call('pamnet_narrow_global_bwd_f32', P(e), eg, d, P(tgt), P(src), P(Pn), we, 3 * d, P(bm), P(wea), d, P(dagg), P(dz), P(de), P(partial), P(dwe), P(dwea), P(db), st))
x = torch.randn(tp, d, device=dev); g = torch.randn(tp, d, device=dev)
w1, w2 = torch.randn(d, d, device=dev) * 0.3, torch.randn(d, d, device=dev) * 0.3
b1, b2 = torch.randn(d, device=dev), torch.randn(d, device=dev)
y, dx = torch.empty_like(x), torch.empty_like(x)
dw, dbb = torch.empty(2, d, d, device=dev), torch.empty(2, d, device=dev)
t_mf = timeit(lambda: lib.call('pamnet_narrow_mlp2_fwd_f32', P(x), tp, d, P(w1), P(b1), P(w2), P(b2), 0, None, P(y), st))
t_mb = timeit(lambda: lib.call('pamnet_narrow_mlp2_bwd_f32', P(x), tp, d, P(w1), P(b1), P(w2), P(b2), P(g), 0, P(dx), P(partial), P(dw), P(dbb), st))
fl_g = eg * d * d * 2 * 2 / 1e12
fl_m = tp * d * d * 2 * 2 / 1e12
print('d=%d  global fwd %6.1f us (%4.1f TF)  bwd %6.1f us (%4.1f TF) | mlp2 fwd %6.1f us (%4.1f TF)  bwd %6.1f us (%4.1f TF)   env %s' % (
    d, t_gf, fl_g / t_gf * 1e6, t_gb, 3 * fl_g / t_gb * 1e6, t_mf, fl_m / t_mf * 1e6, t_mb, 3 * fl_m / t_mb * 1e6,
    {k: v for k, v in os.environ.items() if k.startswith('PAMNET_N')}))

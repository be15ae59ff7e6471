"""Per-call time of the narrow edge-embedding kernels at the RNA B=8 shape (HIP events around repeated C-ABI calls):
Bessel-row embedding fwd / bwd on E_g edge lengths, spherical-basis embedding fwd / bwd on T+P rows (two weight sets)."""
import math
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    sys.path.insert(0, p)
import torch
from pamnet_amd import lib, narrow
dev = torch.device('cuda:0')
lib.load()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eg, tp = 867252, 669280


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


torch.manual_seed(0)
P = lib.ptr
dist = (torch.rand(eg, device=dev) * 19.5 + 0.4).contiguous()
freq = (torch.arange(1, 17, device=dev, dtype=torch.float32) * math.pi).contiguous()
w, b = torch.randn(d, 16, device=dev) * 0.3, torch.randn(d, device=dev)
y, g = torch.empty(eg, d, device=dev), torch.randn(eg, d, device=dev)
partial = torch.empty(narrow._blocks(eg), d * 16 + d + 16, device=dev)
dw, dbf = torch.empty(d, 16, device=dev), torch.empty(d + 16, device=dev)
st = lib.stream_of(dist)
t_f = timeit(lambda: lib.call('pamnet_narrow_embed_rbf_fwd_f32', P(dist), P(freq), 20.0, eg, d, P(w), P(b), P(y), st))
t_b = timeit(lambda: lib.call('pamnet_narrow_embed_rbf_bwd_f32', P(dist), P(freq), 20.0, eg, d, P(w), P(b), P(g), P(partial),
                              P(dw), P(dbf), st))
print('d=%d Bessel-row embedding, %d edges: fwd %.1f us (%.0f GB/s of y), bwd %.1f us (%.0f GB/s of dy)' % (
    d, eg, t_f, eg * d * 4 / t_f / 1e3, t_b, eg * d * 4 / t_b / 1e3))
F = torch.randn(tp, 42, device=dev)
kind = (torch.rand(tp, device=dev) < 0.5).to(torch.int32)
wa, wb = torch.randn(d, 42, device=dev) * 0.2, torch.randn(d, 42, device=dev) * 0.2
ba, bb = torch.randn(d, device=dev), torch.randn(d, device=dev)
y2, g2 = torch.empty(tp, d, device=dev), torch.randn(tp, d, device=dev)
partial2 = torch.empty(narrow._blocks(tp), 2 * (d * 48 + d), device=dev)
dw2, db2 = torch.empty(2, d, 42, device=dev), torch.empty(2, d, device=dev)
t_f2 = timeit(lambda: lib.call('pamnet_narrow_embed_fwd_f32', P(F), tp, 42, d, P(kind), P(wa), P(ba), P(wb), P(bb), P(y2), st))
t_b2 = timeit(lambda: lib.call('pamnet_narrow_embed_bwd_f32', P(F), tp, 42, d, P(kind), P(wa), P(ba), P(wb), P(bb), P(g2), None,
                               P(partial2), P(dw2), P(db2), st))
print('d=%d spherical-basis embedding, %d rows: fwd %.1f us (%.0f GB/s of F + y), bwd %.1f us (%.0f GB/s of F + dy)' % (
    d, tp, t_f2, tp * (42 + d) * 4 / t_f2 / 1e3, t_b2, tp * (42 + d) * 4 / t_b2 / 1e3))
print('checksums', float(dw.double().sum()), float(dbf.double().sum()), float(dw2.double().sum()), float(db2.double().sum()),
      float(y.double().sum()), float(y2.double().sum()))

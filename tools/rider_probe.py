#!/usr/bin/env python
"""Is the riding triplet / pair MLP what a forward chain launch waits for?  pamnet_node_tail_fwd_rider_f32 at the QM9 batch (2 286
rows = 143 chain workgroups, next head with 4 blocks) without riders and with half / a quarter / all of the 1 103 MLP tiles riding
on the 113 idle CUs; HIP-event timed.  Run on the GPU box: python tools/rider_probe.py"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import lib  # noqa: E402

dev = torch.device('cuda:0')
D, n, tp = 128, 2286, 17640
P = ctypes.c_void_p
rnd = lambda *s: torch.randn(*s, device=dev) * 0.05
x2, rx = rnd(n, D), rnd(n, D)
NW = 15
W = [rnd(D, D) for _ in range(NW)]
b = [torch.zeros(D, device=dev) for _ in range(11)]
w_out, b_out, w_att = rnd(D), torch.zeros(1, device=dev), rnd(D)
Z, R, xo = torch.empty(10, n, D, device=dev), torch.empty(2, n, D, device=dev), torch.empty(n, D, device=dev)
zx1, x1, Pn = torch.empty(n, D, device=dev), torch.empty(n, D, device=dev), torch.empty(4, n, D, device=dev)
st = torch.cuda.current_stream().cuda_stream
PACKED = int(os.environ.get('PAMNET_PROBE_PACKED', '2'))      # 1: fp32 fragment images (fp32-MFMA chain), 2: bf16x3 (bf16x6 chain)
images = torch.empty(NW, D * D * 3 // 2, device=dev)
lib.call('pamnet_pack_weights_f32' if PACKED == 1 else 'pamnet_pack_weights_bf16x3', NW, (P * NW)(*[t.data_ptr() for t in W]),
         (ctypes.c_int64 * NW)(*([D] * NW)), 0, lib.ptr(images), st)
img = [images[i].data_ptr() for i in range(NW)]
sbf = rnd(tp, D)
M = [rnd(D, D), rnd(D), rnd(D, D), rnd(D)]
mo = [torch.empty(tp, D, device=dev) for _ in range(3)]
PA10 = P * 10


def run(tile0, ntiles, wgs):
    lib.call('pamnet_node_tail_fwd_rider_f32', lib.ptr(x2), lib.ptr(rx), n, PA10(*img[:10]), PA10(*[t.data_ptr() for t in b[:10]]),
             lib.ptr(w_out), lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(xo), img[10], lib.ptr(b[10]),
             (P * 4)(*img[11:15]), D, 4, lib.ptr(zx1), lib.ptr(x1), lib.ptr(Pn), lib.ptr(sbf), tp, tile0, ntiles,
             (P * 4)(*[t.data_ptr() for t in M]), (P * 3)(*[t.data_ptr() for t in mo]), wgs, PACKED, st)


def event_us(fn, reps=50, groups=5):
    fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(groups):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        t.append(a.elapsed_time(e) * 1e3 / reps)
    return sorted(t)[len(t) // 2]


tiles = (tp + 15) // 16
print('packed = %d' % PACKED)
for name, nt in (('no riders', 0), ('a quarter of the MLP tiles riding', tiles // 4), ('half (the engine\'s split)', tiles - tiles // 2),
                 ('all of them', tiles)):
    print('%-40s %6.1f us' % (name, event_us(lambda: run(0, nt, 256 - (n + 15) // 16))))

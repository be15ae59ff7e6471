#!/usr/bin/env python
"""How long the per-direction weight-image pack launch takes (pamnet_pack_weights_mixed_f32, 168 images = 6 layer pairs):
fp32 fragment images against bf16x3 ones, forward and transposed orientation.  Run on the GPU box."""
import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo/physics-aware-multiplex-gnn_amd')
from pamnet_amd import lib
lib.load()
dev = torch.device('cuda:0')
D = 128
N = 168
W = [torch.randn(D, D, device=dev) for _ in range(N)]
P = ctypes.c_void_p
Wp = (P * N)(*[w.data_ptr() for w in W])
ld = (ctypes.c_int64 * N)(*([D] * N))
img = torch.empty(N * 24576, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for tr in (0, 1):
    for kind in (0, 1):
        k = (ctypes.c_int32 * N)(*([kind] * N))
        off = (ctypes.c_int64 * N)(*[i * 24576 for i in range(N)])
        us = t(lambda: lib.call('pamnet_pack_weights_mixed_f32', N, Wp, ld, k, off, tr, lib.ptr(img), st))
        print('pack %d images, transposed=%d kind=%d: %.1f us' % (N, tr, kind, us))

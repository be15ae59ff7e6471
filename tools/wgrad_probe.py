#!/usr/bin/env python
"""Cycle breakdown of the batched weight-gradient kernel for the middle workgroup (global-layer job mix of the QM9 B=128
step: 13 node-level jobs of N rows + 2 edge-level jobs of E_g rows).  Run on the GPU box."""
import ctypes
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
f32 = False
so = '/tmp/libpamnet_wgprobe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       '-DPAMNET_PHASE_PROBE', '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                       os.path.join(CSRC, 'wgrad.hip'), '-o', so])
lib = ctypes.CDLL(so)
N, EG = 2286, 32888
dev = torch.device('cuda:0')
rows = [N] * 13 + [EG] * 2
nj = len(rows)
dZ = [torch.randn(r, 128, device=dev) for r in rows]
A = [torch.randn(r, 128, device=dev) for r in rows]
dW = [torch.empty(128, 128, device=dev) for _ in rows]
db = [torch.empty(128, device=dev) for _ in rows]
I64 = ctypes.c_int64 * nj
PA = ctypes.c_void_p * nj
need = ctypes.c_int64(0)
lib.pamnet_wgrad_scratch_floats(ctypes.c_int64(nj), I64(*rows), ctypes.byref(need))
partial = torch.empty(need.value, device=dev)
P = ctypes.c_void_p
lib.pamnet_wgrad_batched_f32.argtypes = [ctypes.c_int64, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_int64, P, P, P, P]
args = (nj, PA(*[t.data_ptr() for t in dZ]), I64(*[128] * nj), PA(*[t.data_ptr() for t in A]), I64(*[128] * nj),
        (ctypes.c_int32 * nj)(*[0] * nj), I64(*rows), PA(*[t.data_ptr() for t in dW]), I64(*[128] * nj),
        PA(*[t.data_ptr() for t in db]), partial.data_ptr(), None, 0, None, None, None,
        torch.cuda.current_stream().cuda_stream)
import torch as _t
ref = [(z.double().t() @ a.double()) for z, a in zip(dZ, A)]
for it in range(3):
    for _ in range(50 if it else 1):
        assert lib.pamnet_wgrad_batched_f32(*args) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    assert lib.pamnet_wgrad_batched_f32(*args) == 0
    e.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.pamnet_wgrad_probe_read(buf)
    t = list(buf)
    print('run %d: 3 kernels %.1f us (events)' % (it, s.elapsed_time(e) * 1e3))
    print('  prologue (first block: fetch, split, store, barrier): %d cycles' % (t[1] - t[0]))
    print('  block: fragments+barrier+MFMAs(+split of next)   store pieces+fetch   barrier')
    k = 0
    while 5 + 3 * k < 64 and t[5 + 3 * k] > t[2 + 3 * k] > 0:
        print('  %2d    %8d %8d %8d' % (k, t[3 + 3 * k] - t[2 + 3 * k], t[4 + 3 * k] - t[3 + 3 * k], t[5 + 3 * k] - t[4 + 3 * k]))
        k += 1
    if it == 2:
        errs = [float((w.double() - r).abs().max() / r.abs().max()) for w, r in zip(dW, ref)]
        e32 = [float(((z.t() @ a).double() - r).abs().max() / r.abs().max()) for z, a, r in zip(dZ, A, ref)]
        print('  max|dW - fp64| / max|fp64|: %.2e (N-row jobs) %.2e (E_g-row jobs); torch fp32 matmul: %.2e / %.2e' % (
            max(errs[:13]), max(errs[13:]), max(e32[:13]), max(e32[13:])))
    print('  epilogue (tile transpose + 64 KB partial store): %d cycles; whole workgroup %d cycles' % (
        t[3 + 3 * k] - t[2 + 3 * k], t[3 + 3 * k] - t[0]))

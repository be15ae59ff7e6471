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
    print('  iter: regs->LDS(+load wait)  barrier   fetch issue+bias+MFMA   barrier')
    k = 0
    while 4 * k + 4 < 64 and t[4 * k + 4] > t[4 * k] > 0:
        print('  %2d    %8d %8d %8d %8d' % (k, t[4 * k + 1] - t[4 * k], 0, t[4 * k + 2] - t[4 * k + 1], t[4 * k + 3] - t[4 * k + 2]))
        k += 1
    print('  epilogue (tile transpose + 64 KB partial store): %d cycles; whole workgroup %d cycles' % (
        t[4 * k + 1] - t[4 * k], t[4 * k + 1] - t[0]))

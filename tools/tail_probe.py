#!/usr/bin/env python
"""Cycle breakdown of the 10-layer node chain (forward) for the middle workgroup; see tools/phase_probe.py."""
import ctypes
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
so = '/tmp/libpamnet_tailprobe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       '-DPAMNET_PHASE_PROBE', '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                       os.path.join(CSRC, 'node_tail.hip'), '-o', so])
lib = ctypes.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2286
dev = torch.device('cuda:0')
x2, rx = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
W = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(10)]
b = [torch.zeros(128, device=dev) for _ in range(10)]
w_out, b_out, w_att = torch.randn(128, device=dev), torch.zeros(1, device=dev), torch.randn(128, device=dev)
Z, R = torch.empty(10, n, 128, device=dev), torch.empty(2, n, 128, device=dev)
xo, out, att = torch.empty(n, 128, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
PA = ctypes.c_void_p * 10
P = ctypes.c_void_p
lib.pamnet_node_tail_fwd_f32.argtypes = [P, P, ctypes.c_int64, P, P, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_int64,
                                         ctypes.c_int64, P, P, P, ctypes.c_int32, P]
Wp, bp = PA(*[t.data_ptr() for t in W]), PA(*[t.data_ptr() for t in b])
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    for _ in range(100 if it else 1):
        rc = lib.pamnet_node_tail_fwd_f32(x2.data_ptr(), rx.data_ptr(), n, Wp, bp, w_out.data_ptr(), b_out.data_ptr(),
                                          w_att.data_ptr(), Z.data_ptr(), R.data_ptr(), xo.data_ptr(), out.data_ptr(),
                                          att.data_ptr(), None, None, None, 0, 0, None, None, None, 0, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.pamnet_tail_probe_read(buf)
    t = list(buf)
    print('run %d: chain total %d cycles' % (it, t[39] - t[0]))
    print('  layer:   wait+MFMA   prefetch+epilogue   barrier')
    for k in range(10):
        print('  %2d      %8d   %8d            %8d' % (k, t[4 * k + 1] - t[4 * k], t[4 * k + 2] - t[4 * k + 1], t[4 * k + 3] - t[4 * k + 2]))

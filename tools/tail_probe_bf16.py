#!/usr/bin/env python
"""Cycle breakdown of the 7-layer forward node chain for the middle workgroup, fp32-MFMA form (packed fp32 images) against
the bf16x6 form (bf16x3 images): see tools/tail_probe.py.  Run on the GPU box."""
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
# (node_tail.hip calls entry points of other translation units: resolved from the product library)
ctypes.CDLL(os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'pamnet_amd', 'libpamnet_hip.so'), mode=ctypes.RTLD_GLOBAL)
lib = ctypes.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2286
dev = torch.device('cuda:0')
x2, rx = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
W = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(10)]
b = [torch.zeros(128, device=dev) for _ in range(10)]
w_out, b_out, w_att = torch.randn(128, device=dev), torch.zeros(1, device=dev), torch.randn(128, device=dev)
Z, R = torch.empty(10, n, 128, device=dev), torch.empty(2, n, 128, device=dev)
xo = torch.empty(n, 128, device=dev)
PA = ctypes.c_void_p * 10
P = ctypes.c_void_p
I64 = ctypes.c_int64 * 10
lib.pamnet_node_tail_fwd_f32.argtypes = [P, P, ctypes.c_int64, P, P, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_int64,
                                         ctypes.c_int64, P, P, P, ctypes.c_int32, P]
st = torch.cuda.current_stream().cuda_stream
Wp = PA(*[t.data_ptr() for t in W])
bp = PA(*[t.data_ptr() for t in b])
ld = I64(*[128] * 10)
results = {}
for packed, name in ((1, 'fp32 MFMA, fp32 fragment images'), (2, 'bf16x6, bf16x3 images')):
    img = torch.empty(10 * 24576, device=dev)
    fn = lib.pamnet_pack_weights_f32 if packed == 1 else lib.pamnet_pack_weights_bf16x3
    fn.argtypes = [ctypes.c_int64, P, P, ctypes.c_int32, P, P]
    assert fn(10, Wp, ld, 0, img.data_ptr(), st) == 0
    stride = 16384 if packed == 1 else 24576
    Ip = PA(*[img.data_ptr() + 4 * stride * k for k in range(10)])
    for it in range(3):
        for _ in range(100 if it else 1):
            rc = lib.pamnet_node_tail_fwd_f32(x2.data_ptr(), rx.data_ptr(), n, Ip, bp, w_out.data_ptr(), b_out.data_ptr(),
                                              w_att.data_ptr(), Z.data_ptr(), R.data_ptr(), xo.data_ptr(), None, None,
                                              None, None, None, 0, 0, None, None, None, packed, st)
            assert rc == 0, rc
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        lib.pamnet_node_tail_fwd_f32(x2.data_ptr(), rx.data_ptr(), n, Ip, bp, w_out.data_ptr(), b_out.data_ptr(),
                                     w_att.data_ptr(), Z.data_ptr(), R.data_ptr(), xo.data_ptr(), None, None, None, None,
                                     None, 0, 0, None, None, None, packed, st)
        e.record()
        torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.pamnet_tail_probe_read(buf)
    t = list(buf)
    print('%s: launch %.1f us; layers 0..6 of the middle workgroup %d cycles' % (name, s.elapsed_time(e) * 1e3, t[27] - t[0]))
    print('  layer:   wait+MFMA   prefetch+epilogue   barrier')
    for k in range(7):
        print('  %2d      %8d   %8d            %8d' % (k, t[4 * k + 1] - t[4 * k], t[4 * k + 2] - t[4 * k + 1], t[4 * k + 3] - t[4 * k + 2]))
    results[packed] = (xo.clone(), Z[:7].clone())
x64 = x2.double()
print('x_out: max|bf16x6 - fp32 MFMA| / max = %.2e;  z_k: %.2e' % (
    float((results[2][0] - results[1][0]).abs().max() / results[1][0].abs().max()),
    float((results[2][1] - results[1][1]).abs().max() / results[1][1].abs().max())))

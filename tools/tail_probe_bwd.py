#!/usr/bin/env python
"""Cycle breakdown of the 7-layer backward node chain for the middle workgroup (csrc/node_tail.hip node_tail_bwd_kernel on fp32
MFMAs against node_tail_bwd_bf16_kernel on bf16x6 piece products): shader-clock stamps, a private -DPAMNET_PHASE_PROBE build.
Run on the GPU box: python tools/tail_probe_bwd.py [n]"""
import ctypes
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
so = '/tmp/libpamnet_tailprobe_bwd.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       '-DPAMNET_PHASE_PROBE', '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                       os.path.join(CSRC, 'node_tail.hip'), '-o', so])
# (node_tail.hip calls entry points of other translation units: resolved from the product library)
ctypes.CDLL(os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'pamnet_amd', 'libpamnet_hip.so'), mode=ctypes.RTLD_GLOBAL)
lib = ctypes.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2286
dev = torch.device('cuda:0')
P = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
W = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(7)]
Wp = (P * 7)(*[t.data_ptr() for t in W])
ld = (ctypes.c_int64 * 7)(*([128] * 7))
Z, dZ = torch.randn(10, n, 128, device=dev), torch.empty(10, n, 128, device=dev)
d_xout, g_head = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
dx2, drx = torch.empty(n, 128, device=dev), torch.empty(n, 128, device=dev)
lib.pamnet_node_tail_main_bwd_f32.argtypes = [P, P, ctypes.c_int64, P, P, P, P, P, ctypes.c_int32, P]
out = {}
for packed, name, fn, stride in ((1, 'fp32 MFMA, fp32 fragment images', 'pamnet_pack_weights_f32', 16384),
                                 (2, 'bf16x6, bf16x3 images', 'pamnet_pack_weights_bf16x3', 24576)):
    images = torch.empty(7, stride, device=dev)
    f = getattr(lib, fn)
    f.argtypes = [ctypes.c_int64, P, P, ctypes.c_int32, P, P]
    assert f(7, Wp, ld, 1, images.data_ptr(), st) == 0
    img = (P * 7)(*[images[i].data_ptr() for i in range(7)])
    call = lambda: lib.pamnet_node_tail_main_bwd_f32(d_xout.data_ptr(), g_head.data_ptr(), n, img, Z.data_ptr(), dZ.data_ptr(),
                                                     dx2.data_ptr(), drx.data_ptr(), packed, st)
    assert call() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        call()
    e.record()
    e.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.pamnet_tail_probe_read(buf)
    t = list(buf)
    print('%s: launch %.1f us; layers 6..0 of the middle workgroup %d cycles' % (name, s.elapsed_time(e) * 10, t[3] - t[24]))
    print('  layer:   wait+MFMA   prefetch+epilogue   barrier')
    for k in range(6, -1, -1):
        print('  %2d      %8d   %8d            %8d' % (k, t[4 * k + 1] - t[4 * k], t[4 * k + 2] - t[4 * k + 1], t[4 * k + 3] - t[4 * k + 2]))
    out[packed] = (dx2.clone(), dZ[:7].clone())
a, b = out[1], out[2]
print('d_x2: max|bf16x6 - fp32 MFMA| / max = %.2e;  dZ: %.2e' % ((a[0] - b[0]).abs().max() / a[0].abs().max(),
                                                                  (a[1] - b[1]).abs().max() / a[1].abs().max()))

# ---- the production launch: the next head's backward in front (nblk planes read, no gathers), bf16x6
lib.pamnet_node_pre_tail_bwd_f32.argtypes = [P, P, P, ctypes.c_int64, P, P, ctypes.c_int64, P, P, P, P, P, P, P, P, P, P]
Wh = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(5)]
Whp = (P * 5)(*[t.data_ptr() for t in Wh])
ld5 = (ctypes.c_int64 * 5)(*([128] * 5))
himg = torch.empty(5, 24576, device=dev)
f = lib.pamnet_pack_weights_bf16x3
assert f(5, Whp, ld5, 1, himg.data_ptr(), st) == 0
cimg = torch.empty(7, 24576, device=dev)
assert f(7, Wp, ld, 1, cimg.data_ptr(), st) == 0
cw = (P * 7)(*[cimg[i].data_ptr() for i in range(7)])
dP, zx1, dzx1 = torch.randn(4, n, 128, device=dev), torch.randn(n, 128, device=dev), torch.empty(n, 128, device=dev)
for nblk in (2, 4):
    hw = (P * 4)(*[himg[1 + i].data_ptr() for i in range(4)])
    call = lambda: lib.pamnet_node_pre_tail_bwd_f32(dP.data_ptr(), dx2.data_ptr(), drx.data_ptr(), n, himg[0].data_ptr(), hw, nblk | 16,
                                                    zx1.data_ptr(), dzx1.data_ptr(), g_head.data_ptr(), cw, Z.data_ptr(), dZ.data_ptr(),
                                                    dx2.data_ptr(), drx.data_ptr(), None, st)
    assert call() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        call()
    e.record()
    e.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.pamnet_tail_probe_read(buf)
    t = list(buf)
    print('head backward (%d planes) + chain, bf16x6: launch %.1f us | staging %d | head GEMMs %d | chain + flush %d | total %d cycles' % (
        nblk, s.elapsed_time(e) * 10, t[41] - t[40], t[42] - t[41], t[43] - t[42], t[43] - t[40]))

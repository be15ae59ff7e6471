#!/usr/bin/env python
"""tools/tail_probe.py for the PRODUCTION form of the forward node chain: packed weight images, heads deferred (7 layers),
optionally the next layer's head on the tile.  Usage on the GPU box: python tools/tail_probe_packed.py [n] [nblk]"""
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
# (node_tail.hip calls entry points of other translation units -- the fused aggregation falls back to them: resolved from the product)
ctypes.CDLL(os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'pamnet_amd', 'libpamnet_hip.so'), mode=ctypes.RTLD_GLOBAL)
lib = ctypes.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2286
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')
x2, rx = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
NW = 10 + 1 + 4
W = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(NW)]
b = [torch.zeros(128, device=dev) for _ in range(11)]
w_out, b_out, w_att = torch.randn(128, device=dev), torch.zeros(1, device=dev), torch.randn(128, device=dev)
Z, R = torch.empty(10, n, 128, device=dev), torch.empty(2, n, 128, device=dev)
xo = torch.empty(n, 128, device=dev)
zx1, x1, Pn = torch.empty(n, 128, device=dev), torch.empty(n, 128, device=dev), torch.empty(4, n, 128, device=dev)
P = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
images = torch.empty(NW, 128 * 128, device=dev)
lib.pamnet_pack_weights_f32.argtypes = [ctypes.c_int64, P, P, ctypes.c_int32, P, P]
Wp = (P * NW)(*[t.data_ptr() for t in W])
ld = (ctypes.c_int64 * NW)(*([128] * NW))
assert lib.pamnet_pack_weights_f32(NW, Wp, ld, 0, images.data_ptr(), st) == 0
img = [images[i].data_ptr() for i in range(NW)]
PA = P * 10
lib.pamnet_node_tail_fwd_f32.argtypes = [P, P, ctypes.c_int64, P, P, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_int64,
                                         ctypes.c_int64, P, P, P, ctypes.c_int32, P]
Wi, bp = PA(*img[:10]), PA(*[t.data_ptr() for t in b[:10]])
wpn = (P * 4)(*img[11:15])
for it in range(3):
    for _ in range(100 if it else 1):
        rc = lib.pamnet_node_tail_fwd_f32(x2.data_ptr(), rx.data_ptr(), n, Wi, bp, w_out.data_ptr(), b_out.data_ptr(),
                                          w_att.data_ptr(), Z.data_ptr(), R.data_ptr(), xo.data_ptr(), None, None,
                                          img[10] if nblk else None, b[10].data_ptr() if nblk else None,
                                          wpn if nblk else None, 128, nblk, zx1.data_ptr() if nblk else None,
                                          x1.data_ptr() if nblk else None, Pn.data_ptr() if nblk else None, 1, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.pamnet_tail_probe_read(buf)
    t = list(buf)
    print('run %d: layers 0..6 total %d cycles' % (it, t[27] - t[0]))
    print('  kernel: staging %d | chain %d | park -> memory %d | next head (x1 + %d blocks) %d | its flush %d | total %d cycles' % (
        t[0] - t[40], t[27] - t[0], t[41] - t[27], nblk, t[42] - t[41], t[43] - t[42], t[43] - t[40]))
    print('  layer:   wait+MFMA   prefetch+epilogue   barrier')
    for k in range(7):
        print('  %2d      %8d   %8d            %8d' % (k, t[4 * k + 1] - t[4 * k], t[4 * k + 2] - t[4 * k + 1], t[4 * k + 3] - t[4 * k + 2]))

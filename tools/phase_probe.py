#!/usr/bin/env python
"""Where does a tile's time go?  Builds a private copy of csrc/edge_chain.hip with -DPAMNET_PHASE_PROBE (shader-clock
timestamps of the middle workgroup at phase boundaries), runs the 8-wave 2-layer MLP at the workload's row count and
prints the phase durations in shader cycles / us.  Run on the GPU box:  python tools/phase_probe.py [rows]"""
import ctypes
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
so = '/tmp/libpamnet_probe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       '-DPAMNET_PHASE_PROBE', '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                       os.path.join(CSRC, 'edge_chain.hip'), '-o', so])
os.environ['PAMNET_MLP2_W8'] = '1'
lib = ctypes.CDLL(so)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 17640
dev = torch.device('cuda:0')
x = torch.randn(rows, 128, device=dev)
W1, W2 = torch.randn(128, 128, device=dev) * 0.05, torch.randn(128, 128, device=dev) * 0.05
b1, b2 = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
z1, z2, y = (torch.empty(rows, 128, device=dev) for _ in range(3))
P = ctypes.c_void_p
lib.pamnet_mlp2_fwd_f32.argtypes = [P, ctypes.c_int64, P, P, P, P, P, P, P, P]
st = torch.cuda.current_stream().cuda_stream
names = ['loads issued -> x tile in LDS', 'GEMM1 (MFMA)', 'W2 prefetch issue + acc->LDS + barrier', 'SiLU sweep + z1 store',
         'GEMM2 (MFMA)', 'acc->LDS + barrier', 'final sweep (z2, y stores issued)']
for it in range(4):
    if it >= 2:                      # probe the last launch of a back-to-back burst: clocks under sustained load
        for _ in range(300):
            lib.pamnet_mlp2_fwd_f32(x.data_ptr(), rows, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                    z1.data_ptr(), z2.data_ptr(), y.data_ptr(), st)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = lib.pamnet_mlp2_fwd_f32(x.data_ptr(), rows, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                 z1.data_ptr(), z2.data_ptr(), y.data_ptr(), st)
    e.record()
    torch.cuda.synchronize()
    assert rc == 0
    buf = (ctypes.c_longlong * 32)()
    lib.pamnet_probe_read(buf)
    t = [buf[i] for i in range(8)]
    wall = (buf[16 + 7] - buf[16]) / 100.0                      # us
    print('   shader clock during this workgroup: %.2f GHz  (workgroup lifetime %.1f us)' % (
        (t[7] - t[0]) / wall / 1e3 if wall > 0 else float('nan'), wall))
    print('run %d: kernel %.1f us (event)  workgroup total %d cycles' % (it, s.elapsed_time(e) * 1e3, t[7] - t[0]))
    for i, n in enumerate(names):
        print('   %-44s %7d cycles' % (n, t[i + 1] - t[i]))

    t16 = (rows + 15) // 16
    per = (t16 + 255) // 256
    n = (t16 + per - 1) // per
    wg = (ctypes.c_longlong * (2 * n))()
    lib.pamnet_probe_read_wg(wg, n)
    st0 = min(wg[2 * i] for i in range(n))
    starts = sorted((wg[2 * i] - st0) / 100.0 for i in range(n))
    ends = sorted((wg[2 * i + 1] - st0) / 100.0 for i in range(n))
    life = sorted((wg[2 * i + 1] - wg[2 * i]) / 100.0 for i in range(n))
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print('   %d workgroups: start  p0 %.1f p50 %.1f p90 %.1f max %.1f us | end p50 %.1f max %.1f us | lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (
        n, starts[0], q(starts, .5), q(starts, .9), starts[-1], q(ends, .5), ends[-1], q(life, .1), q(life, .5), q(life, .9), life[-1]))

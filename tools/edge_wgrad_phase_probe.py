#!/usr/bin/env python
"""Where a workgroup of the fused global-edge backward + weight-gradient kernel (csrc/edge_agg.hip
global_edge_agg_bwd_wg_kernel) spends a 32-row chunk: private -DPAMNET_PHASE_PROBE build, shader-clock timestamps of the
middle workgroup's last chunk (waves 0 and 4).  Run on the GPU box: python tools/edge_wgrad_phase_probe.py [qm9|pdbbind]"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import graph as G, lib, synth  # noqa: E402

CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
so = '/tmp/libpamnet_ewgprobe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       '-DPAMNET_PHASE_PROBE'] + os.environ.get('PAMNET_PROBE_FLAGS', '').split() + [
                       '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                       os.path.join(CSRC, 'edge_agg.hip'), '-o', so])
lib.load()
plib = ctypes.CDLL(so)
dev = torch.device('cuda:0')
D = 128
if len(sys.argv) > 1 and sys.argv[1] == 'qm9':
    b = synth.qm9_batch(0, 0, 128).to(dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=128)
else:
    b = synth.pdbbind_batch(0, 0, 32).to(dev)
    g = G.build_graph('PDBbind', 2.0, 6.0, 'source_to_target', b.x, b.batch, num_graphs=32)
n, eg = g.n, g.glob.m
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
Wm, Wea = rnd(D, 3 * D) / 8, rnd(D, D) / 8
e, z, ea, d_agg = rnd(eg, D), rnd(eg, D), rnd(eg, D), rnd(n, D)
dz, d_e, dPi = torch.empty(eg, D, device=dev), torch.zeros(eg, D, device=dev), torch.empty(n, D, device=dev)
csr = g.glob
st = torch.cuda.current_stream().cuda_stream
cuts = torch.empty(257, dtype=torch.int32, device=dev)
lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts), None, st)
need, slots = ctypes.c_int64(0), ctypes.c_int64(0)
lib.call('pamnet_global_edge_agg_wg_floats', eg, ctypes.addressof(need), ctypes.addressof(slots))
partial = torch.empty(int(need.value), device=dev)
P, I = ctypes.c_void_p, ctypes.c_int64
fb = plib.pamnet_global_edge_agg_bwd_wg_f32
fb.argtypes = [P, I, I, P, P, P, P, P, P, P, I, P, I, P, P, ctypes.c_int32, P, P, P]
bw = lambda: fb(d_agg.data_ptr(), eg, n, csr.ptr.data_ptr(), csr.row_of.data_ptr(), cuts.data_ptr(), z.data_ptr(), ea.data_ptr(),
                e.data_ptr(), Wm.data_ptr() + 8 * D, 3 * D, Wea.data_ptr(), D, dz.data_ptr(), d_e.data_ptr(), 1, dPi.data_ptr(),
                partial.data_ptr(), st)
for _ in range(50):
    assert bw() == 0
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
assert bw() == 0
t.record()
torch.cuda.synchronize()
nwg = min(256, (eg + 15) // 16)
buf = (ctypes.c_longlong * 64)()
wg = (ctypes.c_longlong * (2 * nwg))()
plib.pamnet_agg_probe_read(buf, wg, nwg)
print('N=%d E_g=%d: kernel %.1f us (event), %.1f chunks of 32 rows per workgroup' % (n, eg, s.elapsed_time(t) * 1e3, eg / nwg / 32.0))
names = ['plan + CSR offsets', 'sweep: gather d_agg, dz/dea math, dz store, 3 splits -> images', 'prefetch + accumulate-operand issue',
         'barrier', 'node sums d P_i (3 MFMA per 16 nodes)', 'dW k-step (96 MFMA / wave)', 'dX GEMMs (96 MFMA / wave)', 'acc -> LDS, barrier',
         'd_e sweep (+ accumulate, store)', 'end barrier']
for w in (0, 1):
    v = [buf[32 * w + i] for i in range(32)]
    print(' wave %d (last chunk of the middle workgroup):' % (4 * w))
    for i, nm in enumerate(names):
        print('   %-66s %7d cycles' % (nm, v[i + 2] - v[i + 1]))
    print('   chunk total %d cycles; whole workgroup %d cycles; epilogue (partial tiles) %d' % (v[11] - v[1], v[12] - v[0], v[12] - v[11]))
life = sorted((wg[2 * i + 1] - wg[2 * i]) / 100.0 for i in range(nwg))
print(' lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (life[nwg // 10], life[nwg // 2], life[9 * nwg // 10], life[-1]))

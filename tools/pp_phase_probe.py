#!/usr/bin/env python
"""Round 6: where do the waves of the ping-pong forward edge kernel spend an iteration?  Private build of csrc/edge_agg.hip with
-DPAMNET_PHASE_PROBE; shader-clock stamps of every wave of the middle workgroup in iteration PAMNET_PROBE_T (default 20) at
the two phase ends and barrier releases.  GPU box:  python tools/pp_phase_probe.py [pdbbind|qm9]   (PAMNET_PROBE_FLAGS=-D...)"""
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
so = '/tmp/libpamnet_ppprobe.so'
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
Wm, bm, Wea = rnd(D, 3 * D) / 8, rnd(D), rnd(D, D) / 8
e, Pi, Pj, x1 = rnd(eg, D), rnd(n, D), rnd(n, D), rnd(n, D)
z, ea, out = torch.empty(eg, D, device=dev), torch.empty(eg, D, device=dev), torch.empty(n, D, device=dev)
csr = g.glob
P, I = ctypes.c_void_p, ctypes.c_int64
fn = plib.pamnet_global_edge_agg_fwd_pp_f32
fn.argtypes = [P, I, I, P, I, P, P, I, P, P, P, P, P, P, P, P, P, P, P]
st = torch.cuda.current_stream().cuda_stream
cuts = torch.empty(257, dtype=torch.int32, device=dev)
lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts), None, st)


def call(save):
    return fn(e.data_ptr(), eg, n, Wm.data_ptr() + 8 * D, 3 * D, bm.data_ptr(), Wea.data_ptr(), D, Pi.data_ptr(), Pj.data_ptr(),
              csr.ptr.data_ptr(), csr.row_of.data_ptr(), csr.col.data_ptr(), cuts.data_ptr(), x1.data_ptr(),
              z.data_ptr() if save else None, ea.data_ptr() if save else None, out.data_ptr(), st)


print('N=%d E_g=%d  (rows per workgroup ~%d = %d groups of 32)' % (n, eg, eg // 256, eg // 256 // 32))
for save in (True, False):
    for _ in range(20):
        call(save)
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    assert call(save) == 0
    t.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    wg = (ctypes.c_longlong * 512)()
    plib.pamnet_agg_probe_read(buf, wg, 256)
    print('save=%s kernel %.1f us (event, with the probe stores)' % (save, s.elapsed_time(t) * 1e3))
    base = min(buf[8 * w] for w in range(8))
    for w in range(8):
        v = [buf[8 * w + i] for i in range(5)]
        role = 'walker' if w == 3 else ('worker B' if w < 3 else 'worker A')
        first = 'GEMM' if w < 4 else 'vector'
        second = ('walk' if w == 3 else 'vector') if w < 4 else 'GEMM'
        print('  wave %d (%-8s): start +%5d | %-6s %6d | barrier %6d | %-6s %6d | barrier %6d | iteration %6d'
              % (w, role, v[0] - base, first, v[1] - v[0], v[2] - v[1], second, v[3] - v[2], v[4] - v[3], v[4] - v[0]), end='')
        if w != 3:
            x = [buf[8 * w + i] for i in (5, 6, 7)]
            v0 = v[2] if w < 4 else v[0]
            print('   vector = epilogue %5d + stage %5d + requests %5d' % (x[0] - v0, x[1] - x[0], x[2] - x[1]))
        else:
            print()

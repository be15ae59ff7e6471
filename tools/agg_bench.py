"""Fused edge MLP -> segment-sum kernels (csrc/edge_agg.hip) against the unfused pairs, HIP-event timed on the graph of
a real synthetic batch.  Usage on the GPU box:  python tools/agg_bench.py [qm9|pdbbind] [batch]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import graph as G, lib, synth  # noqa: E402
from pamnet_amd.ops import segment_sum_raw  # noqa: E402

dev = torch.device('cuda:0')
D = 128
kind = sys.argv[1] if len(sys.argv) > 1 else 'qm9'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (128 if kind == 'qm9' else 32)


def timeit(name, fn, reps=50, bytes_=None):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    print('%-52s %8.1f us%s' % (name, us, '  %.0f GB/s' % (bytes_ / us / 1e3) if bytes_ else ''), flush=True)
    return us


if kind == 'qm9':
    b = synth.qm9_batch(0, 0, B).to(dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=B)
else:
    b = synth.pdbbind_batch(0, 0, B).to(dev)
    g = G.build_graph('PDBbind', 2.0, 6.0, 'source_to_target', b.x, b.batch, num_graphs=B)
n, eg, el, tp = g.n, g.glob.m, g.loc.m, g.tp.m
print('N=%d E_g=%d E_l=%d T+P=%d' % (n, eg, el, tp))
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
Wm, bm, Wea = rnd(D, 3 * D) / 8, rnd(D), rnd(D, D) / 8
sub = lambda w, c0: w.data_ptr() + 4 * c0
st = lib.stream_of(Wm)
e, Pi, Pj, x1 = rnd(eg, D), rnd(n, D), rnd(n, D), rnd(n, D)
z, ea, msg = (torch.empty(eg, D, device=dev) for _ in range(3))
out = torch.empty(n, D, device=dev)
csr = g.glob
cuts_t = torch.empty(257, dtype=torch.int32, device=dev)
lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts_t), None, st)
cuts = lib.ptr(cuts_t)


def unfused(save):
    lib.call('pamnet_global_edge_fwd_f32', lib.ptr(e), eg, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D, lib.ptr(Pi),
             lib.ptr(Pj), lib.ptr(csr.row_of), lib.ptr(csr.col), lib.ptr(z) if save else None, lib.ptr(ea) if save else None,
             lib.ptr(msg), st)
    segment_sum_raw(out, x1, msg, None, None, None, None, csr.ptr, n, D)


def fusedk(save):
    lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), eg, n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
             lib.ptr(Pi), lib.ptr(Pj), lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(csr.col), cuts, lib.ptr(x1),
             lib.ptr(z) if save else None, lib.ptr(ea) if save else None, lib.ptr(out), st)


alg = 4.0 * D * eg + 8.0 * eg + 4.0 * D * n * 4        # e + indices + x1, Pi, Pj in, x2 out
timeit('global fwd unfused (edge + segsum), train', lambda: unfused(True))
timeit('global fwd fused, train (z, ea saved)', lambda: fusedk(True), bytes_=alg + 8.0 * D * eg)
timeit('global fwd unfused, inference', lambda: unfused(False))
timeit('global fwd fused, inference', lambda: fusedk(False), bytes_=alg)

d_agg = rnd(n, D)
dz, dea, d_e = (torch.empty(eg, D, device=dev) for _ in range(3))
dP = torch.empty(2, n, D, device=dev)
gT = g.glob_T


def bwd_unfused():
    lib.call('pamnet_global_edge_bwd_f32', lib.ptr(d_agg), lib.ptr(csr.row_of), eg, lib.ptr(z), lib.ptr(ea), sub(Wm, 2 * D),
             3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 1, st)
    segment_sum_raw(dP[0], None, dz, None, None, None, None, csr.ptr, n, D)
    segment_sum_raw(dP[1], None, dz, None, None, None, gT.perm, gT.ptr, n, D)


def bwd_fused():
    lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), eg, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), cuts, lib.ptr(z),
             lib.ptr(ea), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 1, lib.ptr(dP[0]), st)
    segment_sum_raw(dP[1], None, dz, None, None, None, gT.perm, gT.ptr, n, D)


import ctypes  # noqa: E402

need, slots = ctypes.c_int64(0), ctypes.c_int64(0)
lib.call('pamnet_global_edge_agg_wg_floats', eg, ctypes.addressof(need), ctypes.addressof(slots))
partial = torch.empty(int(need.value), device=dev)


def bwd_fused_wg():          # round 5: the backward with the step's own weight gradients formed in the kernel
    lib.call('pamnet_global_edge_agg_bwd_wg_f32', lib.ptr(d_agg), eg, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), cuts, lib.ptr(z),
             lib.ptr(ea), lib.ptr(e), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(d_e), 1, lib.ptr(dP[0]),
             lib.ptr(partial), st)
    segment_sum_raw(dP[1], None, dz, None, None, None, gT.perm, gT.ptr, n, D)


unfused(True)
timeit('global bwd unfused (edge + 2 segsums)', bwd_unfused)
timeit('global bwd fused (+ transposed segsum)', bwd_fused)
if eg < (1 << 23):
    timeit('global bwd fused WITH dW_e / dW_ea / db_m (+ transposed segsum)', bwd_fused_wg)

m_ji, m_nb, q3, s, mt, x2 = rnd(el, D), rnd(el, D), rnd(el, D), rnd(tp, D), torch.empty(el, D, device=dev), torch.empty(n, D, device=dev)
loc, tpc = g.loc, g.tp


def loc_unfused():
    segment_sum_raw(mt, m_ji, m_nb, tpc.col, s, None, None, tpc.ptr, el, D)
    segment_sum_raw(x2, x1, mt, None, q3, None, None, loc.ptr, n, D)


def loc_fused():
    lib.call('pamnet_local_agg_fwd_f32', lib.ptr(m_ji), lib.ptr(m_nb), lib.ptr(s), lib.ptr(q3), lib.ptr(tpc.ptr),
             lib.ptr(tpc.col), lib.ptr(loc.ptr), lib.ptr(x1), n, lib.ptr(mt), lib.ptr(x2), st)


timeit('local aggregations unfused (2 segsums)', loc_unfused)
timeit('local aggregations fused', loc_fused)

"""The fused global-edge backward WITH its own weight gradients (csrc/edge_agg.hip global_edge_agg_bwd_wg_kernel, round 5)
against what it replaces: the plain backward kernel + the two E_g-row jobs of a split-K weight-gradient launch.
HIP-event timed on the graph of a real synthetic batch.  Usage on the GPU box: python tools/edge_wgrad_probe.py [qm9|pdbbind] [batch]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import graph as G, lib, synth  # noqa: E402
from pamnet_amd.fused import DeferredWgrad  # noqa: E402

dev = torch.device('cuda:0')
D = 128
kind = sys.argv[1] if len(sys.argv) > 1 else 'pdbbind'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (128 if kind == 'qm9' else 32)


def timeit(name, fn, reps=30, bytes_=None):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    print('%-72s %8.1f us%s' % (name, us, '  %.2f TB/s of %.3f GB' % (bytes_ / us / 1e6, bytes_ / 1e9) if bytes_ else ''), flush=True)
    return us


if kind == 'qm9':
    b = synth.qm9_batch(0, 0, B).to(dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=B)
else:
    b = synth.pdbbind_batch(0, 0, B).to(dev)
    g = G.build_graph('PDBbind', 2.0, 6.0, 'source_to_target', b.x, b.batch, num_graphs=B)
n, eg = g.n, g.glob.m
print('N=%d E_g=%d' % (n, eg))
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
Wm, Wea = rnd(D, 3 * D) / 8, rnd(D, D) / 8
sub = lambda w, c0: w.data_ptr() + 4 * c0
st = lib.stream_of(Wm)
e, z, ea, d_agg = rnd(eg, D), rnd(eg, D), rnd(eg, D), rnd(n, D)
dz, dea, d_e = (torch.empty(eg, D, device=dev) for _ in range(3))
dPi = torch.empty(n, D, device=dev)
csr = g.glob
cuts_t = torch.empty(257, dtype=torch.int32, device=dev)
lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts_t), None, st)
cuts = lib.ptr(cuts_t)
need, slots = ctypes.c_int64(0), ctypes.c_int64(0)
lib.call('pamnet_global_edge_agg_wg_floats', eg, ctypes.addressof(need), ctypes.addressof(slots))
partial = torch.empty(int(need.value), device=dev)
gWm, gWea, gb = torch.empty(D, 3 * D, device=dev), torch.empty(D, D, device=dev), torch.empty(D, device=dev)
row = 4.0 * D


def plain():
    lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), eg, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), cuts, lib.ptr(z),
             lib.ptr(ea), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 1, lib.ptr(dPi), st)


def fused():
    lib.call('pamnet_global_edge_agg_bwd_wg_f32', lib.ptr(d_agg), eg, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), cuts, lib.ptr(z),
             lib.ptr(ea), lib.ptr(e), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(d_e), 1, lib.ptr(dPi),
             lib.ptr(partial), st)


dw = DeferredWgrad(d_agg)


def fused_and_reduce():
    fused()
    lib.call('pamnet_wgrad_edge_enqueue_f32', ctypes.addressof(dw.ctx), int(slots.value), sub(gWm, 2 * D), 3 * D, lib.ptr(gb),
             lib.ptr(gWea), D, lib.ptr(partial))
    dw.flush()


jobs = [(lib.ptr(dz), D, lib.ptr(e), D, 0, eg, sub(gWm, 2 * D), 3 * D, lib.ptr(gb)),
        (lib.ptr(dea), D, lib.ptr(e), D, 0, eg, lib.ptr(gWea), D, None)]


def wgrad_jobs():
    dw.launch(jobs)


a = timeit('plain backward kernel (dz, dea, d_e written)', plain, bytes_=row * eg * 6 + row * n * 2)
w = timeit('+ its two E_g weight-gradient jobs (one split-K launch, reduction deferred)', wgrad_jobs, bytes_=row * eg * 4)
dw.flush()
f = timeit('fused backward + weight gradients (partial tiles)', fused, bytes_=row * eg * 6 + row * n * 2 + 4.0 * need.value)
fr = timeit('fused + the fixed-order reduction of its tiles as a launch of its own', fused_and_reduce)
print('replaced: %.1f us -> %.1f us (%.1f with a stand-alone reduction)' % (a + w, f, fr))

"""Round 6: the fused global-edge forward, chunked form against the ping-pong form (csrc/edge_agg.hip
global_edge_agg_fwd_pp_kernel), HIP-event timed on the graph of a real synthetic batch, alternated.
Usage on the GPU box:  python tools/pp_probe.py [qm9|pdbbind] [batch]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import graph as G, lib, synth  # noqa: E402

dev = torch.device('cuda:0')
D = 128
kind = sys.argv[1] if len(sys.argv) > 1 else 'pdbbind'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (128 if kind == 'qm9' else 32)
if kind == 'qm9':
    b = synth.qm9_batch(0, 0, B).to(dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=B)
else:
    b = synth.pdbbind_batch(0, 0, B).to(dev)
    g = G.build_graph('PDBbind', 2.0, 6.0, 'source_to_target', b.x, b.batch, num_graphs=B)
n, eg = g.n, g.glob.m
print('%s B=%d: N=%d E_g=%d' % (kind, B, n, eg))
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
Wm, bm, Wea = rnd(D, 3 * D) / 8, rnd(D), rnd(D, D) / 8
sub = lambda w, c0: w.data_ptr() + 4 * c0
st = lib.stream_of(Wm)
e, Pi, Pj, x1 = rnd(eg, D), rnd(n, D), rnd(n, D), rnd(n, D)
csr = g.glob
cuts_t = torch.empty(257, dtype=torch.int32, device=dev)
lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts_t), None, st)


def run(entry, save, out, z, ea):
    lib.call(entry, lib.ptr(e), eg, n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D, lib.ptr(Pi), lib.ptr(Pj),
             lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(csr.col), lib.ptr(cuts_t), lib.ptr(x1),
             lib.ptr(z) if save else None, lib.ptr(ea) if save else None, lib.ptr(out), st)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    t.synchronize()
    return s.elapsed_time(t) / reps * 1e3


os.environ['PAMNET_AGG_PP'] = '0'
bufs = [[torch.empty(n, D, device=dev), torch.empty(eg, D, device=dev), torch.empty(eg, D, device=dev)] for _ in range(2)]
for save in (True, False):
    run('pamnet_global_edge_agg_fwd_f32', save, *bufs[0])
    run('pamnet_global_edge_agg_fwd_pp_f32', save, *bufs[1])
    torch.cuda.synchronize()
    same = torch.equal(bufs[0][0], bufs[1][0]) and (not save or (torch.equal(bufs[0][1], bufs[1][1]) and torch.equal(bufs[0][2], bufs[1][2])))
    alg = 4.0 * D * eg + 8.0 * eg + 4.0 * D * n * 4 + (8.0 * D * eg if save else 0.0)
    for rep in range(3):
        a = timeit(lambda: run('pamnet_global_edge_agg_fwd_f32', save, *bufs[0]))
        p = timeit(lambda: run('pamnet_global_edge_agg_fwd_pp_f32', save, *bufs[1]))
        print('%-10s chunked %7.1f us (%.2f TB/s)   ping-pong %7.1f us (%.2f TB/s)   bitwise equal: %s'
              % ('training' if save else 'inference', a, alg / a / 1e6, p, alg / p / 1e6, same), flush=True)

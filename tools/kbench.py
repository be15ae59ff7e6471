"""Micro-benchmarks of individual C-ABI kernels at the QM9 B=128 workload shapes (HIP-event timed).
Usage on the GPU box:  python tools/kbench.py [filter]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import fused, lib, modules  # noqa: E402

dev = torch.device('cuda:0')
D = 128
N, EG, EL, TP = 2286, 32888, 4316, 17640
FILT = sys.argv[1] if len(sys.argv) > 1 else ''


def timeit(name, fn, flops=None, bytes_=None, reps=30):
    if FILT and FILT not in name:
        return
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    extra = ''
    if flops:
        extra += '  %.1f TFLOP/s' % (flops / us / 1e6)
    if bytes_:
        extra += '  %.0f GB/s' % (bytes_ / us / 1e3)
    print('%-44s %9.1f us%s' % (name, us, extra), flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=dev)


def wgrad_case(name, specs, split_rows=512):
    """specs: list of (rows, a_mode, with_bias)"""
    jobs = []
    keep = []
    for rows, mode, wb in specs:
        dz, a = rnd(rows, D), rnd(rows, D)
        dw, db = rnd(D, D), rnd(D)
        keep += [dz, a, dw, db]
        jobs.append((dz, D, a, D, mode, rows, dw, D, db if wb else None))
    fl = sum(2.0 * r * D * D for r, _, _ in specs)
    by = sum(2.0 * r * D * 4 for r, _, _ in specs)
    timeit(name, lambda: fused.wgrad(jobs, keep[0]), fl, by)


wgrad_case('wgrad 1 job  N rows raw', [(N, 0, False)])
wgrad_case('wgrad 13 jobs N rows raw', [(N, 0, False)] * 13)
wgrad_case('wgrad 13 jobs N rows silu+bias', [(N, 1, True)] * 13)
wgrad_case('wgrad 2 jobs EG rows raw', [(EG, 0, False)] * 2)
wgrad_case('wgrad 2 jobs EG rows raw+bias', [(EG, 0, True)] * 2)
wgrad_case('wgrad global-layer mix (15 jobs)', [(N, 1, True)] * 13 + [(EG, 0, True), (EG, 0, False)])

# chains
layer = modules.GlobalMP(D).to(dev)
lloc = modules.LocalMP(D).to(dev)
tp = fused.tail_params(layer)
x2, rx = rnd(N, D), rnd(N, D)
timeit('node_tail_fwd N', lambda: fused.k_tail_fwd(x2, rx, tp), 10 * 2.0 * N * D * D)
Z = fused.k_tail_fwd(x2, rx, tp)[0]
gx, go, ga = rnd(N, D), rnd(N), rnd(N)
gw, gb, gatt = torch.empty(1, D, device=dev), torch.empty(1, device=dev), torch.empty(D, 1, device=dev)
timeit('node_tail_bwd N', lambda: fused.k_tail_bwd(gx, go, ga, tp, Z, gw, gb, gatt), 10 * 2.0 * N * D * D)
Wm = layer.mlp_m[0][0].weight
wps = [fused._sub(Wm, 0), fused._sub(Wm, D)]
lin = layer.mlp_x1[0][0]
timeit('node_pre_fwd N (2 blk)', lambda: fused.k_pre_fwd(x2, lin.weight, lin.bias, wps, 3 * D), 3 * 2.0 * N * D * D)

e = rnd(EG, D)
P = rnd(2, N, D)
row = torch.sort(torch.randint(0, N, (EG,), device=dev)).values.to(torch.int32)
col = torch.randint(0, N, (EG,), device=dev, dtype=torch.int32)
z, ea, msg = rnd(EG, D), rnd(EG, D), rnd(EG, D)
st = lib.stream_of(e)


def ge_fwd():
    lib.call('pamnet_global_edge_fwd_f32', lib.ptr(e), EG, fused._sub(Wm, 2 * D), 3 * D, lib.ptr(layer.mlp_m[0][0].bias),
             lib.ptr(layer.W_edge_attr.weight), D, lib.ptr(P[0]), lib.ptr(P[1]), lib.ptr(row), lib.ptr(col),
             lib.ptr(z), lib.ptr(ea), lib.ptr(msg), st)


timeit('global_edge_fwd EG', ge_fwd, 2 * 2.0 * EG * D * D, 6.0 * EG * D * 4)
dagg, dz, dea, de = rnd(N, D), rnd(EG, D), rnd(EG, D), rnd(EG, D)


def ge_bwd():
    lib.call('pamnet_global_edge_bwd_f32', lib.ptr(dagg), lib.ptr(row), EG, lib.ptr(z), lib.ptr(ea),
             fused._sub(Wm, 2 * D), 3 * D, lib.ptr(layer.W_edge_attr.weight), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(de),
             1, st)


timeit('global_edge_bwd EG', ge_bwd, 2 * 2.0 * EG * D * D, 7.0 * EG * D * 4)
xs = rnd(TP, D)
s1, s2 = lloc.mlp_sbf[0][0], lloc.mlp_sbf[1][0]
z1, z2, y = rnd(TP, D), rnd(TP, D), rnd(TP, D)
timeit('mlp2_fwd TP', lambda: lib.call('pamnet_mlp2_fwd_f32', lib.ptr(xs), TP, lib.ptr(s1.weight), lib.ptr(s1.bias),
                                       lib.ptr(s2.weight), lib.ptr(s2.bias), lib.ptr(z1), lib.ptr(z2), lib.ptr(y), st),
       2 * 2.0 * TP * D * D, 4.0 * TP * D * 4)
# plain segment sum at workload shape
ptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
ptr[1:] = torch.bincount(row.long(), minlength=N).cumsum(0).to(torch.int32)
out = rnd(N, D)
from pamnet_amd import ops  # noqa: E402
timeit('segment_sum EG->N', lambda: ops.segment_sum_raw(out, None, msg, None, None, None, None, ptr, N, D), None,
       (EG + N) * D * 4.0)

#!/usr/bin/env python
"""Where does a workgroup of the fused edge MLP -> segment-sum kernel spend its time?  Builds a private copy of
csrc/edge_agg.hip with -DPAMNET_PHASE_PROBE (shader-clock timestamps of the middle workgroup, waves 0 and 4, at the phase
boundaries of the sub-chunk pipeline) and runs it on the QM9 B=128 graph.  Run on the GPU box: python tools/agg_probe.py"""
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
so = '/tmp/libpamnet_aggprobe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       '-DPAMNET_PHASE_PROBE'] + os.environ.get('PAMNET_PROBE_FLAGS', '').split() + [
                       '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                       os.path.join(CSRC, 'edge_agg.hip'), '-o', so])
lib.load()
plib = ctypes.CDLL(so)
dev = torch.device('cuda:0')
D = 128
if len(sys.argv) > 1 and sys.argv[1] == 'pdbbind':        # the shape where the kernels walk many chunks (probe: last chunk)
    b = synth.collate([synth.pdbbind_complex(1, i) for i in range(32)]).to(dev)
    g = G.build_graph('PDBbind', 2.0, 6.0, 'source_to_target', b.x, b.batch, num_graphs=32)
else:
    b = synth.qm9_batch(0, 0, 128).to(dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=128)
n, eg = g.n, g.glob.m
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
Wm, bm, Wea = rnd(D, 3 * D) / 8, rnd(D), rnd(D, D) / 8
e, Pi, Pj, x1 = rnd(eg, D), rnd(n, D), rnd(n, D), rnd(n, D)
z, ea, out = torch.empty(eg, D, device=dev), torch.empty(eg, D, device=dev), torch.empty(n, D, device=dev)
csr = g.glob
P = ctypes.c_void_p
I = ctypes.c_int64
fn = plib.pamnet_global_edge_agg_fwd_f32
fn.argtypes = [P, I, I, P, I, P, P, I, P, P, P, P, P, P, P, P, P, P, P]
st = torch.cuda.current_stream().cuda_stream


cuts = torch.empty(257, dtype=torch.int32, device=dev)
lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts), None, st)
use_cuts = True
# PAMNET_PROBE_IMAGES=1: the weights as ready-made fragment images (pamnet_pack_weights_mixed_f32 kind 1; row stride 0)
IMAGES = os.environ.get('PAMNET_PROBE_IMAGES', '0') != '0'
IMGF = 3 * D * D // 2
imgs = torch.empty(4 * IMGF, device=dev)
for tr in (0, 1):
    lib.call('pamnet_pack_weights_mixed_f32', 2, (P * 2)(Wm.data_ptr() + 8 * D, Wea.data_ptr()), (I * 2)(3 * D, D),
             (ctypes.c_int32 * 2)(1, 1), (I * 2)(2 * tr * IMGF, (2 * tr + 1) * IMGF), tr, lib.ptr(imgs), st)
wf = (imgs.data_ptr(), 0, imgs.data_ptr() + 4 * IMGF, 0) if IMAGES else (Wm.data_ptr() + 8 * D, 3 * D, Wea.data_ptr(), D)
wb = (imgs.data_ptr() + 8 * IMGF, 0, imgs.data_ptr() + 12 * IMGF, 0) if IMAGES else (Wm.data_ptr() + 8 * D, 3 * D, Wea.data_ptr(), D)
print('weights as %s' % ('fragment images' if IMAGES else 'fp32 matrices'))


def call(save):
    return fn(e.data_ptr(), eg, n, wf[0], wf[1], bm.data_ptr(), wf[2], wf[3], Pi.data_ptr(), Pj.data_ptr(),
              csr.ptr.data_ptr(), csr.row_of.data_ptr(), csr.col.data_ptr(), cuts.data_ptr() if use_cuts else None, x1.data_ptr(), z.data_ptr() if save else None,
              ea.data_ptr() if save else None, out.data_ptr(), st)


for save in (True, False):
    for _ in range(200):
        call(save)
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    assert call(save) == 0
    t.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    nwg = min(256, (eg + 15) // 16)
    wg = (ctypes.c_longlong * (2 * nwg))()
    plib.pamnet_agg_probe_read(buf, wg, nwg)
    print('save=%s kernel %.1f us (event)' % (save, s.elapsed_time(t) * 1e3))
    for w in (0, 1):
        v = [buf[32 * w + i] for i in range(32)]
        t0 = v[0]
        top = v[30] if v[30] else v[21]                   # (the probed chunk's loop top; PAMNET_PROBE_FLAGS=-DPAMNET_PROBE_CHUNK=k)
        print('  wave %d: weights issued %d | cuts+ptr %d | plan+myp %d | e rows -> LDS %d | idx %d | barrier %d' % (
            4 * w, v[20] - t0, v[21] - v[20], v[22] - top, v[23] - v[22], v[1] - v[23], v[2] - v[1]))
        for sb in range(4):
            a, m, c, d = v[3 + 4 * sb], v[4 + 4 * sb], v[5 + 4 * sb], v[6 + 4 * sb]
            if a == 0 or sb > 3:
                continue
            print('    stage %d: GEMMs %6d  barrier + acc -> LDS %6d  epilogue %6d   (start at %d)' % (sb, m - a, c - m, d - c, a - top))
        print('    barrier %d | reduce + barrier %d ; chunk %d cycles ; kernel so far %d cycles' % (v[28] - max(v[6 + 4 * k] for k in range(4)), v[29] - v[28], v[29] - top, v[29] - t0))
    st0 = min(wg[2 * i] for i in range(nwg))
    life = sorted((wg[2 * i + 1] - wg[2 * i]) / 100.0 for i in range(nwg))
    starts = sorted((wg[2 * i] - st0) / 100.0 for i in range(nwg))
    ends = sorted((wg[2 * i + 1] - st0) / 100.0 for i in range(nwg))
    print('  %d workgroups: start p50 %.1f max %.1f us | end p50 %.1f max %.1f | lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (
        nwg, starts[nwg // 2], starts[-1], ends[nwg // 2], ends[-1], life[nwg // 10], life[nwg // 2], life[9 * nwg // 10], life[-1]))


# ---- backward kernel
d_agg = rnd(n, D)
dz, dea, d_e, dPi = torch.empty(eg, D, device=dev), torch.empty(eg, D, device=dev), torch.zeros(eg, D, device=dev), torch.empty(n, D, device=dev)
fb = plib.pamnet_global_edge_agg_bwd_f32
fb.argtypes = [P, I, I, P, P, P, P, P, P, I, P, I, P, P, P, ctypes.c_int32, P, P]
call(True)
bw = lambda: fb(d_agg.data_ptr(), eg, n, csr.ptr.data_ptr(), csr.row_of.data_ptr(), cuts.data_ptr(), z.data_ptr(), ea.data_ptr(), wb[0],
                wb[1], wb[2], wb[3], dz.data_ptr(), dea.data_ptr(), d_e.data_ptr(), 1, dPi.data_ptr(), st)
for _ in range(200):
    bw()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
assert bw() == 0
t.record()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
wg = (ctypes.c_longlong * (2 * nwg))()
plib.pamnet_agg_probe_read(buf, wg, nwg)
v = [buf[i] for i in range(32)]
print('backward kernel %.1f us (event)' % (s.elapsed_time(t) * 1e3))
names = ['weights + cuts + plan', 'load sweep (gather d_agg, z, ea; dz / dea stores; -> LDS)', 'barrier', 'reduce d P_i', '2 GEMMs',
         'acc -> LDS + barriers', 'd_e sweep (accumulate load + store)']
for i, nm in enumerate(names):
    print('   %-62s %7d cycles' % (nm, v[i + 1] - v[i]))
print('   total %d cycles, of which the prologue (weights, work split, plan, first rows requested) %d' % (v[7] - v[0], v[13] - v[0]))
life = sorted((wg[2 * i + 1] - wg[2 * i]) / 100.0 for i in range(nwg))
print('   lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (life[nwg // 10], life[nwg // 2], life[9 * nwg // 10], life[-1]))

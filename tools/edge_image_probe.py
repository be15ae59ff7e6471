#!/usr/bin/env python
"""Edge-level kernels on ready-made weight images (pamnet_pack_weights_mixed_f32) against the same launches on fp32 matrices, at
the QM9 (B = 128) or PDBbind (B = 32) row counts: event-timed, back to back.  Run on the GPU box: python tools/edge_image_probe.py [qm9|pdbbind]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import lib  # noqa: E402

dev = torch.device('cuda:0')
D = 128
shape = sys.argv[1] if len(sys.argv) > 1 else 'qm9'
rows, edges = (17640, 4316) if shape == 'qm9' else (126928, 36656)
rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
st = torch.cuda.current_stream().cuda_stream


def event_us(fn, reps=50, groups=5):
    fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        t.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(t)[len(t) // 2]


dy, z1, z2 = rnd(rows, D), rnd(rows, D), rnd(rows, D)
W1, W2 = rnd(D, D) / 8, rnd(D, D) / 8
d_mji, d_mnb, d_q3, z_ji, z_kj, q2 = (rnd(edges, D) for _ in range(6))
Wq = [rnd(D, 3 * D) / 8, rnd(D, 3 * D) / 8, rnd(D, D) / 8, rnd(D, D) / 8]
wqp = [Wq[0].data_ptr() + 8 * D, Wq[1].data_ptr() + 8 * D, Wq[2].data_ptr(), Wq[3].data_ptr()]
o_r = [torch.empty(rows, D, device=dev) for _ in range(3)]
o_e = [torch.empty(edges, D, device=dev) for _ in range(4)]
IMG1, IMG0 = 3 * D * D // 2, D * D
images = torch.empty(6 * IMG1, device=dev)
srcs = (ctypes.c_void_p * 6)(W1.data_ptr(), W2.data_ptr(), *wqp)
lds = (ctypes.c_int64 * 6)(D, D, 3 * D, 3 * D, D, D)
kinds = (ctypes.c_int32 * 6)(1, 1, 1, 1, 1, 1)
offs = (ctypes.c_int64 * 6)(*[i * IMG1 for i in range(6)])
lib.call('pamnet_pack_weights_mixed_f32', 6, srcs, lds, kinds, offs, 1, lib.ptr(images), st)
ip = lambda i: images.data_ptr() + 4 * offs[i]
P4 = ctypes.c_void_p * 4
I4 = ctypes.c_int64 * 4


def pair(w1, w2, flag, wq, ldq):
    lib.call('pamnet_local_bwd_pair_f32', lib.ptr(dy), rows, lib.ptr(z1), lib.ptr(z2), w1, w2, lib.ptr(o_r[0]), lib.ptr(o_r[1]),
             lib.ptr(o_r[2]), flag, lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji), lib.ptr(z_kj),
             lib.ptr(q2), wq, ldq, lib.ptr(o_e[0]), lib.ptr(o_e[1]), lib.ptr(o_e[2]), lib.ptr(o_e[3]), 0, st)


plain_wq, plain_ld = P4(*wqp), I4(3 * D, 3 * D, D, D)
img_wq, img_ld = P4(ip(2), ip(3), ip(4), ip(5)), I4(0, 0, 0, 0)
print('%s: %d triplet / pair rows, %d local edges' % (shape, rows, edges))
for rep in range(2):
    print('  local_bwd_pair: matrices %.1f us | all images %.1f us | MLP images only %.1f us | local images only %.1f us' % (
        event_us(lambda: pair(W1.data_ptr(), W2.data_ptr(), 0, plain_wq, plain_ld)),
        event_us(lambda: pair(ip(0), ip(1), 2, img_wq, img_ld)),
        event_us(lambda: pair(ip(0), ip(1), 2, plain_wq, plain_ld)),
        event_us(lambda: pair(W1.data_ptr(), W2.data_ptr(), 0, img_wq, img_ld))))


# ---- the local edge forward (QM9: one chunk of 2-3 tiles per workgroup)
n_nodes = max(edges // 2, 8)
rbf, Pl = rnd(edges, D), [rnd(n_nodes, D) for _ in range(4)]
bji, bkj = rnd(D), rnd(D)
row_of = torch.sort(torch.randint(0, n_nodes, (edges,), device=dev))[0].to(torch.int32)
col = torch.randint(0, n_nodes, (edges,), device=dev).to(torch.int32)
fimg = torch.empty(4 * IMG1, device=dev)
lib.call('pamnet_pack_weights_mixed_f32', 4, P4(*wqp), I4(3 * D, 3 * D, D, D), (ctypes.c_int32 * 4)(1, 1, 1, 1),
         I4(0, IMG1, 2 * IMG1, 3 * IMG1), 0, lib.ptr(fimg), st)
fo = [torch.empty(edges, D, device=dev) for _ in range(6)]


def lfwd(wq, ldq):
    lib.call('pamnet_local_edge_fwd_f32', lib.ptr(rbf), edges, wq, ldq, lib.ptr(bji), lib.ptr(bkj),
             P4(*[p.data_ptr() for p in Pl]), lib.ptr(row_of), lib.ptr(col), *[lib.ptr(o) for o in fo], st)


fwq = P4(*[fimg.data_ptr() + 4 * i * IMG1 for i in range(4)])
for rep in range(2):
    print('  local_edge_fwd: matrices %.1f us | images %.1f us' % (event_us(lambda: lfwd(plain_wq, plain_ld)),
                                                                  event_us(lambda: lfwd(fwq, img_ld))))

# ---- the two halves of the backward pair alone (each on its own plan over all CUs)
def mlp_alone():
    lib.call('pamnet_mlp2_bwd_f32', lib.ptr(dy), rows, lib.ptr(z1), lib.ptr(z2), lib.ptr(W1), lib.ptr(W2), lib.ptr(o_r[0]),
             lib.ptr(o_r[1]), lib.ptr(o_r[2]), 0, st)


def edge_alone(wq, ldq):
    lib.call('pamnet_local_edge_bwd_f32', lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji), lib.ptr(z_kj),
             lib.ptr(q2), wq, ldq, lib.ptr(o_e[0]), lib.ptr(o_e[1]), lib.ptr(o_e[2]), lib.ptr(o_e[3]), 0, st)


print('  alone: mlp2_bwd (fp32 matrices) %.1f us | local_edge_bwd on images %.1f us' % (
    event_us(mlp_alone), event_us(lambda: edge_alone(img_wq, img_ld))))

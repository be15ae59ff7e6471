#!/usr/bin/env python
"""Where the input-embedding launches (csrc/embed.hip: embed_multi_fwd / embed_multi_bwd + reduce) spend their time: every
layer of the stage alone and together, forward and backward, at the PDBbind (B = 32) and QM9 (B = 128) row counts, on private
builds with parts of the backward tile compiled out (-DEMBED_PROBE_NO_Z / _NO_DW / _NO_DX: wrong results, timing only).
Run on the GPU box: python tools/embed_probe.py [pdbbind|qm9] ['<extra hipcc flags>' ...]"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch  # noqa: E402

from pamnet_amd import lib  # noqa: E402

CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
dev = torch.device('cuda:0')
D = 128
P = ctypes.c_void_p


def build(tag, flags):
    so = '/tmp/libpamnet_embedprobe_%s.so' % tag
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] + flags +
                          ['-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on',
                           os.path.join(CSRC, 'embed.hip'), '-o', so])
    return ctypes.CDLL(so)


def event_us(fn, reps=20, groups=5):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(groups):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        t.record()
        t.synchronize()
        best.append(s.elapsed_time(t) * 1e3 / reps)
    best.sort()
    return best[len(best) // 2]


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ('pdbbind', 'qm9') else 'pdbbind'
    if shape == 'pdbbind':
        eg, el, tp, n, k_x = 709656, 36656, 126928, 19088, 18
    else:
        eg, el, tp, n, k_x = 32888, 4316, 17640, 2286, 0
    rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
    st = torch.cuda.current_stream().cuda_stream
    freq = torch.arange(1, 17, device=dev, dtype=torch.float32) * 3.14159265
    need = ctypes.c_int64(0)
    layers = []                                                  # (name, rows, K, dist?, kind?)
    layers.append(('rbf_g  K=16 Bessel rows, %d rows' % eg, eg, 16, True, False))
    layers.append(('rbf_l  K=16 Bessel rows, %d rows' % el, el, 16, True, False))
    layers.append(('sbf    K=42 two sets,    %d rows' % tp, tp, 42, False, True))
    if k_x:
        layers.append(('x init K=18 no act,      %d rows' % n, n, 18, False, False))
    keep = []

    def job(jb, rows, K, dist, kind):
        x = None if dist else rnd(rows, K)
        d = torch.rand(rows, device=dev) * 4.0 + 0.5 if dist else None
        kd = (torch.arange(rows, device=dev) % 3 == 0).to(torch.int32) if kind else None
        W0, b0 = rnd(D, K) / 4, (rnd(D) if K != 18 else None)
        W1, b1 = (rnd(D, K) / 4, rnd(D)) if kind else (None, None)
        out, gout = torch.empty(rows, D, device=dev), rnd(rows, D)
        lib.call('pamnet_embed_scratch_floats', rows, K, ctypes.addressof(need))
        partial = torch.empty(int(need.value), device=dev)
        gW0, gb0, gW1, gb1, gf = torch.empty_like(W0), torch.empty(D, device=dev), torch.empty_like(W0), torch.empty(D, device=dev), \
            torch.empty(16, device=dev)
        keep.extend([x, d, kd, W0, b0, W1, b1, out, gout, partial, gW0, gb0, gW1, gb1, gf])
        jb.x, jb.dist, jb.freq = lib.ptr(x), lib.ptr(d), lib.ptr(freq if dist else None)
        jb.cutoff, jb.K, jb.act, jb.rows = 5.0 if dist else 0.0, K, 0 if K == 18 else 1, rows
        jb.kind, jb.W0, jb.b0, jb.W1, jb.b1 = lib.ptr(kd), lib.ptr(W0), lib.ptr(b0), lib.ptr(W1), lib.ptr(b1)
        jb.out, jb.gout, jb.partial = lib.ptr(out), lib.ptr(gout), lib.ptr(partial)
        jb.dW0, jb.db0 = lib.ptr(gW0), lib.ptr(gb0 if b0 is not None else None)
        jb.dW1, jb.db1 = lib.ptr(gW1 if kind else None), lib.ptr(gb1 if kind else None)
        jb.dfreq = lib.ptr(gf if dist else None)
        return 4.0 * D * rows + (4.0 * rows if dist else 4.0 * K * rows)

    lib.load()
    variants = [('production', [])] + [(f.replace('-DEMBED_PROBE_', '').lower(), f.split()) for f in sys.argv[2:]]
    if len(sys.argv) <= 2:
        variants += [('no_z', ['-DEMBED_PROBE_NO_Z']), ('no_dw', ['-DEMBED_PROBE_NO_DW']), ('no_dx', ['-DEMBED_PROBE_NO_DX']),
                     ('no_z_dw_dx', ['-DEMBED_PROBE_NO_Z', '-DEMBED_PROBE_NO_DW', '-DEMBED_PROBE_NO_DX'])]
    nl = len(layers)
    all_jobs = (lib.EmbedJob * nl)()
    by_all = sum(job(all_jobs[j], *layers[j][1:]) for j in range(nl))
    single = []
    for j in range(nl):
        one = (lib.EmbedJob * 1)()
        single.append((one, job(one[0], *layers[j][1:])))
    print('%s row counts: E_g %d  E_l %d  T+P %d  N %d' % (shape, eg, el, tp, n))
    for tag, flags in variants:
        pl = build(tag, flags)
        for f in (pl.pamnet_embed_multi_fwd_f32, pl.pamnet_embed_multi_bwd_f32):
            f.argtypes = [P, ctypes.c_int32, P, P]
        print('== %s' % tag)
        rows = [('all layers, one launch', all_jobs, nl, by_all)] + [(layers[j][0], single[j][0], 1, single[j][1]) for j in range(nl)]
        for name, jobs, cnt, by in rows:
            fw = lambda: pl.pamnet_embed_multi_fwd_f32(ctypes.addressof(jobs), cnt, None, st)
            bw = lambda: pl.pamnet_embed_multi_bwd_f32(ctypes.addressof(jobs), cnt, None, st)
            assert fw() == 0 and bw() == 0
            uf, ub = event_us(fw), event_us(bw)
            print('  %-40s forward %7.1f us (%5.2f TB/s)   backward + reduce %7.1f us (%5.2f TB/s)' %
                  (name, uf, by / uf / 1e6, ub, by / ub / 1e6))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""bench.py -- molecules/sec of one PAMNet training step (QM9 schema, dim=128, n_layer=6) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 400 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch with inputs already resident in HBM: zero_grad -> forward
(device graph construction, bases, 6x(global+local) message passing, fusion, pooling) -> L1 loss -> backward -> RCCL
all-reduce of the flat gradient (N>1) -> clip -> Adam -> EMA, i.e. the reference loop main_qm9.py:103-118.
Workload (config.workload): BASELINE.json configs[1] -- 128 molecules per GPU (weak scaling: global batch 128*N).
One JSON line on rank 0: whole-job molecules/s + `roofline` (the scatter-add kernel, HIP-event timed, PMC traffic from
profiles/) + `step_kernels` (rooflines of the step's dominant kernels at the workload shapes) + `cpu_baseline` (the
oracle = port of the reference CPU forward+backward, bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA (MI355X_MICROARCH.md), the pipe the bf16x6 kernels issue on
METRIC = 'molecules/sec (QM9 dim=128 n_layer=6) at 1/2/4/8 GPU; scatter-add HBM GB/s vs peak'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400, help='timed steps (default: a timed region of >= 1 s)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch-per-gpu', type=int, default=128)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--n-layer', type=int, default=6)
    ap.add_argument('--n-batches', type=int, default=4, help='distinct resident batches cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-rooflines', action='store_true', help='skip the kernel roofline probes (profiling runs)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the PDBbind / RNA side measurements')
    ap.add_argument('--force-comm', action='store_true',
                    help='N=1 only: run the bucketed gradient all-reduce through a 1-rank RCCL group (overhead check)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--stream-gb', type=float, default=2.0, help='size of the streamed scatter-add roofline probe')
    ap.add_argument('--share-gpu', action='store_true',
                    help='NOT a measurement: all ranks on cuda:0 with gloo as the transport -- the whole N > 1 flow of the real '
                         'model (shards, bucketed overlapped all-reduce, exposed-communication probe) on a 1-GPU box')
    ap.add_argument('--n1-ms', type=float, default=None,
                    help='ms/step of the N = 1 run of the same command: the line then carries comm.efficiency_vs_n1 (weak scaling)')
    ap.add_argument('--init-timeout', type=float, default=60.0, help='seconds the process-group rendezvous may take')
    ap.add_argument('--cpu-dry-run', action='store_true',
                    help='NOT a measurement: exercise the multi-rank control flow (sharding, barriers, max-over-ranks '
                         'timing, gradient all-reduce) on CPU with the gloo backend and a stand-in module (tests/standin.py)')
    return ap.parse_args()


def event_time_ms(fn, reps, groups=1):
    """Duration of fn() in ms with HIP events on torch's current stream (the stream the kernels launch on): `groups`
    runs of `reps` back-to-back calls; returns (median, min) of the per-call averages."""
    vals = []
    for _ in range(groups):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        vals.append(s.elapsed_time(e) / reps)
    vals.sort()
    return vals[len(vals) // 2], vals[0]


def box_calibration(dev, gb=1.0):
    """What THIS box's memory system delivers to the simplest kernels, measured in the same process as the roofline probe (boxes
    of the pool differ by +-5 % at 2 GB streams: a fraction of the 8 TB/s peak alone mixes the kernel with the box): a 16-byte-
    per-lane copy (torch's vectorised elementwise copy: bytes read + written) -- the library's own stream copy, with torch's elementwise copy beside it."""
    from pamnet_amd import lib
    n = int(gb * 1e9 / 16) * 4
    a, b = torch.randn(n, device=dev), torch.empty(n, device=dev)
    st = lib.stream_of(a)
    fn = lambda: lib.call('pamnet_stream_copy_f32', lib.ptr(a), lib.ptr(b), n, st)
    for _ in range(3):
        fn()
    ms_c, ms_best = event_time_ms(fn, 10, 5)
    for _ in range(3):
        b.copy_(a)
    ms_t, _ = event_time_ms(lambda: b.copy_(a), 10, 5)
    return {'copy_gbs': 8.0 * n / ms_c / 1e6, 'copy_gbs_best_group': 8.0 * n / ms_best / 1e6, 'torch_copy_gbs': 8.0 * n / ms_t / 1e6,
            'bytes': 4 * n,
            'note': 'pamnet_stream_copy_f32 (16 bytes per lane, non-temporal, 8 loads in flight; read + write bytes) and '
                    "torch's elementwise copy over %.1f GB, HIP-event timed, median of 5 groups of 10" % gb}


def committed_kernel_avg(stats_file, kernel_substr):
    """(average us, algorithmic GB) of a kernel in a committed rocprofv3 --kernel-trace --stats summary under profiles/ (written
    by tools/pmc_scatter.sh: a header line with the probe's shape and bytes, then per kernel its name and `calls .. average ..
    us`), or None."""
    import re
    path = os.path.join(REPO, 'profiles', stats_file)
    if not os.path.exists(path):
        return None
    text = open(path).read()
    gb = re.search(r'([0-9.]+) GB algorithmic', text)
    m = re.search(re.escape(kernel_substr) + r'[^\n]*\n\s*calls\s+\d+\s+average\s+([0-9.]+)\s+us', text)
    if not (gb and m):
        return None
    return float(m.group(1)), float(gb.group(1))


def scatter_add_roofline(dev, g, d, stream_gb):
    """Roofline of the scatter-add kernel (pamnet_segment_sum_f32).
    (1) at the workload's own global-aggregation shape [E_g, d] -> [N, d]  (fits the 256 MB Infinity Cache);
    (2) on a >= stream_gb streamed input with the same segment-length distribution (honest HBM number).
    Algorithmic bytes per launch = 4*d*M (src) + 4*(R+1) (CSR ptr) + 4*d*R (out)  (SURVEY.md 8d)."""
    from pamnet_amd import ops
    res = {}
    csr = g.glob
    m, r = csr.m, csr.rows
    src = torch.randn(m, d, device=dev)
    out = torch.empty(r, d, device=dev)
    fn = lambda: ops.segment_sum_raw(out, None, src, None, None, None, None, csr.ptr, r, d)
    fn()
    ms, _ = event_time_ms(fn, 50, 3)
    by = 4.0 * d * m + 4.0 * (r + 1) + 4.0 * d * r
    res['workload'] = dict(rows_in=m, rows_out=r, bytes=by, ms=ms, gbs=by / ms / 1e6)
    # streamed probe: replicate the CSR until the source exceeds stream_gb
    rep = max(1, int(stream_gb * 1e9 / (4.0 * d * m)) + 1)
    lens = (csr.ptr[1:] - csr.ptr[:-1]).repeat(rep)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), lens.long().cumsum(0)]).to(torch.int32)
    M, R = m * rep, r * rep
    src = torch.randn(M, d, device=dev)
    out = torch.empty(R, d, device=dev)
    fn = lambda: ops.segment_sum_raw(out, None, src, None, None, None, None, ptr, R, d)
    for _ in range(10):                     # first touches of 2 GB and the clock ramp stay out of the timed groups
        fn()
    torch.cuda.synchronize()
    # median of 7 groups of 20 launches: single-group timings differed by 0.65-0.73 of peak between fresh boxes in round 1
    ms, ms_min = event_time_ms(fn, 20, 7)
    by = 4.0 * d * M + 4.0 * (R + 1) + 4.0 * d * R
    res['streamed'] = dict(rows_in=M, rows_out=R, bytes=by, ms=ms, ms_best=ms_min, gbs=by / ms / 1e6,
                           gbs_best=by / ms_min / 1e6)
    return res


def step_kernel_rooflines(dev, g, d, n_layer):
    """Rooflines of the kernels that dominate the timed step, each timed with HIP events at THIS workload's shapes through
    the same C-ABI entry points the engine calls (profiles/ holds the rocprofv3 kernel trace of the bench command with the
    in-step averages of the same kernels):
      * the weight-gradient launch (dominant kernel of the step): all dW = dZ^T A of one global layer -- MFMA bound;
      * edge MLP -> node segment-sum (the fused kernel north_star names): 2 GEMMs per global edge -- MFMA bound, its
        algorithmic HBM bytes beside it;
      * the local layer's two chained scatter-adds as one launch -- gather / latency bound (bytes)."""
    from pamnet_amd import fused, lib
    n, eg, el, tp = g.n, g.glob.m, g.loc.m, g.tp.m
    rnd = lambda *s: torch.randn(*s, device=dev) * 0.5
    out = []
    # engine.hip, backward of a layer pair with the chains' ten tail jobs riding in chain launches: the pair's own jobs are ONE
    # launch -- the local layer's 11 (mlp_x1, the four node-side projection blocks, the four edge-side blocks, the two
    # triplet/pair MLP layers) + the global layer's 5 (mlp_x1, two node-side blocks, W_e, W_edge_attr) -- which also reduces
    # the previous batch's partials: `wgrad_fused_kernel` in the step's trace
    keep, jobs = [], []
    for rows, cnt in ((n, 5), (el, 4), (tp, 2), (n, 3), (eg, 2)):
        for _ in range(cnt):
            dz, a, dw = rnd(rows, d), rnd(rows, d), torch.empty(d, d, device=dev)
            keep += [dz, a, dw]
            jobs.append((dz, d, a, d, 0, rows, dw, d, None))
    dw_ = fused.DeferredWgrad(keep[0])
    fn = lambda: dw_.launch(jobs)
    fn()
    ms, _ = event_time_ms(fn, 30, 3)
    dw_.flush()
    fl = 2.0 * d * d * (8 * n + 4 * el + 2 * tp + 2 * eg)
    # The kernel computes fp32-accurate products on the bf16 matrix pipe (three exact bf16 pieces per operand, six bf16
    # MFMAs per 32 rows: csrc/gemm_core.h "bf16x6"): `frac` stays against the fp32-MFMA peak the path is priced on
    # (SURVEY 8d); `frac_bf16x6` prices the same algorithmic FLOPs against the ceiling of the instruction stream it
    # actually issues, dense bf16 peak / 6.
    out.append({'kernel': 'wgrad_fused_kernel via pamnet_wgrad_deferred_f32 (a layer pair\'s own launch in the step: 8 node-level + '
                          '4 local-edge + 2 triplet/pair + 2 global-edge dW, plus the fixed-order reduction of the previous batch)',
                'bound': 'mfma',
                'flops_per_launch': fl, 'us_per_launch': ms * 1e3, 'achieved': fl / ms / 1e9, 'peak': FP32_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': fl / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS,
                'arithmetic': 'fp32-accurate on v_mfma_f32_16x16x32_bf16 (3 exact bf16 pieces per operand, 6 products)',
                'peak_bf16x6': BF16_MFMA_PEAK_TFLOPS / 6.0, 'frac_bf16x6': fl / ms / 1e9 / (BF16_MFMA_PEAK_TFLOPS / 6.0),
                'launches_per_step': n_layer - 1})
    Wm, bm, Wea = rnd(d, 3 * d) / 8, rnd(d), rnd(d, d) / 8
    e, Pi, Pj, x1 = rnd(eg, d), rnd(n, d), rnd(n, d), rnd(n, d)
    z, ea, x2 = torch.empty(eg, d, device=dev), torch.empty(eg, d, device=dev), torch.empty(n, d, device=dev)
    csr = g.glob
    st = lib.stream_of(e)
    cuts_t = torch.empty(257, dtype=torch.int32, device=dev)
    lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, eg, lib.ptr(cuts_t), None, st)
    cuts = lib.ptr(cuts_t)       # the engine computes this table once per forward

    def agg(save):
        lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), eg, n, Wm.data_ptr() + 8 * d, 3 * d, lib.ptr(bm), lib.ptr(Wea),
                 d, lib.ptr(Pi), lib.ptr(Pj), lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(csr.col), cuts, lib.ptr(x1),
                 lib.ptr(z) if save else None, lib.ptr(ea) if save else None, lib.ptr(x2), st)

    for save, tag in ((True, 'training (z, ea saved)'), (False, 'inference')):
        agg(save)
        ms, _ = event_time_ms(lambda: agg(save), 30, 3)
        fl = 2.0 * 2.0 * d * d * eg
        by = 4.0 * d * eg + 8.0 * eg + 4.0 * (n + 1) + 4.0 * d * n * 4 + (8.0 * d * eg if save else 0.0)
        out.append({'kernel': 'global_edge_agg_fwd_kernel (edge MLP -> node segment-sum), ' + tag, 'bound': 'mfma',
                    'flops_per_launch': fl, 'us_per_launch': ms * 1e3, 'achieved': fl / ms / 1e9,
                    'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': fl / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS,
                    'algorithmic_bytes': by, 'gbs': by / ms / 1e6, 'hbm_frac': by / ms / 1e6 / HBM_PEAK_GBS,
                    'launches_per_step': n_layer if save else 0})
    m_ji, m_nb, q3, s_, mt = rnd(el, d), rnd(el, d), rnd(el, d), rnd(tp, d), torch.empty(el, d, device=dev)
    fn = lambda: lib.call('pamnet_local_agg_fwd_f32', lib.ptr(m_ji), lib.ptr(m_nb), lib.ptr(s_), lib.ptr(q3),
                          lib.ptr(g.tp.ptr), lib.ptr(g.tp.col), lib.ptr(g.loc.ptr), lib.ptr(x1), n, lib.ptr(mt),
                          lib.ptr(x2), st)
    fn()
    ms, _ = event_time_ms(fn, 50, 3)
    by = 4.0 * d * (2 * tp + 3 * el + 2 * n) + 4.0 * (tp + el + n)     # s + gathered m_nb rows, m_ji/q3/m_t, x1/x2, indices
    out.append({'kernel': 'local_agg_fwd_kernel (rows -> edges -> nodes scatter-adds, one launch)', 'bound': 'hbm',
                'bytes_per_launch': by, 'us_per_launch': ms * 1e3, 'achieved': by / ms / 1e6, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': by / ms / 1e6 / HBM_PEAK_GBS, 'launches_per_step': n_layer,
                'note': 'every operand fits the 256 MB Infinity Cache at this batch size: latency bound'})
    return out


def pmc_record(name):
    """The committed PMC pass of the roofline kernel (profiles/<name>.json, written on the GPU box by tools/pmc_scatter.sh:
    separate --pmc passes, FETCH_SIZE x2 correction as MI355X_MICROARCH.md prescribes), or None."""
    path = os.path.join(REPO, 'profiles', name + '.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def mfma_summary(args, g, ms_per_step):
    """Algorithmic dense-layer FLOPs of one training step of this rank (forward x3 for fwd+bwd) over the measured step
    time, against the fp32 MFMA peak.  Per layer pair (with the W[x_i|x_j|e] split):
    2 d^2 [ N (1+2+10) + 2 E_g ]  (global)  +  2 d^2 [ N (1+4+10) + 4 E_l + 2 (T+P) ]  (local)."""
    d, n, eg, el, tp = args.dim, g.n, g.glob.m, g.loc.m, g.tp.m
    fwd = args.n_layer * 2.0 * d * d * (13 * n + 2 * eg + 15 * n + 4 * el + 2 * tp)
    fwd += 2.0 * d * (16 * (eg + el) + 42 * tp)                       # input embeddings (once per forward)
    tf = 3.0 * fwd / (ms_per_step * 1e-3) / 1e12
    return {'bound': 'mfma', 'scope': 'whole training step, algorithmic dense FLOPs (fwd x3)', 'flops_per_step': 3.0 * fwd,
            'achieved': tf, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / FP32_MFMA_PEAK_TFLOPS}


def other_configs(dev):
    """Full training step and forward of the other BASELINE.json configurations on this GPU (parity-test cases, not the
    bench line's `value`): configs[3] PDBbind schema d=128 L=3 B=32, configs[4] RNA schema d=16 L=1 B=8 and PAMNet_s on the main
    loop's QM9 batches -- graph rebuilt
    every step, input pipeline as in the main loop, 4 distinct resident batches."""
    import models
    from pamnet_amd import synth
    from pamnet_amd.store import MoleculeStore
    from pamnet_amd.train import Trainer, predict
    out = {}
    rna = [synth.rna_chain(2, i) for i in range(8)]        # the 8 graphs of configs[4]; the 4 batches are rotations of them
    pdb = [synth.pdbbind_complex(1, i) for i in range(128)]
    qm9 = [synth.qm9_molecule(0, i) for i in range(512)]  # the molecules of the main loop, for PAMNet_s (models.py:283-353)
    # each configuration with ITS driver's step: main_pdbbind.py:88-95 (F.mse_loss, Adam, no clip, no EMA),
    # main_rna_puzzles.py:86-93 (F.smooth_l1_loss, Adam, no clip, no EMA), main_qm9.py:103-118 (L1, clip 1000, EMA)
    for tag, cfg, graphs, sel, steps, loop in (
            ('pdbbind_b32_d128_l3', models.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0),
             pdb, lambda k: list(range(32 * k, 32 * k + 32)), 40, dict(loss='mse', max_grad_norm=None, ema_decay=None, lr=1e-3)),
            ('rna_b8_d16_l1', models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0,
                                            flow='target_to_source'),
             rna, lambda k: [(i + 2 * k) % 8 for i in range(8)], 100,
             dict(loss='smooth_l1', max_grad_norm=None, ema_decay=None, lr=1e-4)),
            ('pamnet_s_qm9_b128_d128_l6', models.Config(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0),
             qm9, lambda k: list(range(128 * k, 128 * k + 128)), 100, dict(loss='l1', lr=1e-4))):
        torch.manual_seed(7)
        model = (models.PAMNet_s if tag.startswith('pamnet_s') else models.PAMNet)(cfg).to(dev)
        tr = Trainer(model, **loop)
        bs = [synth.collate([graphs[i] for i in sel(k)]).to(dev) for k in range(4)]
        for i in range(20):                                   # (the allocator's cache was emptied after the previous configuration)
            tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(bs[i % 4], next_data=bs[(i + 1) % 4])
        torch.cuda.synchronize()
        step_ms = (time.perf_counter() - t0) / steps * 1e3
        with torch.no_grad():
            for _ in predict(model, (bs[i % 4] for i in range(4))):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in predict(model, (bs[i % 4] for i in range(steps))):
                pass
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t0) / steps * 1e3
            model(bs[0])
        g = model._graph_cache
        # the same molecules through the resident store (pamnet_amd/store.py): one collate launch per batch inside the timed
        # loop, sizes from the per-graph table -> graph construction + basis is ONE engine call, nothing is read back
        st = MoleculeStore(graphs, dev).prepare_for(model)
        nxt = st.collate(sel(0))
        for i in range(12):                                   # three passes over the four batch selections: the store's batches
            cur, nxt = nxt, st.collate(sel((i + 1) % 4))      # differ in size from `bs`, and an allocator segment created inside
            tr.step(cur, next_data=nxt)                       # the timed loop is a ~100 ms stall (seen as 11-12 ms/step on PDBbind)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            cur, nxt = nxt, st.collate(sel((i + 1) % 4))
            tr.step(cur, next_data=nxt)
        torch.cuda.synchronize()
        store_step_ms = (time.perf_counter() - t0) / steps * 1e3
        tr.drain()
        with torch.no_grad():
            for i in range(4):
                model(st.collate(sel(i % 4)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                model(st.collate(sel(i % 4)))
            torch.cuda.synchronize()
            store_fwd_ms = (time.perf_counter() - t0) / steps * 1e3
        model.verify()
        out[tag] = {'train_ms_per_step': step_ms, 'forward_ms': fwd_ms, 'store_train_ms_per_step': store_step_ms,
                    'store_forward_ms_unpipelined': store_fwd_ms, 'graphs_per_batch': int(bs[0].num_graphs),
                    'nodes': int(g.n), 'global_edges': int(g.glob.m), 'local_edges': int(g.loc.m),
                    'triplet_pair_rows': int(g.tp.m), 'steps': steps,
                    'step': 'loss=%s, clip=%s, ema=%s' % (loop['loss'], loop.get('max_grad_norm', 1000.0), loop.get('ema_decay', 0.999)),
                    'note': 'train / forward: plain tensors (the reference calling convention) with the side-stream input '
                            'pipeline; store_*: resident dataset, device-side collation, graph + basis as one engine call'}
        if tag.startswith('pdbbind'):
            out[tag]['kernels'] = guarded(pdbbind_kernel_rooflines, model, bs[0], dev)
        del tr, model, bs, st
        torch.cuda.empty_cache()
    return out


def pdbbind_kernel_rooflines(model, batch, dev):
    """The gather (transposed-CSR, `perm`) form of the scatter-add at the shape where it is real HBM traffic: the source-side
    reduction d P_j[j] = sum over edges LEAVING j of d z[e] in the global layer's backward (csrc/engine.hip:
    pamnet_segment_sum_f32 with gT_perm / gT_ptr), PDBbind B=32: ~690 k rows of 512 B gathered through an index list.
    Algorithmic bytes = 4 d M (rows) + 4 M (perm) + 4 (R + 1) (ptr) + 4 d R (out).  Beside it the streamed form on the same
    rows (no perm) -- the variant `roofline` reports."""
    from pamnet_amd import ops
    model.prepare(batch, need_grad=True)
    g = batch._pamnet_prepared
    batch._pamnet_prepared = None
    d, n, m = model.dim, g.n, g.glob.m
    src, dst = torch.randn(m, d, device=dev), torch.empty(n, d, device=dev)
    res = []
    for name, perm, ptr in (('segment_sum_split_kernel<..., perm> (transposed CSR: d z rows gathered by source node)', g.glob_T.perm, g.glob_T.ptr),
                            ('segment_sum_kernel (streamed rows, same shape)', None, g.glob.ptr)):
        fn = lambda: ops.segment_sum_raw(dst, None, src, None, None, None, perm, ptr, n, d)
        for _ in range(5):
            fn()
        ms, ms_min = event_time_ms(fn, 20, 5)
        # ... and as the step meets it: right behind a kernel that streamed ~2 GB (global_edge_agg_bwd_wg), i.e. with NOTHING of
        # its 354 MB input in the 256 MB Infinity Cache.  Back-to-back repetitions of the probe above find up to ~70 % of the
        # rows still cached from the previous repetition -- the round-5 verdict's "65.7 us alone, 87.9 in the step".
        spill = torch.empty(int(1.5e9 / 4), device=dev)
        cold = []
        for _ in range(7):
            spill.add_(1.0)                                  # 3 GB through the caches
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            fn()
            e_ev.record()
            e_ev.synchronize()
            cold.append(s_ev.elapsed_time(e_ev))
        del spill
        cold.sort()
        ms_cold = cold[len(cold) // 2]
        by = 4.0 * d * m + (4.0 * m if perm is not None else 0.0) + 4.0 * (n + 1) + 4.0 * d * n
        res.append({'kernel': name, 'bound': 'hbm', 'rows_in': int(m), 'rows_out': int(n), 'bytes_per_launch': by,
                    'us_per_launch': ms * 1e3, 'achieved': by / ms / 1e6, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': by / ms / 1e6 / HBM_PEAK_GBS, 'best_group_gbs': by / ms_min / 1e6,
                    'us_per_launch_cold_caches': ms_cold * 1e3, 'frac_cold_caches': by / ms_cold / 1e6 / HBM_PEAK_GBS,
                    'note': 'us_per_launch: back-to-back repetitions (part of the input still in the Infinity Cache); '
                            '*_cold_caches: each launch behind 3 GB of unrelated traffic -- the condition inside the step',
                    'launches_per_step': model.n_layer if perm is not None else 0})
    # The input stage at this shape (verdict r5 item 7: embed_multi_fwd / bwd, mlp2_fwd -- once per step each): all input
    # embeddings as one launch forward (Bessel rows formed in the kernel), two backward; the triplet / pair MLPs of all layers as
    # one launch.  Algorithmic bytes: the [rows, 128] outputs written (forward) / their gradients read (backward) + the raw inputs.
    try:
        import ctypes as _ct
        from pamnet_amd import lib as _lib
        el, tp = int(g.loc.m), int(g.tp.m)
        rows_all = el + m + tp + n
        raw = 4.0 * (el + m) + 4.0 * 42 * tp + 4.0 * 18 * n + 4.0 * tp
        staged = None

        def stage_fwd():
            nonlocal staged
            staged = model._input_stage(batch, g, None, model.mlp_sbf2[0][0], model.mlp_sbf1[0][0])

        with torch.enable_grad():
            stage_fwd()
            ms_f, _ = event_time_ms(stage_fwd, 10, 3)
            gs = [torch.randn_like(t) for t in staged]

            def stage_bwd():
                stage_fwd()
                torch.autograd.backward(list(staged), gs)

            stage_bwd()
            ms_fb, _ = event_time_ms(stage_bwd, 10, 3)
        for p_ in model.parameters():
            p_.grad = None
        ms_b = max(ms_fb - ms_f, 1e-6)
        by_sf, by_sb = 4.0 * d * rows_all + raw, 4.0 * d * rows_all + raw
        res.append({'kernel': 'embed_multi_fwd_kernel (every input embedding of the batch, one launch)', 'bound': 'hbm',
                    'rows': rows_all, 'bytes_per_launch': by_sf, 'us_per_launch': ms_f * 1e3, 'achieved': by_sf / ms_f / 1e6,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': by_sf / ms_f / 1e6 / HBM_PEAK_GBS, 'launches_per_step': 1})
        res.append({'kernel': 'embed_multi_bwd_kernel + embed_multi_reduce_kernel (their backward; forward + backward timed, forward '
                              'subtracted)', 'bound': 'hbm', 'rows': rows_all, 'bytes_per_launch': by_sb,
                    'us_per_launch': ms_b * 1e3, 'achieved': by_sb / ms_b / 1e6, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': by_sb / ms_b / 1e6 / HBM_PEAK_GBS, 'launches_per_step': 1})
        L = model.n_layer
        xs = torch.randn(tp, d, device=dev) * 0.5
        Ws = [[torch.randn(d, d, device=dev) / 8, torch.randn(d, device=dev), torch.randn(d, d, device=dev) / 8,
               torch.randn(d, device=dev)] for _ in range(L)]
        outs = [[torch.empty(tp, d, device=dev) for _ in range(3)] for _ in range(L)]
        pp = (_ct.c_void_p * (4 * L))(*[t.data_ptr() for w in Ws for t in w])
        oo = (_ct.c_void_p * (3 * L))(*[t.data_ptr() for o in outs for t in o])
        fn2 = lambda: _lib.call('pamnet_mlp2_fwd_multi_f32', _lib.ptr(xs), tp, L, pp, oo, _lib.stream_of(xs))
        fn2()
        ms2, _ = event_time_ms(fn2, 10, 3)
        by2 = 4.0 * d * tp * (1 + 3 * L)                   # the rows once + z1, z2, y of every layer
        fl2 = 2.0 * 2.0 * d * d * tp * L
        res.append({'kernel': 'mlp2_fwd_kernel<7, 8> (the triplet / pair MLPs of all %d layers on the same rows, one launch)' % L,
                    'bound': 'hbm', 'rows': tp, 'bytes_per_launch': by2, 'us_per_launch': ms2 * 1e3, 'achieved': by2 / ms2 / 1e6,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': by2 / ms2 / 1e6 / HBM_PEAK_GBS,
                    'fp32_equivalent_tflops': fl2 / ms2 / 1e9, 'frac_bf16x6': fl2 / ms2 / 1e9 / (BF16_MFMA_PEAK_TFLOPS / 6.0),
                    'launches_per_step': 1})
    except Exception as ex:                                 # noqa: BLE001 -- side entries
        res.append({'kernel': 'input stage entries', 'error': '%s: %s' % (type(ex).__name__, ex)})
    # The fused edge MLP -> segment-sum kernels of the global layer (csrc/edge_agg.hip) on the same graph.  The backward forms
    # wait for HBM (priced against it); the forward issues on the bf16 matrix pipe (priced against dense bf16 / 6).
    # Algorithmic bytes (profiles/r05_edge_agg_pmc.json measures the traffic against them):
    #   backward: z, ea in; d z, d ea out; d e read-modify-write (accumulate = 1); indices; d_agg in, d P_i out
    #   forward (training): e in; z, ea saved; indices; x1, P_i, P_j in, x2 out
    from pamnet_amd import lib
    D = d
    rnd = lambda *sh: torch.randn(*sh, device=dev) * 0.5
    Wm, bm, Wea = rnd(D, 3 * D) / 8, rnd(D), rnd(D, D) / 8
    st = lib.stream_of(Wm)
    csr = g.glob
    e, Pi, Pj, x1, d_agg = rnd(m, D), rnd(n, D), rnd(n, D), rnd(n, D), rnd(n, D)
    z, ea, dz, dea, d_e = (torch.empty(m, D, device=dev) for _ in range(5))
    out, dPi = torch.empty(n, D, device=dev), torch.empty(n, D, device=dev)
    cuts = torch.empty(257, dtype=torch.int32, device=dev)
    lib.call('pamnet_seg_cuts_i32', lib.ptr(csr.ptr), lib.ptr(csr.row_of), n, m, lib.ptr(cuts), None, st)
    we = Wm.data_ptr() + 4 * 2 * D

    def fwd():
        lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), m, n, we, 3 * D, lib.ptr(bm), lib.ptr(Wea), D, lib.ptr(Pi),
                 lib.ptr(Pj), lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(csr.col), lib.ptr(cuts), lib.ptr(x1), lib.ptr(z),
                 lib.ptr(ea), lib.ptr(out), st)

    def bwd():
        lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), m, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(cuts),
                 lib.ptr(z), lib.ptr(ea), we, 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 1, lib.ptr(dPi), st)

    import ctypes
    need, slots = ctypes.c_int64(0), ctypes.c_int64(0)
    lib.call('pamnet_global_edge_agg_wg_floats', m, ctypes.addressof(need), ctypes.addressof(slots))
    partial = torch.empty(int(need.value), device=dev)

    def bwd_wg():          # what the step runs at this shape since round 5: the backward WITH the step's own weight gradients
        lib.call('pamnet_global_edge_agg_bwd_wg_f32', lib.ptr(d_agg), m, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(cuts),
                 lib.ptr(z), lib.ptr(ea), lib.ptr(e), we, 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(d_e), 1, lib.ptr(dPi),
                 lib.ptr(partial), st)

    fwd()
    d_e.zero_()
    by_f = 4.0 * D * m + 8.0 * m + 4.0 * D * n * 4 + 8.0 * D * m
    by_b = 4.0 * D * m * 6 + 4.0 * m + 4.0 * (n + 1) + 4.0 * D * n * 2
    # fused backward + weight gradients: z, ea, e in; d z out; d e read-modify-write; indices; d_agg in, d P_i (zero-fill +
    # sums) out; the partial tiles out.  FLOPs: the two dX GEMMs + the two dW products (4 d^2 per row each pair).
    by_w = 4.0 * D * m * 6 + 4.0 * m + 4.0 * (n + 1) + 4.0 * D * n * 3 + 4.0 * need.value
    bf16x6_peak = BF16_MFMA_PEAK_TFLOPS / 6.0
    for name, fn, by, gf, bound, per_step in (
            ('global_edge_agg_bwd_wg_kernel (edge MLP backward + target-side reduction + the step\'s dW_e / dW_ea / db_m, one kernel: '
             'the form the step runs at this shape)', bwd_wg, by_w, 4.0 * D * D * m * 4, 'hbm', model.n_layer),
            ('global_edge_agg_bwd_kernel<3, PRE, pieces> (the same backward without the weight gradients: small batches)', bwd, by_b,
             4.0 * D * D * m * 2, 'hbm', 0),
            ('global_edge_agg_fwd_pp_kernel<SAVE> (edge MLP -> node segment-sum, training form; round 6: the two halves of a workgroup in '
             'opposite phases, node sums by a walking wave -- the form the step runs from 131 072 edges)', fwd, by_f,
             4.0 * D * D * m, 'mfma(bf16x6)', model.n_layer)):
        for _ in range(3):
            fn()
        ms, ms_min = event_time_ms(fn, 20, 5)
        tf = gf / ms / 1e9
        ent = {'kernel': name, 'bound': bound, 'rows_in': int(m), 'rows_out': int(n), 'bytes_per_launch': by,
               'us_per_launch': ms * 1e3, 'gbs': by / ms / 1e6, 'hbm_frac': by / ms / 1e6 / HBM_PEAK_GBS,
               'best_group_gbs': by / ms_min / 1e6, 'fp32_equivalent_tflops': tf, 'peak_bf16x6': bf16x6_peak,
               'frac_bf16x6': tf / bf16x6_peak, 'launches_per_step': per_step}
        # `frac` against the bound named: the forward issues on the bf16 matrix pipe (six piece products per fp32 product:
        # dense bf16 peak / 6; 0.40 GB in ~380 us is not a memory-bound kernel -- r04 verdict), the backward forms wait for HBM
        if bound == 'hbm':
            ent.update(achieved=by / ms / 1e6, peak=HBM_PEAK_GBS, unit='GB/s', frac=by / ms / 1e6 / HBM_PEAK_GBS)
        else:
            ent.update(achieved=tf, peak=bf16x6_peak, unit='TFLOP/s (fp32-equivalent)', frac=tf / bf16x6_peak)
        res.append(ent)
    return res


def parity_beside_baseline(dev, args):
    """Part of the cpu_baseline leg: the CPU port's outputs (fp64 and fp32) on the batches it is timed / sampled on next
    to the HIP path's -- the checker role of oracle/, never the thing measured.  QM9 configs[0] (B=32, d=128, L=6) and one
    8-complex shard of the PDBbind configuration (d=128, L=3), whose pooled output is complex - pocket - ligand (~1000x
    cancellation): reported raw (max|d| / max|out|) and relative to the summed per-complex magnitude."""
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    res = {}
    err = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    for tag, cfg, b in (
            ('qm9_b32_d%d_l%d' % (args.dim, args.n_layer),
             O.Config(dataset='QM9', dim=args.dim, n_layer=args.n_layer, cutoff_l=5.0, cutoff_g=5.0), synth.qm9_batch(0, 0, 32)),
            ('pdbbind_b8_d128_l3', O.Config(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0),
             synth.collate([synth.pdbbind_complex(1, i) for i in range(8)]))):
        sd = O.init_state_dict(cfg, seed=3)
        model = models.PAMNet(models.Config(cfg.dataset, cfg.dim, cfg.n_layer, cfg.cutoff_l, cfg.cutoff_g))
        model.load_state_dict(sd, strict=True)
        model = model.to(dev)
        pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
        with torch.no_grad():
            hip = model(b.to(dev)).cpu()
            inter = {}
            x64 = b.x.double() if cfg.dataset == 'PDBbind' else b.x
            r64 = O.pamnet_forward({k: v.double() for k, v in sd.items()}, cfg, x64, b.batch, pos, ei, dtype=torch.float64,
                                   intermediates=inter)
            r32 = O.pamnet_forward(sd, cfg, b.x, b.batch, pos, ei)
        rec = {'hip_vs_cpu_fp64': err(hip, r64), 'cpu_fp32_vs_cpu_fp64': err(r32, r64)}
        if cfg.dataset == 'PDBbind':
            pin = inter['pool_in'].abs()
            scale = max(float(pin[b.batch == k].sum()) for k in range(int(b.batch.max()) + 1))
            rec['hip_vs_cpu_fp64_over_summed_magnitude'] = float((hip.double() - r64).abs().max()) / scale
            rec['cpu_fp32_vs_cpu_fp64_over_summed_magnitude'] = float((r32.double() - r64).abs().max()) / scale
            rec['note'] = 'pooled output = complex - pocket - ligand: raw figures are dominated by the cancellation'
        res[tag] = rec
    return res


def cpu_baseline(args, seconds):
    """The oracle (pure-torch CPU port of the reference forward, oracle/pamnet_oracle.py) timed on the host cores:
    BASELINE.json configs[0]: B=32, d=128, L=6, forward+backward, bounded to ~`seconds` of CPU work."""
    from oracle import pamnet_oracle as O
    from pamnet_amd import synth
    cores = os.cpu_count() or 1
    cfg = O.Config(dataset='QM9', dim=args.dim, n_layer=args.n_layer, cutoff_l=5.0, cutoff_g=5.0)
    sd = O.as_params(O.init_state_dict(cfg, seed=0))
    b = synth.qm9_batch(0, 0, 32)

    def one(train):
        out = O.pamnet_forward(sd, cfg, b.x, b.batch, b.pos, b.edge_index)
        if train:
            for p in sd.values():
                p.grad = None
            torch.nn.functional.l1_loss(out, b.y).backward()

    # torch's intra-op pool oversubscribes badly on many-core hosts for these small ops (256 threads: ~400 s/step):
    # calibrate the thread count on one forward each and keep the fastest -- the baseline gets its best configuration.
    best = None
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            one(False)
            t0 = time.time()
            one(False)
            dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
        if dt > 5.0:
            break
    threads = best[1]
    torch.set_num_threads(threads)
    one(True)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds * 0.7 or n < 2:
        one(True)
        n += 1
    train_mps = 32.0 * n / (time.time() - t0)
    t0, n = time.time(), 0
    with torch.no_grad():
        while time.time() - t0 < seconds * 0.3 or n < 2:
            one(False)
            n += 1
    fwd_mps = 32.0 * n / (time.time() - t0)
    return dict(value=train_mps, unit='molecules/s', cores=threads, host_cores=cores, kind='port',
                sample='oracle fwd+bwd, QM9-schema B=32 d=%d L=%d, ~%ds on %d threads (best of 8/16/32/64)' % (args.dim, args.n_layer, int(seconds), threads),
                forward_only=fwd_mps)


def self_launch(args, result_out):
    """`python bench.py --gpus N` without a launcher: run N ranks of this file under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1 at a port the kernel hands out) and pass rank 0's JSON line through.  Returns the exit code."""
    import socket
    import subprocess
    if not args.cpu_dry_run and not args.share_gpu and torch.cuda.device_count() < args.gpus:
        sys.stderr.write('bench.py: --gpus %d but only %d visible\n' % (args.gpus, torch.cuda.device_count()))
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    for l in lines[-1:]:                                       # exactly one line on stdout, as with a launcher
        result_out.write(l + '\n')
    result_out.flush()
    return proc.returncode if (proc.returncode or lines) else 1


def reference_loop_unchanged(dev, args, batches):
    """main_qm9.py:103-118 verbatim on the HIP model -- no Trainer, no input pipeline -- in both forms of the model's parameter
    interface: ~390 reference-named tensors (default), and PAMNET_FLAT_PARAMS=1 (ONE flat parameter for optimiser / clip / EMA;
    state_dict() unchanged; models._FlatView).  Same loop text, same numbers (tests/test_train_golden.py)."""
    import models
    from torch.nn.utils import clip_grad_norm_
    from utils import EMA
    nb = len(batches)

    def run(flat):
        old = os.environ.get('PAMNET_FLAT_PARAMS')
        os.environ['PAMNET_FLAT_PARAMS'] = '1' if flat else '0'
        try:
            torch.manual_seed(1234)
            model = models.PAMNet(models.Config(dataset='QM9', dim=args.dim, n_layer=args.n_layer, cutoff_l=5.0,
                                                cutoff_g=5.0)).to(dev)
            optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, amsgrad=False)
            ema = EMA(model, decay=0.999)
            model.train()

            def step(data):
                optimizer.zero_grad()
                output = model(data)
                loss = torch.nn.functional.l1_loss(output, data.y)
                loss_item = loss.item() * data.num_graphs              # main_qm9.py:109 (a host read-back per step)
                loss.backward()
                clip_grad_norm_(model.parameters(), max_norm=1000, norm_type=2)
                optimizer.step()
                ema(model)
                return loss_item

            for i in range(5):
                step(batches[i % nb])
            torch.cuda.synchronize()
            n = min(args.steps, 100)
            t0 = time.perf_counter()
            for i in range(n):
                step(batches[i % nb])
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, n, len(list(model.parameters()))
        finally:
            if old is None:
                os.environ.pop('PAMNET_FLAT_PARAMS', None)
            else:
                os.environ['PAMNET_FLAT_PARAMS'] = old

    ms, n, n_par = run(False)
    ms_flat, _, n_flat = run(True)
    return {'reference_loop_ms_per_step': ms_flat, 'molecules_per_s': args.batch_per_gpu / (ms_flat / 1e3), 'steps': n,
            'parameters_seen_by_the_loop': n_flat,
            'per_tensor_interface': {'reference_loop_ms_per_step': ms, 'molecules_per_s': args.batch_per_gpu / (ms / 1e3),
                                     'parameters_seen_by_the_loop': n_par},
            'note': 'main_qm9.py:103-118 verbatim on the HIP model: torch.optim.Adam, loss.item(), autograd backward, '
                    'clip_grad_norm_, utils.EMA; no Trainer, no input pipeline.  Headline of this object: PAMNET_FLAT_PARAMS=1 '
                    '(model.parameters() is one flat tensor, state_dict() keeps the reference keys); per_tensor_interface: the '
                    'default (~390 tensors through torch\'s multi-tensor Adam / clip / EMA: host-bound)'}


def init_group(backend, args, **kw):
    """dist.init_process_group with a bounded rendezvous and a ONE-LINE diagnosis instead of a hang or a traceback wall: nobody
    can run the N > 1 RCCL path before the driver does, so whatever goes wrong there has to explain itself."""
    import datetime
    try:
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=args.init_timeout), **kw)
    except Exception as e:                                              # noqa: BLE001 -- reported, then fatal
        sys.exit('bench.py: %s process group did not form within %.0f s (RANK=%s WORLD_SIZE=%s MASTER_ADDR=%s MASTER_PORT=%s '
                 'LOCAL_RANK=%s HSA_ENABLE_IPC_MODE_LEGACY=%s visible devices=%d): %s: %s'
                 % (backend, args.init_timeout, os.environ.get('RANK'), os.environ.get('WORLD_SIZE'),
                    os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'), os.environ.get('LOCAL_RANK'),
                    os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), torch.cuda.device_count() if torch.cuda.is_available() else 0,
                    type(e).__name__, str(e).splitlines()[0] if str(e) else ''))


def describe_ranks(args, world, rank, dev, dry):
    """Who is in the job: every rank's process, device index, PCI address and device name, gathered over the process group the
    step uses; a one-element all-reduce over the same group counts the ranks the transport really reaches.  Fails loudly when
    the group is not what --gpus promised: wrong size, a rank that did not get a device of its own, a sum that is not N."""
    import socket
    info = {'rank': rank, 'pid': os.getpid(), 'host': socket.gethostname(), 'local_rank': int(os.environ.get('LOCAL_RANK', '0'))}
    if not dry:
        pr = torch.cuda.get_device_properties(dev)
        info['device'] = int(torch.cuda.current_device())
        info['name'] = pr.name
        bus = [getattr(pr, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')]
        info['pci'] = ('%04x:%02x:%02x' % tuple(bus)) if all(b is not None for b in bus) else None
        info['uuid'] = str(getattr(pr, 'uuid', '')) or None
    got = dist.get_world_size()
    if got != args.gpus:
        sys.exit('bench.py: the process group has %d ranks, --gpus says %d' % (got, args.gpus))
    everyone = [None] * world
    dist.all_gather_object(everyone, info)
    one = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(one)
    if int(one.item()) != world:
        sys.exit('bench.py: an all-reduce of ones over %d ranks returned %s' % (world, one.item()))
    backend = dist.get_backend()
    if not dry and not args.share_gpu:
        if backend != 'nccl':
            sys.exit('bench.py: backend is %s, expected nccl (= RCCL on ROCm)' % backend)
        ids = [(e['host'], e['pci'] or e['uuid'] or e['device']) for e in everyone]
        if len(set(ids)) != world:
            sys.exit('bench.py: %d ranks but only %d distinct devices: %s (one rank per GPU is the contract; --share-gpu is the '
                     'explicit flow check on one device)' % (world, len(set(ids)), ids))
    return {'backend': backend + (' (RCCL)' if backend == 'nccl' else ''), 'rccl_ranks': got if backend == 'nccl' else 0,
            'ranks': got, 'allreduce_of_ones': int(one.item()),
            'devices': [{k: e.get(k) for k in ('rank', 'host', 'pid', 'device', 'pci', 'name')} for e in everyone]}


def guarded(fn, *a):
    """A side leg of the line: its failure is reported in its field, the line is still printed."""
    try:
        return fn(*a)
    except Exception as exc:                      # noqa: BLE001
        return {'error': '%s: %s' % (type(exc).__name__, exc)}


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on fd 1 (flushed at exit,
    # i.e. after our line), so fd 1 is pointed at stderr for everything but that line.
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on a free
        # local port; the ranks run this same file with the same arguments and rank 0's JSON line is passed through
        sys.exit(self_launch(args, result_out))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU: python -m torch.distributed.run '
                 '--nproc-per-node %d bench.py --gpus %d, or plainly python bench.py --gpus %d)'
                 % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    dry = args.cpu_dry_run
    if dry:
        dev = torch.device('cpu')
        if world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            init_group('gloo', args)
    else:
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
        if world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if args.share_gpu:
                init_group('gloo', args)
            else:
                if local_rank >= torch.cuda.device_count():
                    sys.exit('bench.py: LOCAL_RANK=%d but %d devices are visible' % (local_rank, torch.cuda.device_count()))
                init_group('nccl', args, device_id=dev)
        elif args.force_comm:
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1, device_id=dev)

    who = describe_ranks(args, world, rank, dev, dry) if world > 1 else None

    from pamnet_amd import synth
    from pamnet_amd.train import Trainer, shard_range
    torch.manual_seed(1234)                        # identical random-init weights on every rank
    if dry:
        sys.path.insert(0, os.path.join(REPO, 'tests'))
        from standin import LayeredStandIn          # test infrastructure: only reachable through --cpu-dry-run
        model = LayeredStandIn(n_layer=args.n_layer)
    else:
        import models
        from pamnet_amd import lib
        lib.load()
        cfg = models.Config(dataset='QM9', dim=args.dim, n_layer=args.n_layer, cutoff_l=5.0, cutoff_g=5.0)
        model = models.PAMNet(cfg).to(dev)
    overlap = os.environ.get('PAMNET_OVERLAP_COMM', '1') != '0'
    trainer = Trainer(model, lr=1e-4, world_size=world,
                      overlap_comm=(('force' if overlap else 'force_single') if args.force_comm else overlap),
                      native_optimizer=os.environ.get('PAMNET_NATIVE_OPT', '1') != '0',
                      n_buckets=int(os.environ.get('PAMNET_BUCKETS', '3')))
    B = args.batch_per_gpu
    gB = B * world
    # resident batches: global batch k = molecules [k*gB, (k+1)*gB); this rank's shard = its contiguous slice
    batches = []
    for k in range(args.n_batches):
        lo, hi = shard_range(gB, rank, world)
        assert hi - lo == B
        batches.append(synth.qm9_batch(0, k * gB + lo, B).to(dev))
    if not dry:
        torch.cuda.synchronize()

    def sync():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    nb = len(batches)
    # Input pipelining: while step i runs, the graph of batch i+1 is constructed on a side stream (what a loader
    # worker does); every step still builds its own graph inside the timed region -- nothing is cached across steps.
    for i in range(args.warmup):
        trainer.step(batches[i % nb], global_graphs=gB, next_data=batches[(i + 1) % nb])
    sync()
    # one HIP event per step on the stream the step runs on (a marker, not a synchronisation): the spread of the
    # per-step durations travels with the line
    marks = [] if not dry else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        k = args.warmup + i
        trainer.step(batches[k % nb], global_graphs=gB, next_data=batches[(k + 1) % nb])
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
    sync()
    dt = time.perf_counter() - t0
    spread = None
    if marks and len(marks) > 2:
        per = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        q = lambda f: per[min(len(per) - 1, int(f * len(per)))]
        spread = {'p10': q(0.10), 'p50': q(0.50), 'p90': q(0.90), 'max': per[-1],
                  'note': 'ms between consecutive per-step HIP events inside the timed region'}
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank_ms = None
    if world > 1:
        mine = t / args.steps * 1e3
        each = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(each, mine)
        per_rank_ms = [float(v[0]) for v in each]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    ms_per_step = dt / args.steps * 1e3
    value = gB * args.steps / dt

    # Exposed all-reduce time (N > 1, or N = 1 with --force-comm): the same loop with the gradient exchange skipped; the
    # ranks' parameters drift apart from here on, which nothing below depends on (timings only).
    exposed = None
    if (world > 1 or args.force_comm) and not dry:
        ksteps = min(args.steps, 100)
        orig_sync = trainer.sync_gradients
        trainer.sync_gradients = lambda: None
        for i in range(3):
            trainer.step(batches[i % nb], global_graphs=gB, next_data=batches[(i + 1) % nb])
        sync()
        t0 = time.perf_counter()
        for i in range(ksteps):
            trainer.step(batches[i % nb], global_graphs=gB, next_data=batches[(i + 1) % nb])
        sync()
        tn = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        trainer.sync_gradients = orig_sync
        no_comm_ms = float(tn[0]) / ksteps * 1e3
        exposed = {'allreduce_exposed_ms': ms_per_step - no_comm_ms, 'ms_per_step_without_allreduce': no_comm_ms,
                   'buckets': (len(trainer._buckets) + 1) if trainer._buckets else 1,
                   'gradient_bytes': int(trainer.fp.grad.numel()) * 4,
                   'note': 'step time with minus without sync_gradients(), max over ranks, %d steps' % ksteps}

    comm = None
    if world > 1 or exposed is not None:
        comm = dict(exposed or {})
        if who is not None:
            comm.update(who)
        if per_rank_ms is not None:
            comm['per_rank_ms_per_step'] = {'min': min(per_rank_ms), 'max': max(per_rank_ms), 'all': per_rank_ms}
        if args.n1_ms:
            # weak scaling: per-GPU work is fixed, so the ideal N-rank step takes the N = 1 step's time
            comm['efficiency_vs_n1'] = args.n1_ms / ms_per_step
            comm['n1_ms_per_step'] = args.n1_ms
    if dry:
        if rank == 0:
            line = {'metric': 'DRY RUN on CPU with a stand-in module -- control-flow check, NOT a measurement',
                    'value': value, 'unit': 'molecules/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                    'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                    'dtype': 'f32', 'data': 'synthetic', 'dry_run': True, 'comm': comm,
                    'config': {'workload': 'cpu dry run, %d molecules/rank' % B, 'global_batch': gB,
                               'parallelism': 'dp%d (molecule-sharded, gloo all-reduce of flat grad)' % world}}
            result_out.write(json.dumps(line) + '\n')
            result_out.flush()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # forward-only rate (reported beside the training rate; not `value`): the product's forward-only loop is
    # train.predict -- the counterpart of test() in main_qm9.py:29-37 -- which builds the graph of batch i+1 on a side stream
    # while batch i runs (every batch's graph is still built inside the timed region); the plain loop of model(batch)
    # calls, each stalling on its own graph's size round trip, is reported beside it.
    from pamnet_amd.train import predict
    fsteps = min(args.steps, 200)
    with torch.no_grad():
        for i in range(2):
            model(batches[i % nb])
        sync()
        t0 = time.perf_counter()
        for i in range(fsteps):
            model(batches[i % nb])
        sync()
        fwd_plain_ms = (time.perf_counter() - t0) / fsteps * 1e3
        for _ in predict(model, (batches[i % nb] for i in range(4))):
            pass
        sync()
        t0 = time.perf_counter()
        for _ in predict(model, (batches[i % nb] for i in range(fsteps))):
            pass
        sync()
        fwd_ms = (time.perf_counter() - t0) / fsteps * 1e3

    # Zero-host-sync path (pamnet_amd/store.py): the same molecules resident on the device as one
    # concatenated dataset, every batch collated by one gather launch INSIDE the timed loops and carrying its data-dependent
    # sizes as host integers (per-molecule counts taken once per dataset) -- graph construction reads nothing back.
    # Reported beside the classic numbers, never as `value`.
    zero_sync = None
    if not args.cpu_dry_run and world == 1:       # a side measurement: single-GPU runs only (nothing the scaling runs need)
        try:                                      # (a side leg must never cost the run its result line)
            from pamnet_amd.store import MoleculeStore
            lo, hi = shard_range(gB, rank, world)
            mols = [synth.qm9_molecule(0, k * gB + lo + i) for k in range(args.n_batches) for i in range(B)]
            store = MoleculeStore(mols, dev).prepare_for(model)
            idx = [list(range(k * B, (k + 1) * B)) for k in range(args.n_batches)]
            with torch.no_grad():
                for i in range(3):
                    model(store.collate(idx[i % nb]))
                sync()
                t0 = time.perf_counter()
                for i in range(fsteps):
                    model(store.collate(idx[i % nb]))
                sync()
                zs_fwd = (time.perf_counter() - t0) / fsteps * 1e3
            model.verify()
            zsteps = min(args.steps, 200)
            nxt = store.collate(idx[0])
            for i in range(5):
                cur, nxt = nxt, store.collate(idx[(i + 1) % nb])
                trainer.step(cur, global_graphs=gB, next_data=nxt)
            sync()
            t0 = time.perf_counter()
            for i in range(zsteps):
                cur, nxt = nxt, store.collate(idx[(i + 1) % nb])
                trainer.step(cur, global_graphs=gB, next_data=nxt)
            sync()
            zs_step = (time.perf_counter() - t0) / zsteps * 1e3
            trainer.drain()
            zero_sync = {'forward_ms_unpipelined': zs_fwd, 'forward_only_molecules_per_s': gB / (zs_fwd / 1e3),
                         'train_ms_per_step': zs_step, 'train_molecules_per_s': gB / (zs_step / 1e3),
                         'note': 'resident dataset, one device-side collate launch per batch inside the timed loop, sizes from '
                                 'per-molecule counts: no device->host read in forward, graph construction or step '
                                 '(tests/test_store.py runs the forward under torch.cuda.set_sync_debug_mode("error"))'}
        except Exception as exc:                  # noqa: BLE001
            zero_sync = {'error': '%s: %s' % (type(exc).__name__, exc)}

    # The reference's loop body UNCHANGED on the HIP model (main_qm9.py:103-118): torch.optim.Adam + loss.item() +
    # loss.backward() through autograd + clip_grad_norm_ + utils.EMA, one graph-size round trip per forward and no input
    # pipeline -- what "drops into main_qm9.py unchanged" delivers before anything of pamnet_amd.train is adopted.  A side
    # field on a model of its own (same architecture, same batches); never `value`.
    ref_loop = None
    if not args.cpu_dry_run and world == 1:
        ref_loop = guarded(reference_loop_unchanged, dev, args, batches)

    # What graph construction costs the step although it runs beside it (a side-stream launch displaces a workgroup
    # of the model's full-chip launches): the same steps on graphs prepared once and reused.  A bound, never `value` -- the
    # timed region above rebuilds the graph of every batch.
    graph_cost = None
    if not args.cpu_dry_run and world == 1:
        try:
            keep = {}
            for b in batches:
                model.prepare(b)
                keep[id(b)] = b._pamnet_prepared
                b._pamnet_prepared = None
            build = model._graph
            model._graph = lambda data: keep[id(data)]
            try:
                csteps = min(args.steps, 100)
                for i in range(5):
                    trainer.step(batches[i % nb], global_graphs=gB)
                sync()
                t0 = time.perf_counter()
                for i in range(csteps):
                    trainer.step(batches[i % nb], global_graphs=gB)
                sync()
                cached_ms = (time.perf_counter() - t0) / csteps * 1e3
            finally:
                del model._graph                                   # back to the class's method
            assert model._graph.__func__ is build.__func__
            graph_cost = {'ms_per_step_graphs_cached': cached_ms, 'ms_per_step': ms_per_step,
                          'note': 'bound only: the same training steps on graphs prepared once and reused (no graph construction); '
                                  '`value` rebuilds the graph of every batch on the side stream'}
        except Exception as exc:                  # noqa: BLE001
            graph_cost = {'error': '%s: %s' % (type(exc).__name__, exc)}
            if '_graph' in model.__dict__:
                del model._graph

    if rank == 0:
        with torch.no_grad():
            model(batches[0])
        g = model._graph_cache
        line = {
            'metric': METRIC,
            'value': value, 'unit': 'molecules/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'QM9-schema synthetic radius-graph batches, PAMNet dim=%d n_layer=%d, %d molecules/GPU '
                                   '(BASELINE configs[1]), full training step fwd+bwd+allreduce+clip+Adam+EMA'
                                   % (args.dim, args.n_layer, B),
                       'global_batch': gB, 'parallelism': 'dp%d (molecule-sharded, RCCL all-reduce of flat grad)' % world,
                       'nodes_per_batch': int(g.n), 'global_edges': int(g.glob.m), 'local_edges': int(g.loc.m),
                       'triplets': int(g.n_trip), 'pairs': int(g.n_pair)},
            'timed_region_s': dt, 'step_ms_spread': spread, 'comm': comm,
            'forward_only_molecules_per_s': gB / (fwd_ms / 1e3), 'forward_ms': fwd_ms,
            'forward_ms_unpipelined': fwd_plain_ms,
            'zero_host_sync': zero_sync,
            'graph_construction_cost': graph_cost,
            'reference_loop_unchanged': ref_loop,
            'mfma': mfma_summary(args, g, ms_per_step),
        }
        if args.share_gpu:
            line['shared_gpu'] = 'all %d ranks on one device, gloo transport: a flow check, NOT a measurement' % world
        if not args.no_rooflines:
            roof = scatter_add_roofline(dev, g, args.dim, args.stream_gb)
            s = roof['streamed']
            pmc = pmc_record('r06_scatter_add_pmc') or pmc_record('r05_scatter_add_pmc') or pmc_record('r04_scatter_add_pmc')
            cal = guarded(box_calibration, dev)
            prof = committed_kernel_avg('r06_scatter_add_kernel_stats.txt', 'segment_sum_kernel') \
                or committed_kernel_avg('r05_scatter_add_kernel_stats.txt', 'segment_sum_kernel')
            line['roofline'] = {
                'bound': 'hbm', 'kernel': 'segment_sum_kernel (pamnet_segment_sum_f32, scatter-add)',
                'achieved': s['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': s['gbs'] / HBM_PEAK_GBS,
                'traffic': (pmc['traffic_bytes_per_launch'] * s['bytes'] / pmc['algorithmic_bytes_per_launch']) if pmc else None,
                'traffic_over_algorithmic': pmc['traffic_over_algorithmic'] if pmc else None,
                'traffic_source': pmc['source'] if pmc else 'no PMC record under profiles/',
                'bytes_per_launch': s['bytes'], 'ms_per_launch': s['ms'], 'best_group_gbs': s['gbs_best'],
                'shape': '[%d,%d]->[%d,%d] streamed (%.2f GB)' % (s['rows_in'], args.dim, s['rows_out'], args.dim, s['bytes'] / 1e9),
                # the same kernel, the same bytes, by the committed rocprofv3 average of the evidence run (another box): the two
                # fractions differ by the box, not by the method
                'frac_rocprofv3_committed': (prof[1] / (prof[0] * 1e-6) / HBM_PEAK_GBS) if prof else None,
                'rocprofv3_committed': {'average_us': prof[0], 'algorithmic_gb': prof[1]} if prof else None,
                'box_calibration': cal,
                'frac_of_box_copy': (s['gbs'] / cal['copy_gbs']) if isinstance(cal, dict) and 'copy_gbs' in cal else None,
                'at_workload_shape': roof['workload'],
                'in_config': [{'config': 'BASELINE configs[1] (QM9 B=128): the global aggregation shape [E_g, d] -> [N, d]; every '
                                         'operand is Infinity-Cache resident (latency bound, not an HBM number)',
                               'rows_in': roof['workload']['rows_in'], 'rows_out': roof['workload']['rows_out'],
                               'us_per_launch': roof['workload']['ms'] * 1e3, 'gbs': roof['workload']['gbs'],
                               'frac': roof['workload']['gbs'] / HBM_PEAK_GBS}]}
            line['step_kernels'] = guarded(step_kernel_rooflines, dev, g, args.dim, args.n_layer)
        if world == 1 and not args.no_other_configs:
            line['other_configs'] = guarded(other_configs, dev)
            try:                                            # the scatter-add at the shape a configuration streams from HBM
                k0 = line['other_configs']['pdbbind_b32_d128_l3']['kernels'][0]
                line['roofline']['in_config'].append({
                    'config': 'BASELINE configs[3] (PDBbind B=32): the source-side reduction of the global layer\'s backward, '
                              'gather (transposed-CSR) form, as the step meets it (cold caches)',
                    'rows_in': k0['rows_in'], 'rows_out': k0['rows_out'], 'us_per_launch': k0['us_per_launch_cold_caches'],
                    'gbs': k0['bytes_per_launch'] / k0['us_per_launch_cold_caches'] / 1e3, 'frac': k0['frac_cold_caches'],
                    'us_per_launch_warm': k0['us_per_launch'], 'frac_warm': k0['frac']})
            except Exception:                               # noqa: BLE001 -- a side field
                pass
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args, args.cpu_seconds)
            line['cpu_baseline']['parity'] = guarded(parity_beside_baseline, dev, args)
            line['speedup_vs_cpu'] = value / line['cpu_baseline']['value']
        # The driver keeps the first ~2 000 characters of the line: the figures a reader needs first -- the other BASELINE
        # configurations' step times, the roofline fraction, the CPU baseline -- as a compact object right behind `config`.
        summary = {}
        oc = line.get('other_configs')
        if isinstance(oc, dict):
            for tag, key in (('pdbbind_b32_d128_l3', 'pdbbind'), ('rna_b8_d16_l1', 'rna'), ('pamnet_s_qm9_b128_d128_l6', 'pamnet_s')):
                v = oc.get(tag)
                if isinstance(v, dict) and 'train_ms_per_step' in v:
                    summary[key + '_train_ms'] = round(v['train_ms_per_step'], 3)
                    summary[key + '_store_train_ms'] = round(v['store_train_ms_per_step'], 3)
                    summary[key + '_forward_ms'] = round(v['forward_ms'], 3)
        if 'roofline' in line:
            summary['scatter_add_hbm_frac'] = round(line['roofline']['frac'], 4)
        if 'cpu_baseline' in line:
            summary['cpu_molecules_per_s'] = round(line['cpu_baseline']['value'], 1)
        if isinstance(ref_loop, dict) and 'reference_loop_ms_per_step' in ref_loop:
            summary['reference_loop_unchanged_ms'] = round(ref_loop['reference_loop_ms_per_step'], 3)
        head = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config')
        ordered = {k: line[k] for k in head}
        ordered['summary'] = summary
        for k in ('roofline', 'cpu_baseline'):
            if k in line:
                ordered[k] = line[k]
        ordered.update({k: v for k, v in line.items() if k not in ordered})
        result_out.write(json.dumps(ordered) + '\n')
        result_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

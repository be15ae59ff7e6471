#!/usr/bin/env python
"""bench.py -- molecules/sec of one PAMNet training step (QM9 schema, dim=128, n_layer=6) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch with inputs already resident in HBM: zero_grad -> forward
(device graph construction, bases, 6x(global+local) message passing, fusion, pooling) -> L1 loss -> backward -> RCCL
all-reduce of the flat gradient (N>1) -> clip -> Adam -> EMA, i.e. the reference loop main_qm9.py:103-118.
Workload (config.workload): BASELINE.json configs[1] -- 128 molecules per GPU (weak scaling: global batch 128*N).
One JSON line on rank 0: whole-job molecules/s + `roofline` (segment-sum = scatter-add kernel, HIP-event timed) +
`cpu_baseline` (the oracle = port of the reference CPU forward+backward, bounded sample, N=1 only).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch-per-gpu', type=int, default=128)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--n-layer', type=int, default=6)
    ap.add_argument('--n-batches', type=int, default=4, help='distinct resident batches cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-comm', action='store_true',
                    help='N=1 only: run the bucketed gradient all-reduce through a 1-rank RCCL group (overhead check)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--stream-gb', type=float, default=2.0, help='size of the streamed scatter-add roofline probe')
    return ap.parse_args()


def event_time_ms(fn, reps):
    """Average duration of fn() in ms with HIP events on torch's current stream (the stream the kernels launch on)."""
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


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
    ms = event_time_ms(fn, 50)
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
    for _ in range(3):
        fn()
    ms = event_time_ms(fn, 20)
    by = 4.0 * d * M + 4.0 * (R + 1) + 4.0 * d * R
    res['streamed'] = dict(rows_in=M, rows_out=R, bytes=by, ms=ms, gbs=by / ms / 1e6)
    return res


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


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on fd 1 (flushed at exit,
    # i.e. after our line), so fd 1 is pointed at stderr for everything but that line.
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    elif args.force_comm:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1, device_id=dev)

    import models
    from pamnet_amd import lib, synth
    from pamnet_amd.train import Trainer
    lib.load()

    torch.manual_seed(1234)                        # identical random-init weights on every rank
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
        lo = k * gB + rank * B
        batches.append(synth.qm9_batch(0, lo, B).to(dev))
    torch.cuda.synchronize()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nb = len(batches)
    # Input pipelining: while step i runs, the graph of batch i+1 is constructed on a side stream (what a loader
    # worker does); every step still builds its own graph inside the timed region -- nothing is cached across steps.
    for i in range(args.warmup):
        trainer.step(batches[i % nb], global_graphs=gB, next_data=batches[(i + 1) % nb])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        k = args.warmup + i
        trainer.step(batches[k % nb], global_graphs=gB, next_data=batches[(k + 1) % nb])
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    ms_per_step = dt / args.steps * 1e3
    value = gB * args.steps / dt

    # forward-only rate (reported beside the training rate; not `value`)
    with torch.no_grad():
        for i in range(2):
            model(batches[i % len(batches)])
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            model(batches[i % len(batches)])
        sync()
        fwd_ms = (time.perf_counter() - t0) / args.steps * 1e3

    line = None
    if rank == 0:
        with torch.no_grad():
            model(batches[0])
        g = model._graph_cache
        roof = scatter_add_roofline(dev, g, args.dim, args.stream_gb)
        s = roof['streamed']
        line = {
            'metric': 'molecules/sec (QM9 dim=128 n_layer=6) at 1/2/4/8 GPU; scatter-add HBM GB/s vs peak',
            'value': value, 'unit': 'molecules/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'QM9-schema synthetic radius-graph batches, PAMNet dim=%d n_layer=%d, %d molecules/GPU '
                                   '(BASELINE configs[1]), full training step fwd+bwd+allreduce+clip+Adam+EMA'
                                   % (args.dim, args.n_layer, B),
                       'global_batch': gB, 'parallelism': 'dp%d (molecule-sharded, RCCL all-reduce of flat grad)' % world,
                       'nodes_per_batch': int(g.n), 'global_edges': int(g.glob.m), 'local_edges': int(g.loc.m),
                       'triplets': int(g.n_trip), 'pairs': int(g.n_pair)},
            'forward_only_molecules_per_s': gB / (fwd_ms / 1e3), 'forward_ms': fwd_ms,
            'mfma': mfma_summary(args, g, ms_per_step),
            'roofline': {'bound': 'hbm', 'kernel': 'segment_sum_kernel (pamnet_segment_sum_f32, scatter-add)',
                         'achieved': s['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': s['gbs'] / HBM_PEAK_GBS,
                         'traffic': None, 'traffic_note': 'PMC passes (FETCH_SIZE x2 + WRITE_SIZE) in '
                         'profiles/r01_scatter_add_pmc.txt: 1.002 x algorithmic bytes',
                         'bytes_per_launch': s['bytes'], 'ms_per_launch': s['ms'],
                         'shape': '[%d,%d]->[%d,%d] streamed (%.2f GB)' % (s['rows_in'], args.dim, s['rows_out'], args.dim, s['bytes'] / 1e9),
                         'at_workload_shape': roof['workload']},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args, args.cpu_seconds)
            line['speedup_vs_cpu'] = value / line['cpu_baseline']['value']
        result_out.write(json.dumps(line) + '\n')
        result_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

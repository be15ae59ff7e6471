"""Training-step harness (pamnet_amd.train): reference loop semantics (main_qm9.py:99-118, utils/ema.py) and
molecule-sharded data parallelism.

CPU part (`-m "not gpu"`): world_size-2 `gloo` run of the Trainer on a stand-in module -- the DP algebra (shard,
pre-scale by local/global graphs, all-reduce of the flat gradient, identical update on every rank) must reproduce the
single-process global-batch step.  The HIP model itself cannot run on CPU (no fallback), so the stand-in is a plain
torch module with the same `model(data) -> [num_graphs]` contract.
GPU part: the flat / direct-gradient path of the real model equals plain autograd.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Tiny(nn.Module):
    """Per-graph scalar from node features: sum-pool of an MLP (graphs are independent units, like PAMNet)."""

    def __init__(self):
        super().__init__()
        self.a, self.b = nn.Linear(6, 16), nn.Linear(16, 1)

    def forward(self, data):
        h = self.b(torch.tanh(self.a(data.x))).view(-1)
        return torch.zeros(data.num_graphs, dtype=h.dtype).index_add_(0, data.batch, h)


from standin import LayeredStandIn as Layered  # noqa: E402  (PAMNet-named CPU stand-in, tests/standin.py)


class D(object):
    pass


def _batch(lo, hi, seed=0):
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(3, 9, (64,), generator=g)
    xs = [torch.randn(int(s), 6, generator=g) for s in sizes]
    ys = torch.randn(64, generator=g)
    d = D()
    d.x = torch.cat(xs[lo:hi])
    d.batch = torch.repeat_interleave(torch.arange(hi - lo), sizes[lo:hi])
    d.y, d.num_graphs = ys[lo:hi], hi - lo
    return d


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out, kind):
    from pamnet_amd.train import Trainer, shard_range
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(rank)                      # DIFFERENT seeds: the trainer must broadcast rank 0's parameters
    model = Tiny() if kind == 'tiny' else Layered()
    tr = Trainer(model, lr=1e-2, world_size=world, n_buckets=3)
    if kind == 'layered':
        assert tr._buckets is not None and len(tr._buckets) >= 1      # bucketed path (sequential on CPU, same tiling)
    lo, hi = shard_range(total, rank, world)
    for step in range(3):
        tr.step(_batch(lo, hi), global_graphs=total)
    mae = tr.evaluate([_batch(lo, hi)])
    if rank == 0:
        torch.save({'flat': tr.fp.flat.clone(), 'shadow': tr.shadow.clone(), 'mae': mae}, out)
    # every rank must hold identical parameters
    ref = tr.fp.flat.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, tr.fp.flat)
    dist.destroy_process_group()


@pytest.mark.parametrize('kind,world,total', [('tiny', 2, 13), ('layered', 2, 13), ('layered', 3, 16)])
def test_dp_matches_single_process(tmp_path, kind, world, total):
    """world_size 2 / 3, uneven shards (7+6, 6+5+5): shard -> pre-scale by local/global graphs -> (bucketed) all-reduce of
    the flat gradient -> identical update == the single-process global-batch step."""
    from pamnet_amd.train import Trainer, shard_range
    shards = [shard_range(total, r, world) for r in range(world)]
    assert shards[0][0] == 0 and shards[-1][1] == total and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
    assert len({hi - lo for lo, hi in shards}) == 2                       # really uneven
    out = str(tmp_path / 'dp.pt')
    mp.spawn(_worker, args=(world, _free_port(), total, out, kind), nprocs=world, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = Tiny() if kind == 'tiny' else Layered()
    tr = Trainer(model, lr=1e-2, world_size=1)
    for step in range(3):
        tr.step(_batch(0, total))
    assert torch.allclose(got['flat'], tr.fp.flat, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got['shadow'], tr.shadow, rtol=1e-5, atol=1e-6)
    assert abs(got['mae'] - tr.evaluate([_batch(0, total)])) < 1e-5


def test_flat_layout_and_bucket_tiling():
    """FlatParams on PAMNet-named parameters: backward-completion order (layer pair L-1 first, top level last),
    contiguous layer_ranges, and plan_buckets tiling [0, numel) exactly for every bucket count."""
    from pamnet_amd.train import FlatParams, plan_buckets
    L = 5
    fp = FlatParams(Layered(n_layer=L))
    assert fp.names[0].startswith('global_layer.%d.' % (L - 1))
    assert not any(n.startswith(('global_layer.', 'local_layer.')) for n in fp.names[-4:])
    assert {n.split('.')[0] for n in fp.names[-4:]} == {'embeddings', 'rbf_g', 'mlp_rbf_g'}
    pos = 0
    for k in range(L - 1, -1, -1):                          # pairs laid out back to front, no gaps, global+local adjacent
        lo, hi = fp.layer_ranges[k]
        assert lo == pos and hi > lo
        pos = hi
        names_k = [n for n in fp.names if lo <= fp.offsets[n] < hi]
        assert {n.split('.')[1] for n in names_k} == {str(k)}
        assert {n.split('.')[0] for n in names_k} == {'global_layer', 'local_layer'}
    for nb in (1, 2, 3, 4, 5, 8):
        buckets, tail = plan_buckets(fp.layer_ranges, fp.grad.numel(), nb)
        cover = [(lo, hi) for lo, hi, _ in buckets] + [tail]
        assert cover[0][0] == 0 and cover[-1][1] == fp.grad.numel()
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        ks = [k for _, _, k in buckets]
        assert ks == sorted(ks, reverse=True) and len(buckets) <= max(0, min(nb, L) - 1) + (0 if nb >= L else 0) + nb
        for lo, hi, k in buckets:                           # a slice is complete once its lowest layer pair is done
            assert lo == fp.layer_ranges[max(kk for kk in range(L) if fp.layer_ranges[kk][0] >= lo and
                                             fp.layer_ranges[kk][1] <= hi)][0]
            assert hi == fp.layer_ranges[k][1]
    assert plan_buckets({0: (0, 4), 2: (4, 8)}, 8, 2) is None


def test_node_balanced_sharding_on_shipped_rna_sizes(golden):
    """RNA / PDBbind graphs are sharded by number of nodes, not by count.  On the node counts of the 21
    shipped RNA-Puzzles native structures (841 .. 3 823 nodes) the greedy longest-first packing stays within a few
    percent of the ideal load, where equal-count contiguous shards are off by up to ~40 %."""
    from pamnet_amd.train import balanced_shards, shard_range
    sizes = [int(v) for v in golden('rna_native')['all_num_nodes']]
    assert len(sizes) == 21 and min(sizes) == 841 and max(sizes) == 3823
    for world in (2, 3, 4, 8):
        shards = balanced_shards(sizes, world)
        assert sorted(i for s in shards for i in s) == list(range(21))           # a partition
        assert shards == balanced_shards(sizes, world)                           # deterministic
        loads = [sum(sizes[i] for i in s) for s in shards]
        ideal = sum(sizes) / world
        eq = []
        for r in range(world):
            lo, hi = shard_range(21, r, world)
            eq.append(sum(sizes[lo:hi]))
        assert max(loads) <= 1.12 * ideal, (world, loads)
        assert max(loads) <= max(eq)
    assert max(sum(sizes[i] for i in s) for s in balanced_shards(sizes, 4)) < 1.06 * sum(sizes) / 4
    assert balanced_shards([5, 5, 5], 4)[3] == []                                # more ranks than graphs


def test_node_balanced_sharding_of_pdbbind_sized_complexes_over_eight_ranks():
    """SURVEY.md 8e at N = 8 on the PDBbind-schema synthetic complexes (3-copy layout: 2 x (pocket + ligand) nodes, 320 .. 840
    per complex): 64 complexes over 8 ranks by number of nodes -- a partition, deterministic, every rank within a few percent
    of the ideal load and never worse than equal-count contiguous shards."""
    from pamnet_amd import synth
    from pamnet_amd.train import balanced_shards, shard_range
    sizes = [int(synth.pdbbind_complex(1, i)['x'].shape[0]) for i in range(64)]
    assert min(sizes) >= 300 and max(sizes) > 1.5 * min(sizes)
    shards = balanced_shards(sizes, 8)
    assert sorted(i for s_ in shards for i in s_) == list(range(64)) and shards == balanced_shards(sizes, 8)
    loads = [sum(sizes[i] for i in s_) for s_ in shards]
    ideal = sum(sizes) / 8.0
    eq = [sum(sizes[slice(*shard_range(64, r, 8))]) for r in range(8)]
    assert max(loads) <= 1.03 * ideal and max(loads) <= max(eq), (loads, eq)
    assert all(len(s_) >= 1 for s_ in shards)


def test_trainer_checkpoint_round_trip_and_mismatches():
    """Trainer.load_state_dict: a round trip is exact; a checkpoint saved WITHOUT an EMA shadow restarts the shadow
    from the loaded weights (not from the constructor's); the other optimiser mode or another flat layout raise ValueError."""
    from pamnet_amd.train import Trainer
    torch.manual_seed(2)
    b = _batch(0, 10)
    a = Trainer(Tiny(), lr=1e-2)
    for _ in range(3):
        a.step(b)
    ck = a.state_dict()
    torch.manual_seed(5)
    c = Trainer(Tiny(), lr=1e-2)
    c.load_state_dict(ck)
    assert torch.equal(c.fp.flat, a.fp.flat) and torch.equal(c.shadow, a.shadow)
    a.step(b), c.step(b)
    assert torch.equal(c.fp.flat, a.fp.flat) and torch.equal(c.shadow, a.shadow)       # the resumed run continues bit for bit
    # a no-EMA trainer's checkpoint into an EMA trainer: the shadow starts from the LOADED weights
    n = Trainer(Tiny(), lr=1e-2, ema_decay=None)
    n.step(b)
    ck2 = n.state_dict()
    assert ck2['shadow'] is None
    torch.manual_seed(9)
    e = Trainer(Tiny(), lr=1e-2)
    before = e.shadow.clone()
    e.load_state_dict(ck2)
    assert torch.equal(e.shadow, e.fp.flat) and not torch.equal(e.shadow, before)
    # the other optimiser mode / another layout: a message, not a KeyError or a silent mis-copy
    other = dict(ck)
    other.pop('optimizer', None), other.pop('exp_avg', None)
    if a.native_opt:
        other['optimizer'] = {}
    else:
        other.update(exp_avg=torch.zeros(3), exp_avg_sq=torch.zeros(3), step_count=0)
    with pytest.raises(ValueError):
        c.load_state_dict(other)
    bad = dict(ck)
    bad['shadow'] = torch.zeros(ck['shadow'].numel() + 4)
    with pytest.raises(ValueError):
        c.load_state_dict(bad)


def test_trainer_step_semantics():
    """clip at max_norm, Adam update, EMA decay = min(0.999, (1+n)/(10+n)) with n=99999 (utils/ema.py:14)."""
    from pamnet_amd.train import Trainer, WarmupExpLR
    torch.manual_seed(1)
    model, ref = Tiny(), Tiny()
    ref.load_state_dict(model.state_dict())
    tr = Trainer(model, lr=1e-2, max_grad_norm=0.05)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    shadow = {k: p.data.clone() for k, p in ref.named_parameters()}
    b = _batch(0, 10)
    for _ in range(2):
        tr.step(b)
        opt.zero_grad()
        torch.nn.functional.l1_loss(ref(b), b.y).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=0.05, norm_type=2)
        opt.step()
        for k, p in ref.named_parameters():
            shadow[k] = (1.0 - 0.999) * p.data + 0.999 * shadow[k]
    for (k, p), q in zip(ref.named_parameters(), model.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), k
    sched = WarmupExpLR(1e-4, steps_per_epoch=100)
    assert abs(sched.lr_at(0, 50) - 0.5e-4) < 1e-12 and abs(sched.lr_at(2, 0) - 1e-4 * 0.9961697) < 1e-12
    for (k, p) in model.named_parameters():
        o = tr.fp.offsets[k]
        assert torch.allclose(tr.shadow[o:o + p.numel()].view_as(p), shadow[k], rtol=1e-5, atol=1e-7), k


@pytest.mark.gpu
def test_direct_gradient_path_equals_autograd():
    import models
    from pamnet_amd import fused, synth
    from pamnet_amd.train import FlatParams
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    cfg = models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    model = models.PAMNet(cfg).to(dev)
    b = synth.qm9_batch(2, 0, 8).to(dev)
    torch.nn.functional.l1_loss(model(b), b.y).backward()              # plain autograd: no parameter allows direct writes
    ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    fp = FlatParams(model, direct=True)                                 # the permission is a property of THESE parameters
    assert all(getattr(p, '_pamnet_direct', False) for p in model.parameters())
    fp.zero_grad()
    torch.nn.functional.l1_loss(model(b), b.y).backward()
    for k, p in model.named_parameters():
        if k in ref:
            assert torch.equal(p.grad, ref[k]), k
    assert float(fp.grad.abs().sum()) > 0
    # a second model in the same process is unaffected (nothing process-global)
    other = models.PAMNet(cfg).to(dev)
    torch.nn.functional.l1_loss(other(b), b.y).backward()
    assert all(not getattr(p, '_pamnet_direct', False) for p in other.parameters())
    assert all(p.grad is not None for k, p in other.named_parameters() if k in ref)
    fp.set_direct(False)
    fp.zero_grad()
    torch.nn.functional.l1_loss(model(b), b.y).backward()              # accumulated by autograd into the zeroed views
    for k, p in model.named_parameters():
        if k in ref:
            assert torch.equal(p.grad, ref[k]), k


@pytest.mark.gpu
def test_bucketed_overlapped_allreduce_single_rank_rccl():
    """The bucketed gradient all-reduce (slices of the last layers reduced on a side stream at per-layer events recorded
    inside the engine's backward) run through a 1-rank RCCL group: same parameters after 3 steps as the plain path, and
    the buckets tile the flat buffer in backward-completion order."""
    import torch.distributed as dist
    import models
    from pamnet_amd import fused, synth
    from pamnet_amd.train import Trainer
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                                device_id=dev)
        created = True
    try:
        cfg = models.Config(dataset='QM9', dim=128, n_layer=4, cutoff_l=5.0, cutoff_g=5.0)
        batches = [synth.qm9_batch(3, 8 * i, 8).to(dev) for i in range(3)]

        def run(overlap):
            torch.manual_seed(9)
            model = models.PAMNet(cfg).to(dev)
            tr = Trainer(model, lr=1e-3, world_size=1, overlap_comm=overlap, n_buckets=3)
            for b in batches:
                tr.step(b, global_graphs=8)
            torch.cuda.synchronize()
            return tr, tr.fp.flat.clone()

        tr0, plain = run(False)
        assert tr0._buckets is None
        tr1, over = run('force')
        assert tr1._buckets is not None and tr1._stack_ctx.recorded
        assert tr0._stack_ctx is None                                      # per-model context: the plain trainer has none
        assert torch.equal(plain, over)
        # buckets: contiguous from offset 0, last layers first; the tail slice reaches the end of the buffer
        edges = [0]
        for lo, hi, k in tr1._buckets:
            assert lo == edges[-1] and hi > lo
            edges.append(hi)
        assert tr1._tail_range == (edges[-1], tr1.fp.grad.numel())
        assert [k for _, _, k in tr1._buckets] == sorted((k for _, _, k in tr1._buckets), reverse=True)
        names = tr1.fp.names
        assert names[0].startswith('global_layer.3.') and names[-1].split('.')[0] not in ('global_layer', 'local_layer')
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_native_optimizer_kernel_matches_torch_adam():
    """csrc/optim.hip (clip + Adam + EMA + zero_grad in one pass) vs clip -> torch.optim.Adam -> EMA on the same flat
    buffers: 4 steps with clipping active and changing learning rates."""
    import models
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer
    dev = torch.device('cuda:0')
    cfg = models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    batches = [synth.qm9_batch(4, 8 * i, 8).to(dev) for i in range(4)]
    res = []
    for native in (True, False):
        torch.manual_seed(11)
        model = models.PAMNet(cfg).to(dev)
        tr = Trainer(model, lr=1e-3, max_grad_norm=2.0, weight_decay=0.0, native_optimizer=native, overlap_comm=False)
        assert tr.native_opt == native
        norms = []
        for i, b in enumerate(batches):
            tr.step(b, lr=1e-3 * (i + 1))
            norms.append(float(tr.last_grad_norm))
        assert float(tr.fp.grad.abs().sum()) == 0.0 or not native          # the kernel leaves the gradient zeroed
        res.append((tr.fp.flat.clone(), tr.shadow.clone(), norms))
    (p1, s1, n1), (p0, s0, n0) = res
    assert max(n1) > 2.0                                                   # the clip was active
    assert np.allclose(n1, n0, rtol=1e-5)
    # a few ulp: fma vs mul/add rounding (the largest parameters are the basis frequencies, up to 16*pi)
    assert torch.allclose(p1, p0, rtol=1e-6, atol=1e-6) and torch.allclose(s1, s0, rtol=1e-6, atol=1e-6)
    # weight decay variant on one step (L2 penalty added to the clipped gradient, as torch's Adam does)
    outs = []
    for native in (True, False):
        torch.manual_seed(12)
        model = models.PAMNet(cfg).to(dev)
        tr = Trainer(model, lr=1e-3, weight_decay=1e-2, native_optimizer=native, overlap_comm=False)
        tr.step(batches[0])
        outs.append(tr.fp.flat.clone())
    assert torch.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6)


def test_bench_world2_dry_run_on_cpu():
    """bench.py's world > 1 branch (shard_range per rank, barrier, max-over-ranks timing, one JSON line from rank 0, process
    group teardown) executed under torch.distributed.run with 2 gloo ranks on the CPU stand-in.  Not a measurement."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--batch-per-gpu', '8', '--n-layer', '3', '--cpu-dry-run']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line, from rank 0 only
    j = json.loads(lines[0])
    assert j['dry_run'] is True and j['n_gpus'] == 2 and j['steps'] == 3 and j['config']['global_batch'] == 16
    assert j['scaling'] == 'weak' and j['value'] > 0 and abs(j['value'] - 16 * 3 / (j['ms_per_step'] * 3e-3)) < 1e-6 * j['value']
    # the line says by itself who was in the job (round 6): the ranks the transport reached, one entry per rank, per-rank times
    c = j['comm']
    assert c['ranks'] == 2 and c['allreduce_of_ones'] == 2 and c['backend'] == 'gloo' and c['rccl_ranks'] == 0
    assert sorted(d['rank'] for d in c['devices']) == [0, 1] and len({d['pid'] for d in c['devices']}) == 2
    assert len(c['per_rank_ms_per_step']['all']) == 2 and c['per_rank_ms_per_step']['max'] <= j['ms_per_step'] * (1 + 1e-9)


def test_bench_multi_rank_reports_efficiency_and_diagnoses_a_missing_peer():
    """--n1-ms puts the weak-scaling efficiency into the line; a rendezvous that cannot complete (one rank of two launched)
    ends with a one-line diagnosis inside the init timeout, not a hang."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch-per-gpu', '8', '--n-layer', '3',
            '--cpu-dry-run']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port())] + base + ['--n1-ms', '5.0']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert abs(j['comm']['efficiency_vs_n1'] - 5.0 / j['ms_per_step']) < 1e-9 and j['comm']['n1_ms_per_step'] == 5.0
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable] + base + ['--init-timeout', '5'], capture_output=True, text=True, timeout=300, cwd=repo,
                       env=env)
    assert r.returncode != 0 and 'process group did not form within 5 s' in r.stderr and 'WORLD_SIZE=2' in r.stderr


def test_multistep_lr_helper_equals_torch_scheduler():
    """train.MultiStepLR == torch.optim.lr_scheduler.MultiStepLR as main_pdbbind.py:83,96 drives it (one step per epoch)."""
    from pamnet_amd.train import MultiStepLR
    ms = [50, 100, 150, 200, 250, 300, 350, 400, 450, 500]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=ms, gamma=0.2)
    mine = MultiStepLR(1e-3, ms, 0.2)
    for epoch in range(520):
        assert abs(opt.param_groups[0]['lr'] - mine.lr_at(epoch)) <= 1e-12 * mine.lr_at(epoch), epoch
        assert mine.lr_for_step(epoch, 3, 10) == mine.lr_at(epoch)
        opt.step()
        ref.step()


def test_trainer_loss_variants_on_cpu_standin():
    """Trainer(loss=..., max_grad_norm=None, ema_decay=None) on the CPU stand-in: the three losses' torch forms drive the step,
    no shadow is kept, evaluate() / predictions() run on the weights themselves; an unknown loss raises."""
    from standin import LayeredStandIn
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer
    F = torch.nn.functional
    for kind, fn in (('l1', F.l1_loss), ('mse', F.mse_loss), ('smooth_l1', F.smooth_l1_loss)):
        torch.manual_seed(0)
        m = LayeredStandIn(n_layer=2)
        ref = LayeredStandIn(n_layer=2)
        ref.load_state_dict(m.state_dict())
        tr = Trainer(m, lr=1e-2, loss=kind, max_grad_norm=None, ema_decay=None, native_optimizer=False)
        assert tr.shadow is None
        b = synth.qm9_batch(0, 0, 9)
        opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
        for _ in range(3):
            loss = tr.step(b)
            opt.zero_grad()
            l2 = fn(ref(b), b.y)
            l2.backward()
            opt.step()
            assert abs(float(loss) - float(l2)) < 1e-6
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            assert torch.allclose(p, q, atol=1e-6), k
        pred, y = tr.predictions([b])
        assert pred.shape == y.shape == (9,)
    with pytest.raises(ValueError):
        Trainer(LayeredStandIn(n_layer=1), loss='huber')


def test_bench_plain_invocation_launches_its_own_ranks_on_cpu():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the form the driver uses for N = 1): bench.py re-launches itself
    under torch.distributed.run on a free local port -- one JSON line, same contract.  And a world size that contradicts
    --gpus is an error message, not an assertion."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    cmd = [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--batch-per-gpu', '8', '--n-layer', '3', '--cpu-dry-run']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=repo, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout      # nothing but the line on stdout
    j = json.loads(lines[0])
    assert j['dry_run'] is True and j['n_gpus'] == 2 and j['steps'] == 3 and j['config']['global_batch'] == 16
    env['WORLD_SIZE'] = '3'
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=repo, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=3' in r.stderr and 'Traceback' not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['qm9_d128', 'rna_d16'])
def test_direct_tape_backward_equals_autograd(kind):
    """Trainer.forward_backward replays the recorded forward's backward directly (ops.backward_whole) instead of going
    through torch.autograd: the flat gradient is bit for bit what out.backward(d_out) leaves."""
    import models
    from pamnet_amd import ops, synth
    from pamnet_amd.train import Trainer
    dev = torch.device('cuda:0')
    if kind == 'qm9_d128':
        cfg = models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
        b = synth.qm9_batch(5, 0, 16).to(dev)
    else:
        cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
        b = synth.rna_batch(2, 0, 2, n_nodes=300).to(dev)
    torch.manual_seed(4)
    model = models.PAMNet(cfg).to(dev)
    tr = Trainer(model, lr=1e-3)
    used = []
    real = ops.backward_whole

    def spy(out, grad):
        used.append(real(out, grad))
        return used[-1]

    ops.backward_whole = spy
    try:
        tr.forward_backward(b)
        direct = tr.fp.grad.clone()
        assert used == [True]
        ops.backward_whole = lambda out, grad: False                      # the autograd route
        tr.forward_backward(b)
        via_autograd = tr.fp.grad.clone()
    finally:
        ops.backward_whole = real
    assert float(direct.abs().max()) > 0
    assert torch.equal(direct, via_autograd)


@pytest.mark.gpu
@pytest.mark.parametrize('n_layer,n_mol', [(1, 16), (2, 16), (3, 40), (4, 128), (2, 150), (3, 170)])
def test_engine_gradients_direct_vs_plain_autograd_across_launch_plans(n_layer, n_mol):
    """The layer-stack backward picks its launch plan from the batch: riders in the chain launches and ONE weight-gradient launch
    per layer pair when the chains leave CUs idle (ceil(n / 16) <= 176 tiles: up to ~140 QM9 molecules), per-layer launches
    with the tail jobs in them beyond; the last pair and n_layer = 1 have their own forms.  Whatever the plan, the gradients
    the Trainer's kernels write in place equal -- to fp32 summation order of the reductions -- those of a twin model under
    plain autograd, and both are repeatable bit for bit."""
    import models
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer
    dev = torch.device('cuda:0')
    cfg = models.Config(dataset='QM9', dim=128, n_layer=n_layer, cutoff_l=5.0, cutoff_g=5.0)
    b = synth.qm9_batch(9, 0, n_mol).to(dev)
    torch.manual_seed(n_layer * 100 + n_mol)
    model = models.PAMNet(cfg).to(dev)
    twin = models.PAMNet(cfg).to(dev)
    twin.load_state_dict(model.state_dict())
    tr = Trainer(model, lr=1e-3)
    tr.forward_backward(b)
    g1 = tr.fp.grad.clone()
    tr.forward_backward(b)
    assert torch.equal(g1, tr.fp.grad)                                       # repeatable
    torch.nn.functional.l1_loss(twin(b), b.y).backward()
    grads = dict(zip(tr.fp.names, tr.fp.grad_views))
    for k, p in twin.named_parameters():
        if p.grad is None:
            assert float(grads[k].abs().max()) == 0.0, k
            continue
        ref = p.grad
        den = float(ref.abs().max())
        if den == 0.0:
            assert float(grads[k].abs().max()) == 0.0, k
        else:
            assert float((grads[k] - ref).abs().max()) / den < 2e-5, (k, float((grads[k] - ref).abs().max()) / den)


def _pamnet_rank(rank, world, port, out, total, shared_gpu=False, steps=3, stride=40):
    """One rank of the 2-rank PAMNet step: its molecule shard of the global batch, on its own device over RCCL -- or, with
    shared_gpu, on the box's one device with gloo carrying the (device-resident) gradient slices between the processes."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p_ in (os.path.dirname(here), os.path.join(os.path.dirname(here), 'physics-aware-multiplex-gnn_amd')):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0 if shared_gpu else rank)
    dev = torch.device('cuda', 0 if shared_gpu else rank)
    if shared_gpu:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    import models
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer, shard_range
    torch.manual_seed(100 + rank)                 # different initial weights per rank: the trainer broadcasts rank 0's
    model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)).to(dev)
    if rank == 0:
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, out + '.init')
    tr = Trainer(model, lr=1e-3, world_size=world, n_buckets=3)
    assert tr._buckets is not None and tr._stack_ctx is not None          # the bucketed, overlapped all-reduce
    lo, hi = shard_range(total, rank, world)
    losses = []
    for step in range(steps):
        losses.append(float(tr.step(synth.qm9_batch(7, stride * step + lo, hi - lo).to(dev), global_graphs=total)))
    torch.cuda.synchronize()
    ref = tr.fp.flat.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, tr.fp.flat)           # identical parameters on every rank
    if rank == 0:
        torch.save({'flat': tr.fp.flat.cpu(), 'shadow': tr.shadow.cpu(), 'norm': float(tr.last_grad_norm)}, out)
    dist.destroy_process_group()


def _check_two_rank_result(out, total, steps=3, stride=40):
    import models
    from pamnet_amd import synth
    from pamnet_amd.train import Trainer
    got, init = torch.load(out), torch.load(out + '.init')
    dev = torch.device('cuda:0')
    model = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0))
    model.load_state_dict(init, strict=True)
    tr = Trainer(model.to(dev), lr=1e-3, world_size=1)
    for step in range(steps):
        tr.step(synth.qm9_batch(7, stride * step, total).to(dev))
    torch.cuda.synchronize()
    # same arithmetic up to the summation order of the ranks' partial gradients (fp32): parameters after the Adam steps
    d = (got['flat'] - tr.fp.flat.cpu()).abs().max() / tr.fp.flat.abs().max().cpu()
    assert float(d) < 1e-5, float(d)
    assert float((got['shadow'] - tr.shadow.cpu()).abs().max()) < 1e-5
    assert abs(got['norm'] / float(tr.last_grad_norm) - 1) < 1e-4


@pytest.mark.gpu
def test_pamnet_two_ranks_on_one_gpu_step_equals_global_batch_step(tmp_path):
    """The molecule-sharded step of the REAL model with world_size = 2 on a 1-GPU box: two processes share the device, the
    process group is gloo (it carries device tensors through the host) -- everything but the transport is what runs on 8
    GPUs: rank 0's parameters broadcast, shards of 13 + 12 molecules, gradients pre-scaled by local / global graphs and
    written in place by the kernels, the flat buffer tiled into buckets by layer pair, the engine's per-pair events gating
    each bucket's all-reduce on the communication stream, the identical fused update on both ranks.  Result = the
    single-process step on the 25-molecule batch."""
    total, out = 25, str(tmp_path / 'dp2_shared.pt')
    mp.spawn(_pamnet_rank, args=(2, _free_port(), out, total, True), nprocs=2, join=True)
    _check_two_rank_result(out, total)


@pytest.mark.gpu
def test_pamnet_eight_ranks_on_one_gpu_step_equals_the_1024_molecule_step(tmp_path):
    """BASELINE configs[2]'s split as EIGHT processes (main_qm9.py:60-67 at batch_size 1 024 -> 128 molecules per rank, d = 128;
    n_layer = 2 for time), all on this box's one device with gloo as the transport: rank 0's parameters broadcast to seven
    ranks seeded differently, eight shards, gradients pre-scaled by 128 / 1 024, three buckets gated by the engine's per-pair
    events, the identical fused update everywhere -- equal to the single-process step on the 1 024-molecule batch.  A flow
    check of the shard arithmetic and the bucket events at N = 8 (only ever tried at N = 2 before), not a scaling number."""
    total, out = 1024, str(tmp_path / 'dp8_shared.pt')
    mp.spawn(_pamnet_rank, args=(8, _free_port(), out, total, True, 2, 1200), nprocs=8, join=True)
    _check_two_rank_result(out, total, steps=2, stride=1200)


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu_prints_one_line():
    """`python bench.py --gpus 8 --share-gpu --steps 3`: the driver's 8-GPU invocation of the real bench with all ranks on one
    device (self-launch of 8 ranks, one JSON line from rank 0, max-over-ranks timing, global batch = 8 x per-GPU batch)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    cmd = [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '8', '--share-gpu', '--steps', '3', '--warmup', '1',
           '--batch-per-gpu', '16', '--n-layer', '2', '--no-rooflines', '--no-cpu-baseline', '--no-other-configs']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=repo, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 8 and j['config']['global_batch'] == 128 and j['scaling'] == 'weak' and 'shared_gpu' in j
    assert j['comm'] and j['comm']['buckets'] >= 2
    assert abs(j['value'] - 128 * 3 / (j['ms_per_step'] * 3e-3)) < 1e-6 * j['value']
    # round 6: the line answers "who was in the job" by itself
    c = j['comm']
    assert c['ranks'] == 8 and c['allreduce_of_ones'] == 8 and sorted(d['rank'] for d in c['devices']) == list(range(8))
    assert len({d['pid'] for d in c['devices']}) == 8 and all(d['pci'] == c['devices'][0]['pci'] for d in c['devices'])
    assert c['backend'] == 'gloo' and c['rccl_ranks'] == 0          # (shared device: gloo transport, said so)
    assert len(c['per_rank_ms_per_step']['all']) == 8 and 'allreduce_exposed_ms' in c


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_runs_the_multi_rank_flow():
    """`python bench.py --gpus 2 --share-gpu`: the driver's N > 1 invocation of the REAL bench (self-launch, one JSON line from
    rank 0, barrier + max-over-ranks timing, the bucketed overlapped gradient exchange and the exposed-communication probe),
    both ranks on this box's one device with gloo as the transport.  Not a measurement (the line says so)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    cmd = [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--share-gpu', '--steps', '6', '--warmup', '2',
           '--batch-per-gpu', '32', '--n-layer', '2', '--no-rooflines', '--no-cpu-baseline', '--no-other-configs']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=repo, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 64 and j['scaling'] == 'weak' and 'shared_gpu' in j
    assert j['comm'] and j['comm']['buckets'] >= 2 and j['comm']['gradient_bytes'] > 0
    assert abs(j['value'] - 64 * 6 / (j['ms_per_step'] * 6e-3)) < 1e-6 * j['value']


@pytest.mark.gpu
def test_pamnet_two_rank_rccl_step_equals_global_batch_step(tmp_path):
    """PAMNet (d=128, L=2) on TWO GPUs over RCCL: shard by molecule (13 + 12), pre-scale by local/global graphs,
    bucketed all-reduce overlapped with the backward, identical update -- equals the single-process step on the global
    batch (what test_dp_matches_single_process shows for the CPU stand-in).  Needs two devices: skipped on a 1-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (this box has %d)' % torch.cuda.device_count())
    total, out = 25, str(tmp_path / 'dp2.pt')
    mp.spawn(_pamnet_rank, args=(2, _free_port(), out, total), nprocs=2, join=True)
    _check_two_rank_result(out, total)

"""Training step with the semantics of the reference loops (main_qm9.py:99-118 + utils/ema.py:3-32; main_pdbbind.py:88-95;
main_rna_puzzles.py:86-93) and molecule-sharded data parallelism (one process per GPU, RCCL over xGMI).

Per step:  zero_grad -> forward -> loss (mean over graphs) -> backward -> [all-reduce of the flat fp32 gradient]
           -> [clip_grad_norm_(L2)] -> Adam(lr, wd, amsgrad=False) -> [EMA].
  main_qm9.py         : Trainer(model, loss='l1', max_grad_norm=1000, ema_decay=0.999) + WarmupExpLR     (the defaults)
  main_pdbbind.py     : Trainer(model, loss='mse', max_grad_norm=None, ema_decay=None) + MultiStepLR(gamma=0.2)
  main_rna_puzzles.py : Trainer(model, loss='smooth_l1', max_grad_norm=None, ema_decay=None), constant rate

MI355X-first layout: all parameters, their gradients, both Adam moments and the EMA shadow live in five flat fp32
buffers (3.58 M floats = 14.3 MB each at d=128/L=6).  Parameters / .grad are views into them, so
  * the gradient all-reduce is ONE RCCL call on one contiguous buffer (ring all-reduce moves 2(n-1)/n * 14.3 MB per GPU,
    per-xGMI-link bound; no bucketing copies),
  * clip / Adam / EMA are a fixed handful of launches over one tensor instead of ~300 per-parameter launches.
Each rank pre-scales its gradient by local_graphs/global_graphs so the summed gradient equals that of the global-batch
mean L1 loss (main_qm9.py:108), then every rank applies the identical update.
"""
import math
import re

import torch
import torch.distributed as dist
import torch.nn.functional as F


_LAYER_RE = re.compile(r'^(?:global_layer|local_layer)\.(\d+)\.')


def _layer_of(name):
    m = _LAYER_RE.match(name)
    return int(m.group(1)) if m else None


class FlatParams(object):
    """Re-home a module's parameters (and their .grad) as views of two flat buffers.

    Buffer order = the order in which the backward finishes gradients: layer pair n_layer-1 first (global_layer.k and
    local_layer.k adjacent), ..., layer pair 0, then the top-level parameters (embeddings, basis frequencies, input
    embeddings).  `layer_ranges[k]` = [lo, hi) of layer pair k, so the gradients of any run of consecutive layers are
    one contiguous slice that can be all-reduced while earlier layers are still being differentiated."""

    def __init__(self, module, direct=True):
        # (a models.PAMNet may present ONE flat parameter to its caller -- PAMNET_FLAT_PARAMS=1 --: this takes the real ones)
        named = [(n, p) for n, p in getattr(module, '_real_named_parameters', module.named_parameters)() if p.requires_grad]
        rank = lambda n: (1, 0) if _layer_of(n) is None else (0, -_layer_of(n))
        named.sort(key=lambda np_: rank(np_[0]))          # stable: declaration order inside a group
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        dev, dt = self.params[0].device, self.params[0].dtype
        align = 64                                        # floats: every parameter starts on a 256-byte boundary
        offs, off = [], 0                                 # (the kernels read weights as 16-byte vectors)
        for p in self.params:
            offs.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = sum(p.numel() for p in self.params)
        self.offsets = dict(zip(self.names, offs))
        self.layer_ranges = {}
        for n, o, p in zip(self.names, offs, self.params):
            k = _layer_of(n)
            if k is not None:
                end = o + (p.numel() + align - 1) // align * align
                lo, hi = self.layer_ranges.get(k, (o, end))
                self.layer_ranges[k] = (min(lo, o), max(hi, end))
        self.flat = torch.zeros(off, device=dev, dtype=dt)          # padding stays zero: no effect on norms / Adam
        self.grad = torch.zeros(off, device=dev, dtype=dt)
        self.grad_views = []
        for p, o in zip(self.params, offs):
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view_as(p)
            self.grad_views.append(self.grad[o:o + k].view_as(p))
            p.grad = self.grad_views[-1]
        self.set_direct(direct)

    def set_direct(self, on):
        """Allow (or forbid) the fused kernels to write these parameters' gradients in place (overwrite semantics: one
        backward per zero_grad).  The permission travels with the parameters of THIS module -- nothing process-global."""
        self.direct = bool(on)
        for p in self.params:
            p._pamnet_direct = self.direct

    def zero_grad(self):
        self.grad.zero_()

    # Per-operator autograd paths (widths other than 128) would add every parameter's gradient into its view with one
    # small kernel each (~150 launches per step); instead autograd hands the gradient tensors over (p.grad = None
    # before the backward) and one multi-tensor add packs them into the flat buffer.
    def release_grads(self):
        for p in self.params:
            p.grad = None

    def pack_grads(self):
        views, grads = [], []
        for p, v in zip(self.params, self.grad_views):
            if p.grad is not None:
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            torch._foreach_add_(views, grads)


class WarmupExpLR(object):
    """lr(t) = lr0 * t for fractional epoch t <= 1, then lr0 * gamma**(t-1): GradualWarmupScheduler(multiplier=1,
    total_epoch=1, after=ExponentialLR(gamma)) stepped per iteration with the fractional epoch (main_qm9.py:93-94,
    113-114).  The warmup_scheduler package is not in the reference tree: semantics restated, parity unpinned."""

    def __init__(self, lr0, gamma=0.9961697, steps_per_epoch=1.0):
        self.lr0, self.gamma, self.spe = lr0, gamma, float(steps_per_epoch)

    def lr_at(self, epoch, step):
        t = epoch + step / self.spe
        return self.lr0 * t if t <= 1.0 else self.lr0 * self.gamma ** (t - 1.0)

    def lr_for_step(self, epoch, step, steps_in_epoch):
        """Learning rate the optimiser uses AT (epoch, step) in the reference loop: the scheduler is stepped after
        optimizer.step() with the fractional epoch of the step just done (main_qm9.py:112-114), and the warm-up scheduler
        starts from 0 -- so step s runs with lr_at of step s-1, the very first step with 0."""
        if step > 0:
            return self.lr_at(epoch, step - 1)
        if epoch == 0:
            return 0.0
        return self.lr_at(epoch - 1, steps_in_epoch - 1)


class MultiStepLR(object):
    """lr(epoch) = lr0 * gamma ** #{milestones <= epoch}: torch.optim.lr_scheduler.MultiStepLR as main_pdbbind.py:83,96
    uses it (stepped once per epoch, after the epoch's optimiser steps; every step of epoch e runs with lr_at(e))."""

    def __init__(self, lr0, milestones=(50, 100, 150, 200, 250, 300, 350, 400, 450, 500), gamma=0.2):
        self.lr0, self.milestones, self.gamma = float(lr0), sorted(int(m) for m in milestones), float(gamma)

    def lr_at(self, epoch):
        return self.lr0 * self.gamma ** sum(1 for m in self.milestones if m <= epoch)

    def lr_for_step(self, epoch, step=0, steps_in_epoch=1):
        return self.lr_at(epoch)


def plan_buckets(layer_ranges, numel, n_buckets):
    """Tile the flat gradient [0, numel) into contiguous all-reduce slices by layer pair, in the order the backward
    completes them (FlatParams lays layer pair L-1 first).  Returns (buckets, tail): buckets = [(lo, hi, k_ready)] --
    slice [lo, hi) is final once layer pair k_ready is done -- and tail = (lo, numel): the first layers together with
    the top-level parameters, final only when the whole backward is.  None when the layers are not 0..L-1."""
    L = len(layer_ranges)
    if L == 0 or sorted(layer_ranges) != list(range(L)):
        return None
    per = max(1, math.ceil(L / float(n_buckets)))
    buckets, k_hi = [], L - 1
    while k_hi >= 0:
        k_lo = max(0, k_hi - per + 1)
        buckets.append((layer_ranges[k_hi][0], layer_ranges[k_lo][1], k_lo))
        k_hi = k_lo - 1
    lo_last = buckets.pop()[0]               # merged with the top-level parameters
    tail = (lo_last, numel)
    pos = 0
    for lo, hi, _ in buckets:                # contiguous, in order, no overlap
        assert lo == pos and hi > lo
        pos = hi
    assert pos == tail[0] <= numel
    return buckets, tail


_SIDE = {}


def _side_stream(device):
    """ONE input-pipeline stream per device for the whole process.  Streams are handed out of a pool and mapped onto a few
    hardware queues round-robin: a fresh stream per Prefetcher made the queue a side stream lands on depend on how many
    trainers the process had created before -- and a side stream that shares the main stream's queue builds its graphs in line
    (PAMNet_s as the third configuration of bench.other_configs: 2.38 ms/step against 2.20 in a fresh process)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class Prefetcher(object):
    """Builds the graph of the NEXT batch (model.prepare: graph construction + spherical basis) on a side stream, so that
    its kernels and its one or two host round trips (data-dependent sizes) overlap the current step instead of draining
    the main queue at the start of the next one.  The batch is handed back through `data._pamnet_ready` (an event on the
    side stream); `wait(data)` makes the current stream wait for it.
    PAMNET_PREFETCH_THREAD=1 moves the construction to a worker thread (the round trips then no longer stop the thread
    that enqueues the step).  Measured and left off: graph construction is ~100 small torch / ctypes calls, the two
    threads fight over the GIL, and every configuration got slower (RNA d = 16 training 1.75 -> 1.99 ms/step, forward
    0.90 -> 1.06 ms; QM9 unchanged)."""

    def __init__(self, model, device):
        import os
        self.model, self.device = model, device
        self.side = _side_stream(device)
        self.pool = None
        if os.environ.get('PAMNET_PREFETCH_THREAD', '0') != '0':
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='pamnet-prefetch')

    def _build(self, data, need_grad, main):
        torch.cuda.set_device(self.device)
        _wait_inputs(self.side, data)
        with torch.cuda.stream(self.side):
            self.model.prepare(data, need_grad=need_grad)
            ev = torch.cuda.Event()
            ev.record(self.side)
        data._pamnet_ready = ev
        # the tensors were allocated on the side stream but will be consumed on the main stream
        for v in _graph_tensors(data._pamnet_prepared):
            v.record_stream(main)

    def submit(self, data, need_grad=True):
        """The batch tensors must be complete when this is called -- OR carry `data.inputs_ready`, a torch.cuda.Event
        recorded on the stream that produces them (pinned .to(device, non_blocking=True), GPU-side collation): the side
        stream then waits for exactly that event (synth.Batch.to(..., non_blocking=True) records it).  The side stream
        does not wait for the main stream (that would serialise it behind the whole step)."""
        main = torch.cuda.current_stream(self.device)
        if self.pool is None:
            self._build(data, need_grad, main)
        else:
            data._pamnet_future = self.pool.submit(self._build, data, need_grad, main)

    def wait(self, data):
        fut = getattr(data, '_pamnet_future', None)
        if fut is not None:
            data._pamnet_future = None
            fut.result()                                   # re-raises what the worker raised (bad inputs, ...)
        ev = getattr(data, '_pamnet_ready', None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            data._pamnet_ready = None


class Trainer(object):
    def __init__(self, model, lr=1e-4, weight_decay=0.0, ema_decay=0.999, max_grad_norm=1000.0, betas=(0.9, 0.999),
                 eps=1e-8, world_size=1, process_group=None, overlap_comm=True, n_buckets=3, native_optimizer=True,
                 loss='l1'):
        """loss: 'l1' | 'mse' | 'smooth_l1' (the three drivers' losses, mean over graphs); max_grad_norm=None: no
        clip_grad_norm_; ema_decay=None: no EMA shadow (evaluate() then runs on the weights themselves)."""
        from .ops import LOSS_ENTRIES
        if loss not in LOSS_ENTRIES:
            raise ValueError("loss must be one of 'l1', 'mse', 'smooth_l1' (got %r)" % (loss,))
        self.loss_kind = loss
        self.model = model
        if hasattr(model, '_disable_flat_view'):
            model._disable_flat_view()                     # (this class owns the flat buffers; no second interface on top)
        self.fp = FlatParams(model, direct=True)           # fused layers write gradients straight into fp.grad
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # On the GPU clip + Adam + EMA (+ the next zero_grad) are one pass of csrc/optim.hip over the flat buffers;
        # torch.optim.Adam on the flat tensor is the CPU path (gloo unit tests) and the cross-check of the kernel.
        self.native_opt = bool(native_optimizer and self.fp.flat.is_cuda)
        if self.native_opt:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.fp.flat), torch.zeros_like(self.fp.flat)
            self.step_count = 0
        else:
            self.opt = torch.optim.Adam([torch.nn.Parameter(self.fp.flat)], lr=lr, betas=betas, eps=eps,
                                        weight_decay=weight_decay, amsgrad=False, fused=self.fp.flat.is_cuda)
            self._p = self.opt.param_groups[0]['params'][0]
            self._p.grad = self.fp.grad
        self._grad_clean = True                            # fp.grad is all zeros (fresh buffer / zeroed by the update)
        # widths without a one-node forward (models._one_node: dim = 128 engine, narrow-width tape): autograd hands the
        # gradient tensors over and one multi-tensor add packs them into the flat buffer
        self._pack = bool(self.fp.flat.is_cuda and getattr(model, 'dim', 128) not in (128, 16, 32, 64))
        self.ema_decay = ema_decay
        self.shadow = self.fp.flat.clone() if ema_decay is not None else None      # utils/ema.py:9-11
        self.max_grad_norm = max_grad_norm
        self.world_size, self.pg = world_size, process_group
        self._buckets = self._stack_ctx = None
        self._force_single = overlap_comm == 'force_single'     # measurement aid: one all-reduce even on a 1-rank group
        distributed = dist.is_available() and dist.is_initialized()
        if world_size > 1 and distributed:
            # identical initial parameters on every rank whatever the seeds were (main_qm9.py has a single process)
            dist.broadcast(self.fp.flat, 0, group=process_group)
            if self.shadow is not None:
                self.shadow.copy_(self.fp.flat)
        if overlap_comm and overlap_comm != 'force_single' and distributed and (world_size > 1 or overlap_comm == 'force'):
            self._setup_buckets(n_buckets)

    # -- pieces (also timed individually by bench.py) ---------------------------------------------------------------
    def forward_backward(self, data, global_graphs=None):
        """One forward + backward into the flat gradient.  Overwrite semantics (the fused kernels write gradients in
        place): a call always starts from a zeroed buffer, there is no gradient accumulation across calls."""
        if not self._grad_clean:
            self.fp.zero_grad()
        self._grad_clean = False
        if self._stack_ctx is not None:
            self._stack_ctx.recorded = False
        if self._pack:
            self.fp.release_grads()
        out = self.model(data)
        # local mean -> contribution to the global-batch mean
        if self.world_size > 1 and global_graphs is None:
            global_graphs = out.numel() * self.world_size            # equal shards unless the caller says otherwise
        scale = float(out.numel()) / float(global_graphs) if self.world_size > 1 else 1.0
        if out.is_cuda and out.dtype == torch.float32:
            from . import ops
            loss, d_out = ops.loss_with_grad(self.loss_kind, out, data.y, scale)      # loss + its gradient: one launch
            if not ops.backward_whole(out, d_out):                       # (a forward that is not one recorded node)
                out.backward(d_out)
        else:                                                            # gloo / CPU unit tests with a plain module
            loss = {'l1': F.l1_loss, 'mse': F.mse_loss, 'smooth_l1': F.smooth_l1_loss}[self.loss_kind](out, data.y)
            (loss * scale).backward()
        if self._pack:
            self.fp.pack_grads()
        return loss

    # -- gradient all-reduce -------------------------------------------------------------------------------------------
    def _setup_buckets(self, n_buckets):
        """Split the flat gradient into contiguous slices by layer pair, last layers first (plan_buckets).  On the GPU
        the layer-stack backward records one event per layer pair (handed to it through the model's fused.StackCtx) and
        the slices are reduced on a side stream behind those events; on CPU (gloo tests) the same slices are reduced
        one after the other."""
        plan = plan_buckets(self.fp.layer_ranges, self.fp.grad.numel(), n_buckets)
        if plan is None:
            return
        self._buckets, self._tail_range = plan
        if not self.fp.flat.is_cuda:
            return
        from . import fused
        dev = self.fp.flat.device
        L = len(self.fp.layer_ranges)
        events = [torch.cuda.Event() for _ in range(L)]
        for e in events:
            e.record(torch.cuda.current_stream(dev))      # materialise the handles the C side records into
        self._events = events
        self._comm = torch.cuda.Stream(device=dev)
        layers = getattr(self.model, 'global_layer', None)
        if layers is not None:
            self._stack_ctx = fused.stack_ctx(layers)
            self._stack_ctx.events = events

    def sync_gradients(self):
        """Sum the gradient over ranks.  With buckets: slices of the last layers are reduced on a side stream as soon as
        the backward has produced them (the host has already enqueued the whole backward when this runs; the device is
        still working through it), the remainder after the backward's end; the main stream then waits for all of it."""
        if self._buckets is None:
            if self.world_size > 1 or self._force_single:
                dist.all_reduce(self.fp.grad, op=dist.ReduceOp.SUM, group=self.pg)
            return
        if not self.fp.flat.is_cuda:                       # same tiling, no streams
            for lo, hi, _ in self._buckets:
                dist.all_reduce(self.fp.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
            lo, hi = self._tail_range
            dist.all_reduce(self.fp.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
            return
        if self._stack_ctx is None or not self._stack_ctx.recorded:
            # this backward did not go through the engine (no per-layer events): one all-reduce
            dist.all_reduce(self.fp.grad, op=dist.ReduceOp.SUM, group=self.pg)
            return
        main = torch.cuda.current_stream(self.fp.flat.device)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(self._comm):
            for lo, hi, k in self._buckets:
                self._comm.wait_event(self._events[k])
                dist.all_reduce(self.fp.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
            self._comm.wait_event(done)
            lo, hi = self._tail_range
            dist.all_reduce(self.fp.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
        main.wait_stream(self._comm)

    def clip(self):
        norm = torch.linalg.vector_norm(self.fp.grad)
        if self.max_grad_norm is None:
            return norm
        coef = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0)    # clip_grad_norm_ (main_qm9.py:111)
        self.fp.grad.mul_(coef)
        return norm

    def optimizer_step(self, lr=None):
        if lr is not None:
            self.opt.param_groups[0]['lr'] = lr
        self.opt.step()

    def native_update(self, lr=None, num_updates=99999):
        """[clip ->] Adam [-> EMA] -> zero the gradient, one kernel; the norm stays on the device."""
        from . import lib
        if lr is not None:
            self.lr = lr
        from . import ops
        # the pre-clip norm is reported either way (last_grad_norm); without a clip its partials only feed norm_out
        part = ops.sumsq_partials(self.fp.grad)            # the kernel below adds the 256 partials and clips by the norm
        norm = torch.empty(1, dtype=torch.float32, device=self.fp.flat.device)
        self.step_count += 1
        decay = 0.0 if self.ema_decay is None else min(self.ema_decay, (1.0 + num_updates) / (10.0 + num_updates))   # utils/ema.py:14
        max_norm = float('inf') if self.max_grad_norm is None else float(self.max_grad_norm)
        lib.call('pamnet_adam_ema_norm_f32', lib.ptr(self.fp.flat), lib.ptr(self.fp.grad), lib.ptr(self.exp_avg),
                 lib.ptr(self.exp_avg_sq), lib.ptr(self.shadow), self.fp.flat.numel(), float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                 self.step_count, float(decay), lib.ptr(part), lib.ptr(norm), max_norm, 1,
                 lib.stream_of(self.fp.flat))
        norm = norm[0]
        self._grad_clean = True
        return norm

    def ema_update(self, num_updates=99999):
        if self.ema_decay is None:
            return
        decay = min(self.ema_decay, (1.0 + num_updates) / (10.0 + num_updates))     # utils/ema.py:14
        self.shadow.mul_(decay).add_(self.fp.flat, alpha=1.0 - decay)

    def step(self, data, lr=None, global_graphs=None, next_data=None):
        """One training step.  `next_data`: the batch of the following step; its graph is built on a side stream
        while this step's kernels run (input pipelining -- graph construction needs host round trips for the
        data-dependent sizes, which would otherwise drain the GPU queue at the start of every step)."""
        self._wait_prepared(data)
        self._throttle()
        loss = self.forward_backward(data, global_graphs)
        self.sync_gradients()
        if self.native_opt:
            self.last_grad_norm = self.native_update(lr)       # pre-clip L2 norm (device scalar; no host sync here)
        else:
            self.last_grad_norm = self.clip()
            self.optimizer_step(lr)
            self.ema_update()
        if next_data is not None:
            self.prefetch(next_data)
        return loss

    MAX_STEPS_IN_FLIGHT = 2

    def _throttle(self):
        """Keep the host at most MAX_STEPS_IN_FLIGHT steps ahead of the device.  Where a step's kernels take longer than
        its enqueueing (RNA at d = 64: 7 ms against 3.5), an unbounded lead means every prefetched graph and every
        saved-activation arena of the queued steps is alive at once: the caching allocator kept creating device segments
        (36-45 hipMalloc per 30 steps, 10 GB reserved, the step 1 ms slower than without the input pipeline)."""
        if not self.fp.flat.is_cuda:
            return
        q = self.__dict__.setdefault('_inflight', [])
        if len(q) >= self.MAX_STEPS_IN_FLIGHT:
            self._retire(q.pop(0))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.fp.flat.device))
        # The flag words of every forward enqueued so far (zero-host-sync batches) travel WITH the event recorded behind
        # them: whoever waits for this event may read exactly these words, whatever model.verify() was called in between
        # (an evaluation pass has its own forwards and its own verify()).
        take = getattr(self.model, 'take_pending_checks', None)
        q.append((ev, take() if take is not None else []))

    @staticmethod
    def _retire(entry):
        ev, flags = entry
        ev.synchronize()
        if flags:                                              # written by forwards that have completed by now:
            from . import graph as G                           # reading them does not stall the queue
            G.raise_for_flag(G.read_flags(flags, completed=True))

    def drain(self):
        """Wait for every step in flight and check its batches' device-side flag words (zero-host-sync batches)."""
        q = self.__dict__.get('_inflight') or []
        while q:
            self._retire(q.pop(0))
        if hasattr(self.model, 'verify'):
            self.model.verify()

    # -- input pipelining -----------------------------------------------------------------------------------------------
    def prefetch(self, data):
        """Build `data`'s graph on the side stream, from the prefetch worker (see Prefetcher); forward() picks it up."""
        if not hasattr(self.model, 'prepare') or not self.fp.flat.is_cuda:
            return
        if getattr(self, '_prefetcher', None) is None:
            self._prefetcher = Prefetcher(self.model, self.fp.flat.device)
        self._prefetcher.submit(data)

    def _wait_prepared(self, data):
        pf = getattr(self, '_prefetcher', None)
        if pf is not None:
            pf.wait(data)

    # -- evaluation under the EMA weights (main_qm9.py:29-37) ---------------------------------------------------------
    def ema_assign(self):
        if self.shadow is None:
            return
        self._saved = self.fp.flat.clone()
        self.fp.flat.copy_(self.shadow)

    def ema_resume(self):
        if self.shadow is None:
            return
        self.fp.flat.copy_(self._saved)
        del self._saved

    @torch.no_grad()
    def evaluate(self, batches):
        """Sum |out - y| over batches / #graphs under EMA weights, all-reduced across ranks."""
        self.drain()                       # the training steps in flight keep their own flag words (see _throttle)
        self.ema_assign()
        try:
            tot = torch.zeros(2, device=self.fp.flat.device, dtype=torch.float64)
            for data, out in predict(self.model, batches):
                tot[0] += (out - data.y).abs().sum().double()
                tot[1] += out.numel()
        finally:
            self.ema_resume()                  # (a failing batch must not leave the shadow in place of the weights)
        if self.world_size > 1:
            dist.all_reduce(tot, group=self.pg)
        res = float(tot[0] / tot[1])
        if hasattr(self.model, 'verify'):
            self.model.verify()
        return res

    @torch.no_grad()
    def predictions(self, batches):
        """(pred, y) as host arrays over `batches` -- what test() of main_pdbbind.py:25-39 / main_rna_puzzles.py:26-46
        collects before handing them to utils.rmse / mae / sd / pearson or F.smooth_l1_loss; under the EMA weights when
        the trainer keeps a shadow.  This rank's batches only."""
        self.drain()
        self.ema_assign()
        preds, ys = [], []
        try:
            for data, out in predict(self.model, batches):
                preds.append(out.reshape(-1)), ys.append(data.y.reshape(-1).to(out.device))
        finally:
            self.ema_resume()
        if not preds:
            import numpy as np
            return np.zeros(0, np.float32), np.zeros(0, np.float32)
        return torch.cat(preds).cpu().numpy(), torch.cat(ys).cpu().numpy()

    # -- checkpoints ----------------------------------------------------------------------------------------------------
    def state_dict(self):
        """Everything a resume needs (model weights under the reference's keys, both Adam moments, the EMA shadow, the update
        count).  Drains the steps in flight first: the deferred device-side checks of the last MAX_STEPS_IN_FLIGHT batches
        (zero-host-sync batches) are read before anything is saved -- a checkpoint never outlives an unchecked step."""
        self.drain()
        sd = {'model': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
              'shadow': None if self.shadow is None else self.shadow.clone(), 'lr': self.lr}
        if self.native_opt:
            sd.update(exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), step_count=self.step_count)
        else:
            import copy
            sd['optimizer'] = copy.deepcopy(self.opt.state_dict())    # (a snapshot like the rest: torch hands out its live
            #                                                             tensors, and a load from them would SHARE the step counter)
        return sd

    def load_state_dict(self, sd):
        """Restore what state_dict() saved.  The optimiser mode (the fused HIP Adam over the flat buffers, or torch's) and the
        flat layout must be those of the trainer that saved it: anything else raises ValueError instead of a bare KeyError or a
        silent mis-copy.  A checkpoint without an EMA shadow (a trainer that kept none) restarts the shadow from the loaded
        weights -- evaluate() / predictions() never run under the constructor's weights."""
        self.drain()
        saved_native = 'exp_avg' in sd
        if saved_native != bool(self.native_opt) or (not saved_native and 'optimizer' not in sd):
            raise ValueError('checkpoint holds %s optimiser state, this trainer runs %s' % (
                'the fused (native)' if saved_native else "torch's" if 'optimizer' in sd else 'no',
                'the fused (native) Adam' if self.native_opt else 'torch.optim.Adam'))
        if self.native_opt:
            for k in ('exp_avg', 'exp_avg_sq'):
                if tuple(sd[k].shape) != tuple(self.exp_avg.shape):
                    raise ValueError('checkpoint %s has %d elements, this model\'s flat layout %d' % (
                        k, sd[k].numel(), self.exp_avg.numel()))
        if sd.get('shadow') is not None and self.shadow is not None and tuple(sd['shadow'].shape) != tuple(self.shadow.shape):
            raise ValueError('checkpoint EMA shadow has %d elements, this model\'s flat layout %d' % (
                sd['shadow'].numel(), self.shadow.numel()))
        self.model.load_state_dict(sd['model'], strict=True)       # (parameters are views of fp.flat: copied in place)
        if self.shadow is not None:
            if sd.get('shadow') is not None:
                self.shadow.copy_(sd['shadow'])
            else:
                self.shadow.copy_(self.fp.flat)
        self.lr = sd.get('lr', self.lr)
        if self.native_opt:
            self.exp_avg.copy_(sd['exp_avg']), self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            self.step_count = int(sd['step_count'])
        else:
            self.opt.load_state_dict(sd['optimizer'])

    def close(self):
        """End of training: wait for the steps in flight and read their deferred checks (raises what they found)."""
        self.drain()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        return False


@torch.no_grad()
def predict(model, batches):
    """Forward-only loop (test() of main_qm9.py:29-37, inference_rna_puzzles.py:60-67) with the input pipeline of
    Trainer.step: while batch i runs, the graph of batch i+1 is built on a side stream (forward-only variant: no
    transposed index lists).  Yields (data, output)."""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    dev = next(model.parameters()).device
    pf = Prefetcher(model, dev) if (dev.type == 'cuda' and hasattr(model, 'prepare')) else None
    while cur is not None:
        nxt = next(it, None)
        if pf is not None:
            pf.wait(cur)
            if nxt is not None:
                # queued BEFORE this batch's forward: the graph kernels are small, and their host round trip resolves
                # (on the worker) while this forward is being enqueued and run
                pf.submit(nxt, need_grad=False)
        out = model(cur)
        yield cur, out
        cur = nxt
    if pf is not None and pf.pool is not None:
        pf.pool.shutdown(wait=True)
    if hasattr(model, 'verify'):
        model.verify()            # batches that carried their sizes (store.MoleculeStore) ran without a host round trip


def _wait_inputs(stream, data):
    """Make `stream` wait for the producer of a batch's tensors when the batch says who that is (data.inputs_ready)."""
    ev = getattr(data, 'inputs_ready', None)
    if ev is not None:
        stream.wait_event(ev)
        for k in ('x', 'pos', 'edge_index', 'batch'):       # allocated on the producer's stream, read on this one
            t = getattr(data, k, None)
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(stream)


def _graph_tensors(g):
    """Every tensor hanging off a prepared graph (one level of nested index structures)."""
    out = []
    for v in vars(g).values():
        if isinstance(v, torch.Tensor):
            out.append(v)
        elif hasattr(v, '__slots__'):
            out += [getattr(v, k) for k in v.__slots__ if isinstance(getattr(v, k, None), torch.Tensor)]
    return out


def shard_range(total, rank, world):
    """Contiguous molecule shard [lo, hi) of a global batch (graphs are independent units: no halo)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shards(sizes, world):
    """Node-balanced molecule sharding for graphs of very different sizes (RNA-Puzzles: 841-3 823 nodes, PDBbind
    complexes): greedy longest-first bin packing by number of nodes -- every rank gets the list of graph
    indices whose node counts sum to roughly total / world.  Deterministic (ties broken by index), identical on every
    rank; each rank's list is returned sorted so that a shard keeps the dataset's order.  QM9-sized molecules are
    uniform enough for shard_range (equal counts)."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads, shards = [0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        loads[r] += int(sizes[i])
        shards[r].append(i)
    return [sorted(s) for s in shards]

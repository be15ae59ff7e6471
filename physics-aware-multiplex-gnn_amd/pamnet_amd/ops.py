"""torch.autograd glue around the C-ABI kernels of libpamnet_hip.so.

torch is plumbing here: it owns device memory, the stream and the autograd tape; every forward/backward body below is a
HIP kernel call.  No CPU fallback: tensors that are not on an MI355X raise in lib.stream_of.
"""
import ctypes

import torch

from . import lib


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _NoGradCtx(object):
    """Stand-in for the autograd context when gradients are disabled: forward() bodies are called directly, skipping
    Function.apply (forward-only loops at the narrow widths are bound by the host, not the GPU)."""
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


import os as _os

TAPE = _os.environ.get('PAMNET_TAPE', '1') != '0'      # measurement aid: 0 = every Function is its own autograd node
import threading as _threading

_tls = _threading.local()  # .stack: the Tapes of the forwards being recorded on THIS thread (pushed / popped by
#                            _Whole.forward only; the input pipeline's worker thread must not see the trainer's tape)


def _tape_stack_of_thread():
    st = getattr(_tls, 'stack', None)
    if st is None:
        st = _tls.stack = []
    return st


def current_tape():
    st = getattr(_tls, 'stack', None)
    return st[-1] if st else None


def apply(fn, *args, tape=None):
    """fn.apply(*args) under autograd; straight fn.forward without a tape in no-grad mode; or -- inside a forward that is
    being recorded as ONE autograd node (see Tape) -- a call on that tape."""
    tape = tape if tape is not None else current_tape()
    if tape is not None:
        return tape.call(fn, *args)
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoGradCtx(), *args)


class _TapeCtx(object):
    """The slice of torch.autograd.function.FunctionCtx the Functions of this package use."""

    def __init__(self, needs):
        self.needs_input_grad = needs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class Tape(object):
    """A training forward of PAMNet is a fixed chain of this package's autograd Functions (dim = 128: eight of them --
    type gather, two Bessel bases, three embeddings, the layer stack, fusion + pooling; the narrow widths: ~50).  Handing
    them to torch's autograd engine one by one costs host time per node (bookkeeping, the hop to the engine's device
    thread, AccumulateGrad): ~0.4 ms per step at dim = 128 and most of the 3.7 ms at the narrow widths, whose steps are
    bound by the host.  With preallocated gradients (train.FlatParams) the chain is recorded here instead and the whole
    forward is ONE autograd node (`_Whole`) whose backward replays the Functions' own `backward` bodies in reverse.
    Gradients that a kernel writes in place come back as None; the others are added into the parameters' (zeroed)
    gradient views by one multi-tensor add at the end."""

    def __init__(self):
        self.nodes = []
        self.live = set()                      # ids of tensors a gradient flows back to

    def call(self, fn, *args):
        live = self.live
        needs = tuple(isinstance(a, torch.Tensor) and (a.requires_grad or id(a) in live) for a in args)
        ctx = _TapeCtx(needs)
        out = fn.forward(ctx, *args)
        outs = out if isinstance(out, tuple) else (out,)
        if any(needs):
            self.nodes.append((fn, ctx, args, outs))
            for o in outs:
                if isinstance(o, torch.Tensor):
                    live.add(id(o))
        return out

    def backward(self, out, grad):
        grads = {id(out): grad}
        pgrads = {}                            # parameter -> summed gradient (a parameter may feed several nodes)
        for fn, ctx, args, outs in reversed(self.nodes):
            gouts = [grads.pop(id(o), None) for o in outs]
            if all(g is None for g in gouts):
                continue
            gin = fn.backward(ctx, *gouts)
            if not isinstance(gin, tuple):
                gin = (gin,)
            for a, g in zip(args, gin):
                if g is None or not isinstance(a, torch.Tensor):
                    continue
                if isinstance(a, torch.nn.Parameter):
                    g = g if g.shape == a.grad.shape else g.reshape(a.grad.shape)
                    prev = pgrads.get(id(a))
                    pgrads[id(a)] = (a.grad, g if prev is None else prev[1] + g)
                elif id(a) in grads:
                    grads[id(a)] = grads[id(a)] + g
                else:
                    grads[id(a)] = g
        if pgrads:                             # one multi-tensor add into the (zeroed) flat gradient views
            torch._foreach_add_([v for v, _ in pgrads.values()], [g for _, g in pgrads.values()])
        self.nodes, self.live = [], set()


class _Whole(torch.autograd.Function):
    """The whole forward as one autograd node (see Tape).  `anchor`: any parameter that requires grad -- it makes
    autograd record the node; every parameter gradient is written / added in place, so the node returns none."""

    @staticmethod
    def forward(ctx, anchor, run, before=None, after=None):
        if before is not None:               # (models._FlatView: the flat gradient buffer starts zeroed)
            before()
        tape = Tape()
        stack = _tape_stack_of_thread()
        stack.append(tape)
        try:
            out = run(tape)
        finally:
            stack.pop()
        ctx.tape, ctx.out, ctx.after = tape, out, after
        return out.view(-1)                  # a fresh tensor object for autograd; the tape keys on `out` itself

    @staticmethod
    def backward(ctx, g):
        ctx.tape.backward(ctx.out, g.contiguous())
        if ctx.after is not None:            # (models._FlatView: the anchor's .grad is the buffer the kernels wrote)
            ctx.after()
        ctx.tape = ctx.out = ctx.after = None
        return None, None, None, None


def run_whole(anchor, run, before=None, after=None):
    return _Whole.apply(anchor, run, before, after)


def backward_whole(out, grad):
    """Backward of a forward recorded as ONE node (run_whole), called directly: the tape is replayed on the calling
    thread, without torch.autograd's graph task and its hop to the engine's device thread (~50 us of host time per
    step -- the narrow-width steps are bound by the host).  Same bodies, same order, same gradients as out.backward(grad).
    Returns False when `out` is not such an output (the caller then uses autograd)."""
    node = out.grad_fn
    if node is None or getattr(node, 'tape', None) is None or not isinstance(node, torch.autograd.function.BackwardCFunction) \
            or not isinstance(node, _Whole._backward_cls):
        return False
    with torch.no_grad():
        node.tape.backward(node.out, grad.contiguous())
        if getattr(node, 'after', None) is not None:
            node.after()
    node.tape = node.out = node.after = None
    return True


def segment_sum_raw(out, init, A, ia, B, ib, perm, ptr, rows, d):
    lib.call('pamnet_segment_sum_f32', lib.ptr(out), lib.ptr(init), lib.ptr(A), lib.ptr(ia), lib.ptr(B), lib.ptr(ib),
             lib.ptr(perm), lib.ptr(ptr), rows, d, lib.stream_of(A))
    return out


def gather_mul_raw(out, A, ia, B, ib, m, d):
    lib.call('pamnet_gather_mul_f32', lib.ptr(out), lib.ptr(A), lib.ptr(ia), lib.ptr(B), lib.ptr(ib), m, d,
             lib.stream_of(A))
    return out


class _Aggregate(torch.autograd.Function):
    """out[r] = (init[r]) + sum_{q in CSR row r} src[q]        (torch_scatter.scatter add with sorted index;
    layers/local_message_passing.py:54, layers/global_message_passing.py:38)."""

    @staticmethod
    def forward(ctx, src, init, csr):
        src = _c(src)
        d = src.size(1)
        out = torch.empty((csr.rows, d), dtype=src.dtype, device=src.device)
        segment_sum_raw(out, _c(init) if init is not None else None, src, None, None, None, None, csr.ptr, csr.rows, d)
        ctx.csr, ctx.has_init = csr, init is not None
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        csr = ctx.csr
        d = g.size(1)
        gsrc = torch.empty((csr.m, d), dtype=g.dtype, device=g.device)
        gather_mul_raw(gsrc, g, csr.row_of, None, None, csr.m, d)
        return gsrc, (g if ctx.has_init else None), None


class _Gather(torch.autograd.Function):
    """out[k] = x[idx[k]]  (x[i], x[j] in MessagePassing.propagate / local_message_passing.py:46).
    `tr` is the transposed CSR of idx (graph.Transpose); when idx is a CSR's row_of, pass (ptr, None)."""

    @staticmethod
    def forward(ctx, x, idx, tr_ptr, tr_perm):
        x = _c(x)
        m, d = idx.numel(), x.size(1)
        out = torch.empty((m, d), dtype=x.dtype, device=x.device)
        gather_mul_raw(out, x, idx, None, None, m, d)
        ctx.tr_ptr, ctx.tr_perm, ctx.rows = tr_ptr, tr_perm, x.size(0)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        d = g.size(1)
        gx = torch.empty((ctx.rows, d), dtype=g.dtype, device=g.device)
        segment_sum_raw(gx, None, g, None, None, None, ctx.tr_perm, ctx.tr_ptr, ctx.rows, d)
        return gx, None, None, None


class _GatherMulAggregate(torch.autograd.Function):
    """out[r] = sum_{q in CSR row r} A[col[q]] * B[q]
    (m_neighbor[idx] * mlp_sbf(sbf) -> scatter to edges, layers/local_message_passing.py:49-50)."""

    @staticmethod
    def forward(ctx, A, B, csr, tr):
        A, B = _c(A), _c(B)
        d = A.size(1)
        out = torch.empty((csr.rows, d), dtype=A.dtype, device=A.device)
        segment_sum_raw(out, None, A, csr.col, B, None, None, csr.ptr, csr.rows, d)
        ctx.save_for_backward(A, B)
        ctx.csr, ctx.tr = csr, tr
        return out

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g = _c(g)
        csr, tr = ctx.csr, ctx.tr
        d = g.size(1)
        gB = torch.empty_like(B)
        gather_mul_raw(gB, A, csr.col, g, csr.row_of, csr.m, d)                 # dB[q] = A[col[q]] * g[row_of[q]]
        gA = torch.empty_like(A)
        # dA[n] = sum_{q: col[q]=n} B[q] * g[row_of[q]]  -- transposed CSR walk
        segment_sum_raw(gA, None, B, None, g, csr.row_of, tr.perm, tr.ptr, A.size(0), d)
        return gA, gB, None, None


_SCRATCH = {}


def reduce_scratch(dev):
    """Scratch of pamnet_type_rows_grad_f32 (csrc/reduce.hip), one per (device, stream)."""
    import ctypes
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    t = _SCRATCH.get(key)
    if t is None:
        need = ctypes.c_int64(0)
        lib.call('pamnet_reduce_scratch_bytes', ctypes.addressof(need))
        t = _SCRATCH[key] = torch.zeros(int(need.value) // 4, dtype=torch.int32, device=dev)
    return t


class _TypeRows(torch.autograd.Function):
    """out[k] = table[idx[k]] for a table of <= 8 rows (`embeddings[x]`, models.py:107,140).  The backward sums the rows
    of the incoming gradient per type in one launch -- no transposed index list to build."""

    @staticmethod
    def forward(ctx, table, idx, direct):
        table = _c(table)
        m, d = idx.numel(), table.size(1)
        out = torch.empty((m, d), dtype=table.dtype, device=table.device)
        gather_mul_raw(out, table, idx, None, None, m, d)
        ctx.idx, ctx.shape, ctx.direct = idx, table.shape, direct
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        rows, d = ctx.shape
        gt = ctx.direct if ctx.direct is not None else torch.empty((rows, d), dtype=g.dtype, device=g.device)
        lib.call('pamnet_type_rows_grad_f32', lib.ptr(g), lib.ptr(ctx.idx), ctx.idx.numel(), rows, d,
                 lib.ptr(reduce_scratch(g.device)), lib.ptr(gt), lib.stream_of(g))
        return (None if ctx.direct is not None else gt), None, None


def type_rows_supported(table):
    d4 = table.size(1) // 4
    return table.is_cuda and table.size(0) <= 8 and table.size(1) % 4 == 0 and 1 <= d4 <= 64 and (d4 & (d4 - 1)) == 0


def type_rows(table, idx, direct_grad=None, tape=None):
    """direct_grad: a preallocated gradient buffer of `table` to be overwritten in place (train.FlatParams), or None."""
    return apply(_TypeRows, table, idx, direct_grad, tape=tape)


LOSS_ENTRIES = {'l1': 'pamnet_l1_loss_f32', 'mse': 'pamnet_mse_loss_f32', 'smooth_l1': 'pamnet_smooth_l1_loss_f32'}


def loss_with_grad(kind, out, y, grad_scale=1.0):
    """(loss, d loss / d out * grad_scale) of the three drivers' losses, mean reduction, in one launch: 'l1' = F.l1_loss
    (main_qm9.py:108), 'mse' = F.mse_loss (main_pdbbind.py:93), 'smooth_l1' = F.smooth_l1_loss, beta = 1
    (main_rna_puzzles.py:92).  `y` is brought to out's device / dtype / shape first (the torch losses convert, broadcast
    or raise; the kernel reads out.numel() contiguous fp32 values): a [B, 1] or float64 target works, a target of
    another size raises."""
    entry = LOSS_ENTRIES.get(kind)
    if entry is None:
        raise ValueError("loss must be one of 'l1', 'mse', 'smooth_l1' (got %r)" % (kind,))
    out = _c(out.detach())
    if y.numel() != out.numel():
        raise ValueError('%s loss: target has %d elements, the model output %d' % (kind, y.numel(), out.numel()))
    if y.device != out.device or y.dtype != torch.float32 or y.shape != out.shape:
        y = y.to(device=out.device, dtype=torch.float32).reshape(out.shape)
    y = _c(y)
    loss = torch.empty(1, dtype=torch.float32, device=out.device)
    d_out = torch.empty_like(out)
    lib.call(entry, lib.ptr(out), lib.ptr(y), out.numel(), float(grad_scale), lib.ptr(loss), lib.ptr(d_out),
             lib.stream_of(out))
    return loss[0], d_out


def l1_loss_with_grad(out, y, grad_scale=1.0):
    return loss_with_grad('l1', out, y, grad_scale)


def sumsq_partials(flat):
    """256 fp64 partial sums of squares of a flat fp32 buffer (one launch); ||flat|| = sqrt(partials.sum()).  The
    optimiser kernel (pamnet_adam_ema_norm_f32) finishes the sum itself."""
    part = torch.empty(256, dtype=torch.float64, device=flat.device)
    lib.call('pamnet_sumsq_partials_f32', lib.ptr(flat), flat.numel(), lib.ptr(part), lib.stream_of(flat))
    return part


class _RBF(torch.autograd.Function):
    """BesselBasisLayer (layers/basic.py:59-76); freq is trainable, dist is not differentiated (pos has no grad)."""

    @staticmethod
    def forward(ctx, dist, freq, cutoff, exponent=5):
        m = dist.numel()
        out = torch.empty((m, 16), dtype=torch.float32, device=dist.device)
        if exponent == 5:
            lib.call('pamnet_rbf_fwd_f32', lib.ptr(dist), lib.ptr(_c(freq)), float(cutoff), m, lib.ptr(out),
                     lib.stream_of(dist))
        else:                                   # any other envelope exponent: the run-time form of the same kernel
            lib.call('pamnet_rbf_fwd_env_f32', lib.ptr(dist), lib.ptr(_c(freq)), float(cutoff), int(exponent), m,
                     lib.ptr(out), lib.stream_of(dist))
        ctx.save_for_backward(dist, freq)
        ctx.cutoff, ctx.exponent = float(cutoff), int(exponent)
        return out

    @staticmethod
    def backward(ctx, g):
        dist, freq = ctx.saved_tensors
        g = _c(g)
        dfreq = torch.empty(16, dtype=torch.float32, device=g.device)
        partial = torch.empty(16 * 2048, dtype=torch.float32, device=g.device)
        if ctx.exponent == 5:
            lib.call('pamnet_rbf_bwd_f32', lib.ptr(dist), lib.ptr(_c(freq)), ctx.cutoff, dist.numel(), lib.ptr(g),
                     lib.ptr(dfreq), lib.ptr(partial), lib.stream_of(g))
        else:
            lib.call('pamnet_rbf_bwd_env_f32', lib.ptr(dist), lib.ptr(_c(freq)), ctx.cutoff, ctx.exponent, dist.numel(),
                     lib.ptr(g), lib.ptr(dfreq), lib.ptr(partial), lib.stream_of(g))
        return None, dfreq, None, None


class _FusePool(torch.autograd.Function):
    """Attention fusion + pooling (models.py:206-224).  outs / atts: [2L, N] rows (global_0, local_0, global_1, ...)."""

    @staticmethod
    def forward(ctx, outs, atts, graph, mean):
        outs, atts = _c(outs), _c(atts)
        n_layer, n = outs.size(0) // 2, outs.size(1)
        node_out = torch.empty(n, dtype=outs.dtype, device=outs.device)
        gout = torch.empty(graph.n_graphs, dtype=outs.dtype, device=outs.device)
        lib.call('pamnet_fuse_pool_fwd_f32', lib.ptr(outs), lib.ptr(atts), n_layer, n, lib.ptr(graph.sign),
                 lib.ptr(graph.gptr), graph.n_graphs, 1 if mean else 0, lib.ptr(node_out), lib.ptr(gout),
                 lib.stream_of(outs))
        ctx.save_for_backward(outs, atts)
        ctx.graph, ctx.mean = graph, mean
        ctx.mark_non_differentiable(node_out)
        return gout, node_out

    @staticmethod
    def backward(ctx, g, _g_node):
        outs, atts = ctx.saved_tensors
        graph = ctx.graph
        g = _c(g)
        n_layer, n = outs.size(0) // 2, outs.size(1)
        go, ga = torch.empty_like(outs), torch.empty_like(atts)
        lib.call('pamnet_fuse_pool_bwd_f32', lib.ptr(outs), lib.ptr(atts), n_layer, n, lib.ptr(graph.sign),
                 lib.ptr(graph.node_graph), lib.ptr(graph.gptr), 1 if ctx.mean else 0, lib.ptr(g), lib.ptr(go),
                 lib.ptr(ga), lib.stream_of(g))
        return go, ga, None, None


def _dense_scratch(n, k, m, like):
    floats = ctypes.c_int64(0)
    lib.call('pamnet_dense_scratch_floats', n, k, m, ctypes.addressof(floats))
    return torch.empty(floats.value, dtype=torch.float32, device=like.device)


def _dense_fwd(x, w, b, act, keep_z):
    if x.dtype != torch.float32 or w.dtype != torch.float32:   # (the kernels read fp32 through raw pointers)
        raise TypeError('pamnet dense kernels compute in float32: got %s x %s' % (x.dtype, w.dtype))
    n, k = x.shape
    m = w.size(0)
    y = torch.empty(n, m, dtype=x.dtype, device=x.device)
    z = torch.empty_like(y) if (act and keep_z) else None
    lib.call('pamnet_dense_fwd_f32', lib.ptr(x), k, lib.ptr(w), k, lib.ptr(b), n, k, m, 1 if act else 0, lib.ptr(z),
             lib.ptr(y), lib.stream_of(x))
    return y, z


def _dense_bwd(g, z, x, w, act, need_dx, need_dw, need_db):
    """(dx, dw, db) of y = act(x w^T + b) for the upstream gradient g: one GEMM launch for dx, one split launch + its
    fixed-order reduction for dw / db; SiLU'(z) is applied while g is staged."""
    n, k = x.shape
    m = g.size(1)                                              # (w is read for dx only and may be None otherwise)
    dx = torch.empty(n, k, dtype=g.dtype, device=g.device) if need_dx else None
    need_dw = need_dw or need_db
    dw = torch.empty(m, k, dtype=g.dtype, device=g.device) if need_dw else None
    db = torch.empty(m, dtype=g.dtype, device=g.device) if need_db else None
    scratch = _dense_scratch(n, k, m, g) if need_dw else None
    lib.call('pamnet_dense_bwd_f32', lib.ptr(g), lib.ptr(z), lib.ptr(x), k, lib.ptr(w), k, n, k, m, 1 if act else 0,
             lib.ptr(dx), lib.ptr(dw), lib.ptr(db), lib.ptr(scratch), lib.stream_of(g))
    return dx, dw, db


class _Dense(torch.autograd.Function):
    """act(x W^T + b) for a width no engine is built for (hidden sizes above 128, models.py:25; the thin bias-free
    `init_linear` of the PDBbind branch, models.py:119, at such a dim): csrc/dense.hip -- fp32-accurate GEMMs on the bf16
    matrix pipe for the forward, dx and dW (+ db) alike; no library GEMM, no transposed copies.  A Function of this package
    so that a recorded forward (Tape) differentiates it like every other stage."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x, w = _c(x), _c(w)
        b = _c(b) if b is not None else None
        needs = ctx.needs_input_grad
        y, z = _dense_fwd(x, w, b, act, keep_z=any(needs))
        ctx.save_for_backward(x, w, z)
        ctx.act, ctx.has_b = act, b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, z = ctx.saved_tensors
        needs = tuple(ctx.needs_input_grad) + (False,) * 4
        dx, dw, db = _dense_bwd(_c(g), z, x, w, ctx.act, needs[0], needs[1], ctx.has_b and needs[2])
        return dx, (dw if needs[1] else None), db, None


def dense(x, w, b=None, act=False, tape=None):
    """act(x w^T + b), act = SiLU or identity, on the hand-written GEMM kernels (any width)."""
    return apply(_Dense, x, w, b, act, tape=tape)


def plain_linear(x, w, tape=None):
    return apply(_Dense, x, w, None, False, tape=tape)


class _DenseAct(torch.autograd.Function):
    """SiLU(x W_kind^T + b_kind) for an input width no embedding kernel is built for (the spherical-basis embedding with a
    non-default num_spherical * num_radial, models.py:187-188) on csrc/dense.hip.  `kind` (int32 [rows], nullable): rows of
    kind 0 use (wa, ba), the others (wb, bb) -- both layers run over all rows and a row keeps its own (the second set is
    the pairs' embedding of the small model: same rows, other weights).  The input carries no gradient (geometry only)."""

    @staticmethod
    def forward(ctx, x, kind, wa, ba, wb, bb):
        x = _c(x)
        keep = any(ctx.needs_input_grad)
        ya, za = _dense_fwd(x, _c(wa), _c(ba), True, keep)
        if kind is None:
            ctx.save_for_backward(x, za, None, None)
            return ya
        yb, zb = _dense_fwd(x, _c(wb), _c(bb), True, keep)
        mask = (kind == 0).unsqueeze(1)
        ctx.save_for_backward(x, za, zb, mask)
        return torch.where(mask, ya, yb)

    @staticmethod
    def backward(ctx, g):
        x, za, zb, mask = ctx.saved_tensors
        g = _c(g)
        if mask is None:
            _, dwa, dba = _dense_bwd(g, za, x, None, True, False, True, True)
            return None, None, dwa, dba, None, None
        zero = torch.zeros((), dtype=g.dtype, device=g.device)
        _, dwa, dba = _dense_bwd(torch.where(mask, g, zero), za, x, None, True, False, True, True)
        _, dwb, dbb = _dense_bwd(torch.where(mask, zero, g), zb, x, None, True, False, True, True)
        return None, None, dwa, dba, dwb, dbb


def dense_act(x, lin_a, lin_b=None, kind=None, tape=None):
    if kind is None:
        return apply(_DenseAct, x, None, lin_a.weight, lin_a.bias, None, None, tape=tape)
    return apply(_DenseAct, x, kind, lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias, tape=tape)


class _StackRows(torch.autograd.Function):
    """torch.stack of per-layer [N] rows (models.py:206-207) as a Function of this package (so that it can run on a Tape)."""

    @staticmethod
    def forward(ctx, *rows):
        return torch.stack(rows)

    @staticmethod
    def backward(ctx, g):
        return tuple(g.unbind(0))


def stack_rows(rows):
    return apply(_StackRows, *rows)


def aggregate(src, csr, init=None):
    return apply(_Aggregate, src, init, csr)


def gather(x, idx, tr_ptr, tr_perm=None):
    return apply(_Gather, x, idx, tr_ptr, tr_perm)


def gather_mul_aggregate(A, B, csr, tr):
    return apply(_GatherMulAggregate, A, B, csr, tr)


def rbf(dist, freq, cutoff, tape=None, exponent=5):
    return apply(_RBF, dist, freq, cutoff, exponent, tape=tape)


def fuse_pool(outs, atts, graph, mean, tape=None):
    return apply(_FusePool, outs, atts, graph, mean, tape=tape)

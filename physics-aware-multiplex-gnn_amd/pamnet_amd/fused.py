"""Autograd functions over the fused fp32-MFMA kernels (csrc/*.hip through the C ABI).  dim = 128 only.

  * `layer_stack` / `_Stack`: the whole n_layer x (global, local) loop as ONE engine call per direction
    (csrc/engine.hip) -- what models.PAMNet uses.  ~7 launches per layer pair forward, ~16 backward; the reference
    issues ~150 per layer pair for the same work.
  * `embed` / `_Embed`: the thin input-embedding layers (csrc/embed.hip).
  * `global_layer`, `local_layer`, `node_tail`: one autograd function per layer / per node chain, built from the same
    kernels -- the granularity the kernel-level parity tests (tests/test_hip_fused.py) exercise.
torch only provides the buffers and the tape between these functions.
"""
import ctypes
import os

import torch

from . import lib
from .ops import apply as _apply, gather_mul_raw, segment_sum_raw

D = 128

# Direct gradients: when every parameter of a layer owns a preallocated `.grad` and carries `_pamnet_direct = True`
# (train.FlatParams hands both out: views of one flat fp32 buffer), the backward kernels write the gradients straight
# into those buffers and return None to autograd -- no per-parameter AccumulateGrad add (~330 launches per step at L=6).
# Semantics: overwrite; valid because every parameter is used exactly once per forward and the trainer runs one backward
# per zero_grad.  The permission is a property of the parameters (i.e. of one model), not of the process.


class StackCtx(object):
    """Per-model hand-over between a trainer and the layer-stack backward: `events` = n_layer torch.cuda.Event (already
    recorded once, so their handles exist) the engine records as each layer pair's gradients are enqueued -- the trainer
    overlaps the gradient all-reduce of the last layers with the backward of the first ones; `recorded` is set by the
    backward when it handed the events to the engine."""

    def __init__(self):
        self.events, self.recorded = None, False


def stack_ctx(global_layers):
    """The StackCtx of a model's layer stack (created on first use, stored on the global_layer ModuleList)."""
    ctx = getattr(global_layers, '_pamnet_ctx', None)
    if ctx is None:
        ctx = global_layers._pamnet_ctx = StackCtx()
    return ctx


def _parr(tensors):
    """Host array of device pointers (NULL for None; ints are raw addresses)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else (t if isinstance(t, int) else t.data_ptr())
    return arr


def _iarr(vals, ctype=ctypes.c_int64):
    return (ctype * len(vals))(*vals)


def _empty(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _sub(w, c0):
    """Address of the 128-column block starting at column c0 of a row-major weight."""
    return w.data_ptr() + 4 * c0


def alloc_like_grouped(params, groups=None):
    """Fresh gradient tensors for `params` with ONE allocation and one `unbind` per distinct shape instead of one
    `empty_like` per tensor (a PAMNet layer stack has ~380 parameters of ~6 shapes: 380 allocator calls were ~1.5 ms of
    host time per backward on the plain-autograd path).  Every result is a contiguous tensor of its own TensorImpl, which
    autograd's AccumulateGrad takes over without a copy.  `groups`: the {shape: [indices]} map when the caller caches it.
    Returns (tensors, device addresses)."""
    if groups is None:
        groups = {}
        for i, p in enumerate(params):
            groups.setdefault((tuple(p.shape), p.dtype, p.device), []).append(i)
    out, ptrs = [None] * len(params), [0] * len(params)
    for (shape, dtype, device), idxs in groups.items():
        buf = torch.empty((len(idxs),) + shape, dtype=dtype, device=device)
        base, step = buf.data_ptr(), buf.element_size() * (buf.numel() // max(len(idxs), 1))
        for k, (i, v) in enumerate(zip(idxs, buf.unbind(0))):
            out[i], ptrs[i] = v, base + k * step
    return out, ptrs


def _grad_buffers(params):
    """(direct, buffers): where the gradients of `params` (a list of Parameters / tensors) are written."""
    if all(getattr(p, '_pamnet_direct', False) and getattr(p, 'grad', None) is not None
           and p.grad.is_contiguous() for p in params):
        return True, [p.grad for p in params]
    return False, alloc_like_grouped(params)[0]


def wgrad(jobs, ref):
    """jobs: list of (dZ, ld_dz, A, ld_a, a_mode, rows, dW(tensor or address), ld_dw, db).  One launch for all."""
    if not jobs:
        return
    n = len(jobs)
    rows = _iarr([j[5] for j in jobs])
    need = ctypes.c_int64(0)
    lib.call('pamnet_wgrad_scratch_floats', n, rows, ctypes.addressof(need))
    partial = torch.empty(int(need.value), dtype=torch.float32, device=ref.device)
    lib.call('pamnet_wgrad_batched_f32', n, _parr([j[0] for j in jobs]), _iarr([j[1] for j in jobs]),
             _parr([j[2] for j in jobs]), _iarr([j[3] for j in jobs]), _iarr([j[4] for j in jobs], ctypes.c_int32),
             rows, _parr([j[6] for j in jobs]), _iarr([j[7] for j in jobs]), _parr([j[8] for j in jobs]),
             lib.ptr(partial), None, 0, None, None, None, lib.stream_of(ref))


class DeferredWgrad(object):
    """pamnet_wgrad_deferred_f32 as the layer-stack backward drives it (csrc/engine.hip run_jobs): the split-K pass of batch
    i and the fixed-order reduction of batch i-1 are ONE launch (`wgrad_fused_kernel`, the form in the step's kernel trace);
    flush() reduces the last batch.  For measurements and kernel-level tests -- the engine calls the C entry directly."""

    def __init__(self, ref):
        nbytes = ctypes.c_int64(0)
        lib.call('pamnet_wgrad_ctx_bytes', ctypes.addressof(nbytes))
        self.ctx = (ctypes.c_char * int(nbytes.value))()            # caller-owned host memory, zeroed
        self.ref, self.parts, self.flip = ref, [None, None], 0

    def launch(self, jobs):
        """jobs: as fused.wgrad.  Consecutive launches alternate between two scratch buffers (the reduction of a batch runs
        inside the NEXT launch)."""
        n = len(jobs)
        rows = _iarr([j[5] for j in jobs])
        need = ctypes.c_int64(0)
        lib.call('pamnet_wgrad_scratch_floats', n, rows, ctypes.addressof(need))
        p = self.parts[self.flip]
        if p is None or p.numel() < need.value:
            p = self.parts[self.flip] = torch.empty(int(need.value), dtype=torch.float32, device=self.ref.device)
        self.flip ^= 1
        lib.call('pamnet_wgrad_deferred_f32', n, _parr([j[0] for j in jobs]), _iarr([j[1] for j in jobs]),
                 _parr([j[2] for j in jobs]), _iarr([j[3] for j in jobs]), _iarr([j[4] for j in jobs], ctypes.c_int32),
                 rows, _parr([j[6] for j in jobs]), _iarr([j[7] for j in jobs]), _parr([j[8] for j in jobs]),
                 lib.ptr(p), None, 0, None, None, None, None, None, None, None, ctypes.addressof(self.ctx),
                 lib.stream_of(self.ref))

    def flush(self):
        lib.call('pamnet_wgrad_flush_f32', ctypes.addressof(self.ctx), lib.stream_of(self.ref))


# ---------------------------------------------------------------------------------------------------- raw kernel calls
def k_pre_fwd(x, Wx1, bx1, wps, ldwp):
    n, nblk = x.size(0), len(wps)
    Zx1, x1, P = _empty(n, D, like=x), _empty(n, D, like=x), _empty(nblk, n, D, like=x)
    lib.call('pamnet_node_pre_fwd_f32', lib.ptr(x), n, lib.ptr(Wx1), lib.ptr(bx1), _parr(wps), ldwp, nblk,
             lib.ptr(Zx1), lib.ptr(x1), lib.ptr(P), lib.stream_of(x))
    return Zx1, x1, P


def k_pre_bwd(dP, dx1_direct, d_add, Wx1, wps, ldwp, Zx1):
    n, nblk = Zx1.size(0), len(wps)
    dZ, dx = _empty(n, D, like=Zx1), _empty(n, D, like=Zx1)
    lib.call('pamnet_node_pre_bwd_f32', lib.ptr(dP), lib.ptr(dx1_direct), lib.ptr(d_add), n, lib.ptr(Wx1), _parr(wps),
             ldwp, nblk, lib.ptr(Zx1), lib.ptr(dZ), lib.ptr(dx), 0, lib.stream_of(Zx1))
    return dZ, dx


def k_tail_fwd(x2, res_x, tp):
    n = x2.size(0)
    W, b, w_out, b_out, w_att = tp[:10], tp[10:20], tp[20], tp[21], tp[22]
    Z, R = _empty(10, n, D, like=x2), _empty(2, n, D, like=x2)
    x_out, out, att = _empty(n, D, like=x2), _empty(n, like=x2), _empty(n, like=x2)
    lib.call('pamnet_node_tail_fwd_f32', lib.ptr(x2), lib.ptr(res_x), n, _parr(W), _parr(b), lib.ptr(w_out),
             lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(x_out), lib.ptr(out), lib.ptr(att),
             None, None, None, 0, 0, None, None, None, 0, lib.stream_of(x2))
    return Z, R, x_out, out, att


def k_tail_bwd(g_x, g_out, g_att, tp, Z, d_wout, d_bout, d_watt):
    """Backward chain of the tail; head-vector gradients go to the given buffers."""
    n = Z.size(1)
    W, w_out, w_att = tp[:10], tp[20], tp[22]
    g_out = torch.zeros(n, device=Z.device) if g_out is None else g_out.contiguous()
    g_att = torch.zeros(n, device=Z.device) if g_att is None else g_att.contiguous()
    g_x = None if g_x is None else g_x.contiguous()
    dZ = _empty(10, n, D, like=Z)
    d_x2, d_resx = _empty(n, D, like=Z), _empty(n, D, like=Z)
    head_partial = _empty(((n + 15) // 16) * 257, like=Z)
    lib.call('pamnet_node_tail_bwd_f32', lib.ptr(g_x), lib.ptr(g_out), lib.ptr(g_att), n, _parr(W), lib.ptr(w_out),
             lib.ptr(w_att), lib.ptr(Z), lib.ptr(dZ), lib.ptr(d_x2), lib.ptr(d_resx), lib.ptr(head_partial),
             lib.ptr(d_wout), lib.ptr(d_watt), lib.ptr(d_bout), 0, lib.stream_of(Z))
    return dZ, d_x2, d_resx


def tail_jobs(dZ, x2, Z, R, x_out, gW, gb):
    """Weight-gradient jobs of the 10-Linear tail.  Layer inputs: x2 | SiLU(z0) | SiLU(z1) | r1 | SiLU(z3) | r2 |
    SiLU(z5) | r3 | SiLU(z7) | SiLU(z8)."""
    n = x2.size(0)
    srcs = [(x2, 0), (Z[0], 1), (Z[1], 1), (R[0], 0), (Z[3], 1), (R[1], 0), (Z[5], 1), (x_out, 0), (Z[7], 1), (Z[8], 1)]
    return [(dZ[k], D, srcs[k][0], D, srcs[k][1], n, gW[k], D, gb[k]) for k in range(10)]


def tail_params(layer):
    """(10 weights, 10 biases, w_out, b_out, w_att) of a GlobalMP / LocalMP module in kernel order."""
    lins = [layer.mlp_x2[0][0], layer.res1.mlp[0][0], layer.res1.mlp[1][0], layer.res2.mlp[0][0],
            layer.res2.mlp[1][0], layer.res3.mlp[0][0], layer.res3.mlp[1][0], layer.mlp_out[0][0],
            layer.mlp_out[1][0], layer.mlp_out[2][0]]
    return [l.weight for l in lins] + [l.bias for l in lins] + [layer.W_out.weight, layer.W_out.bias, layer.W]


def _ret(direct, grads):
    return tuple(None for _ in grads) if direct else tuple(grads)


# ---------------------------------------------------------------------------------------------------- node tail alone
class _NodeTail(torch.autograd.Function):
    """x2, res_x -> x_out, out, att  (layers/global_message_passing.py:39-50 / local_message_passing.py:55-66)."""

    @staticmethod
    def forward(ctx, x2, res_x, plist, *tp):
        x2, res_x = x2.contiguous(), res_x.contiguous()
        Z, R, x_out, out, att = k_tail_fwd(x2, res_x, tp)
        ctx.save_for_backward(x2, Z, R, x_out, *tp)
        ctx.plist = plist
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x2, Z, R, x_out = ctx.saved_tensors[:4]
        tp = ctx.saved_tensors[4:]
        direct, g = _grad_buffers(ctx.plist)
        dZ, d_x2, d_resx = k_tail_bwd(g_x, g_out, g_att, tp, Z, g[20], g[21], g[22])
        wgrad(tail_jobs(dZ, x2, Z, R, x_out, g[:10], g[10:20]), x2)
        return (d_x2, d_resx, None) + _ret(direct, g)


def node_tail(layer, x2, res_x):
    tp = tail_params(layer)
    return _NodeTail.apply(x2, res_x, tp, *tp)


# ---------------------------------------------------------------------------------------------------- input embeddings
class _Embed(torch.autograd.Function):
    """act(W_kind x + b_kind) for the thin input layers (csrc/embed.hip): mlp_rbf_g/l (K=16), mlp_sbf1/2 (K=42, weight
    set chosen per row by `kind`), init_linear (K=18, no bias / activation) -- models.py:119,185-188."""

    @staticmethod
    def forward(ctx, x, kind, act, plist, *params):
        # params: W0, b0 (or None), [W1, b1]
        x = x.contiguous()
        rows, K = x.shape
        W0, b0 = params[0], params[1]
        W1, b1 = (params[2], params[3]) if len(params) > 2 else (None, None)
        out = _empty(rows, D, like=x)
        lib.call('pamnet_embed_fwd_f32', lib.ptr(x), rows, K, lib.ptr(kind), lib.ptr(W0), lib.ptr(b0), lib.ptr(W1),
                 lib.ptr(b1), 1 if act else 0, lib.ptr(out), lib.stream_of(out))
        ctx.save_for_backward(x, kind, *[p for p in params if p is not None])
        ctx.layout = [p is not None for p in params]
        ctx.act, ctx.plist, ctx.need_dx = act, plist, bool(ctx.needs_input_grad and ctx.needs_input_grad[0])
        return out

    @staticmethod
    def backward(ctx, gout):
        x, kind = ctx.saved_tensors[:2]
        it = iter(ctx.saved_tensors[2:])
        params = [next(it) if has else None for has in ctx.layout]
        W0, b0 = params[0], params[1]
        W1, b1 = (params[2], params[3]) if len(params) > 2 else (None, None)
        gout = gout.contiguous()
        rows, K = x.shape
        direct, g = _grad_buffers(ctx.plist)
        gi = iter(g)
        gp = [next(gi) if has else None for has in ctx.layout]
        need = ctypes.c_int64(0)
        lib.call('pamnet_embed_scratch_floats', rows, K, ctypes.addressof(need))
        partial = _empty(int(need.value), like=gout)
        dx = _empty(rows, K, like=gout) if ctx.need_dx else None
        lib.call('pamnet_embed_bwd_f32', lib.ptr(x), rows, K, lib.ptr(kind), lib.ptr(W0), lib.ptr(b0), lib.ptr(W1),
                 lib.ptr(b1), 1 if ctx.act else 0, lib.ptr(gout), lib.ptr(gp[0]), lib.ptr(gp[1]),
                 lib.ptr(gp[2]) if len(gp) > 2 else None, lib.ptr(gp[3]) if len(gp) > 3 else None, lib.ptr(dx),
                 lib.ptr(partial), lib.stream_of(gout))
        grads = tuple(None if (direct or has is False) else gp[i] for i, has in enumerate(ctx.layout))
        return (dx, None, None, None) + grads


def embed(x, lin0, lin1=None, kind=None, act=True, tape=None, need_dx=False):
    """lin0 / lin1: nn.Linear(K -> 128); lin1 with `kind` (int32 [rows], 0 -> lin0, 1 -> lin1).  tape / need_dx: see
    ops.Tape (need_dx: the input's gradient is wanted -- autograd's needs_input_grad[0] when there is no autograd)."""
    params = [lin0.weight, lin0.bias]
    if lin1 is not None:
        params += [lin1.weight, lin1.bias]
    plist = [p for p in params if p is not None]
    # no-grad mode: straight to the kernel, no autograd node
    return _apply(_Embed, x, kind, act, plist, *params, tape=tape)


class InputSpec(object):
    """Non-differentiable inputs of the input stage: `layers` = [(x | None, dist | None, cutoff, kind | None, act), ...] in
    the order of the parameter groups; `types` = (idx int32 [N]) or None."""

    def __init__(self, layers, types=None):
        self.layers, self.types = layers, types


class _InputStage(torch.autograd.Function):
    """Every input embedding of a forward in one launch, their backward in two (csrc/embed.hip, pamnet_embed_multi_*):
    mlp_rbf_l / mlp_rbf_g on Bessel rows formed in the kernel from the edge lengths (models.py:185-186,
    layers/basic.py:59-76), mlp_sbf2 / mlp_sbf1 on the triplet / pair rows (models.py:187-188), and `embeddings[x]` or
    init_linear (models.py:107,119,140).  params, per layer: [freq] (layers with `dist`), W0, b0 | None, [W1, b1] (layers
    with `kind`); then the type table when spec.types is given.  Returns one [rows, 128] tensor per layer (+ the table
    rows last)."""

    @staticmethod
    def _split(spec, params):
        it = iter(params)
        groups = []
        for (x, dist, cutoff, kind, act, has_bias) in spec.layers:
            freq = next(it) if dist is not None else None
            W0 = next(it)
            b0 = next(it) if has_bias else None
            W1, b1 = (next(it), next(it)) if kind is not None else (None, None)
            groups.append((freq, W0, b0, W1, b1))
        table = next(it) if spec.types is not None else None
        return groups, table

    @staticmethod
    def forward(ctx, spec, plist, *params):
        groups, table = _InputStage._split(spec, params)
        n = len(spec.layers)
        jobs = (lib.EmbedJob * n)()
        outs = []
        ref = params[0]
        for j, ((x, dist, cutoff, kind, act, _hb), (freq, W0, b0, W1, b1)) in enumerate(zip(spec.layers, groups)):
            jb = jobs[j]
            rows = dist.numel() if dist is not None else x.size(0)
            out = _empty(rows, D, like=ref)
            jb.x, jb.dist, jb.freq = lib.ptr(x), lib.ptr(dist), lib.ptr(freq)
            jb.cutoff, jb.K, jb.act, jb.rows = float(cutoff or 0.0), (16 if dist is not None else x.size(1)), 1 if act else 0, rows
            jb.kind, jb.W0, jb.b0, jb.W1, jb.b1 = lib.ptr(kind), lib.ptr(W0), lib.ptr(b0), lib.ptr(W1), lib.ptr(b1)
            jb.out = lib.ptr(out)
            outs.append(out)
        tj = None
        if table is not None:
            idx = spec.types
            xo = _empty(idx.numel(), D, like=ref)
            tj = lib.TypeRowsJob(lib.ptr(table), lib.ptr(idx), idx.numel(), table.size(0), lib.ptr(xo), None, None, None)
            outs.append(xo)
        lib.call('pamnet_embed_multi_fwd_f32', ctypes.addressof(jobs), n, ctypes.addressof(tj) if tj is not None else None,
                 lib.stream_of(ref))
        ctx.save_for_backward(*params)
        ctx.spec, ctx.plist = spec, plist
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        from .ops import reduce_scratch
        spec, params = ctx.spec, ctx.saved_tensors
        groups, table = _InputStage._split(spec, params)
        direct, g = _grad_buffers(ctx.plist)
        ggroups, gtable = _InputStage._split(spec, g)
        n = len(spec.layers)
        jobs = (lib.EmbedJob * n)()
        ref = params[0]
        need = ctypes.c_int64(0)
        keep = []
        for j, ((x, dist, cutoff, kind, act, _hb), (freq, W0, b0, W1, b1), (gf, gW0, gb0, gW1, gb1)) in enumerate(
                zip(spec.layers, groups, ggroups)):
            jb = jobs[j]
            rows = dist.numel() if dist is not None else x.size(0)
            K = 16 if dist is not None else x.size(1)
            gout = gouts[j]
            gout = torch.zeros(rows, D, dtype=torch.float32, device=ref.device) if gout is None else gout.contiguous()
            lib.call('pamnet_embed_scratch_floats', rows, K, ctypes.addressof(need))
            partial = _empty(int(need.value), like=ref)
            keep += [gout, partial]
            jb.x, jb.dist, jb.freq = lib.ptr(x), lib.ptr(dist), lib.ptr(freq)
            jb.cutoff, jb.K, jb.act, jb.rows = float(cutoff or 0.0), K, 1 if act else 0, rows
            jb.kind, jb.W0, jb.b0, jb.W1, jb.b1 = lib.ptr(kind), lib.ptr(W0), lib.ptr(b0), lib.ptr(W1), lib.ptr(b1)
            jb.gout, jb.partial = lib.ptr(gout), lib.ptr(partial)
            jb.dW0, jb.db0, jb.dW1, jb.db1, jb.dfreq = lib.ptr(gW0), lib.ptr(gb0), lib.ptr(gW1), lib.ptr(gb1), lib.ptr(gf)
        tj = None
        if table is not None:
            idx = spec.types
            gx = gouts[n]
            gx = torch.zeros(idx.numel(), D, dtype=torch.float32, device=ref.device) if gx is None else gx.contiguous()
            keep.append(gx)
            tj = lib.TypeRowsJob(None, lib.ptr(idx), idx.numel(), table.size(0), None, lib.ptr(gx),
                                 lib.ptr(reduce_scratch(ref.device)), lib.ptr(gtable))
        lib.call('pamnet_embed_multi_bwd_f32', ctypes.addressof(jobs), n, ctypes.addressof(tj) if tj is not None else None,
                 lib.stream_of(ref))
        return (None, None) + _ret(direct, g)


def input_stage(spec, params, tape=None):
    """spec: InputSpec; params: the matching flat parameter list (see _InputStage)."""
    return _apply(_InputStage, spec, params, *params, tape=tape)


def embed_supported(x, lin):
    return x.is_cuda and lin.out_features == D and lin.in_features in (16, 18, 42)


# ---------------------------------------------------------------------------------------------------- global layer
class _GlobalLayer(torch.autograd.Function):
    """Global_MessagePassing.forward (layers/global_message_passing.py:33-56): x, e -> x_out, out, att.
    params = [Wx1, bx1, Wm, bm, Wea] + tail(23)."""

    @staticmethod
    def forward(ctx, x, e, graph, plist, *params):
        Wx1, bx1, Wm, bm, Wea = params[:5]
        tp = params[5:]
        x, e = x.contiguous(), e.contiguous()
        csr = graph.glob
        n, m = x.size(0), e.size(0)
        st = lib.stream_of(x)
        wps = [_sub(Wm, 0), _sub(Wm, D)]
        Zx1, x1, P = k_pre_fwd(x, Wx1, bx1, wps, 3 * D)
        z, ea, x2 = _empty(m, D, like=x), _empty(m, D, like=x), _empty(n, D, like=x)
        # message MLP + add-aggregation, one kernel: x2 = x1 + sum_{e -> i} msg_e   (global_message_passing.py:38,52-56)
        lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), m, n, _sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
                 lib.ptr(P[0]), lib.ptr(P[1]), lib.ptr(csr.ptr), lib.ptr(csr.row_of), lib.ptr(csr.col), None, lib.ptr(x1),
                 lib.ptr(z), lib.ptr(ea), lib.ptr(x2), st)
        Z, R, x_out, out, att = k_tail_fwd(x2, x, tp)
        ctx.save_for_backward(x, e, Zx1, z, ea, x2, Z, R, x_out, *params)
        ctx.graph, ctx.plist = graph, plist
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x, e, Zx1, z, ea, x2, Z, R, x_out = ctx.saved_tensors[:9]
        params = ctx.saved_tensors[9:]
        Wx1, bx1, Wm, bm, Wea = params[:5]
        tp = params[5:]
        graph = ctx.graph
        csr, tr = graph.glob, graph.glob_T
        n, m = x.size(0), e.size(0)
        st = lib.stream_of(x)
        direct, g = _grad_buffers(ctx.plist)
        gWx1, gbx1, gWm, gbm, gWea = g[:5]
        gt = g[5:]
        dZ, d_x2, d_resx = k_tail_bwd(g_x, g_out, g_att, tp, Z, gt[20], gt[21], gt[22])
        dz, dea, d_e = _empty(m, D, like=x), _empty(m, D, like=x), _empty(m, D, like=x)
        dP = _empty(2, n, D, like=x)
        lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_x2), m, n, lib.ptr(csr.ptr), lib.ptr(csr.row_of), None, lib.ptr(z),
                 lib.ptr(ea), _sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 0,
                 lib.ptr(dP[0]), st)                                                    # d P_i (edges into i) fused
        segment_sum_raw(dP[1], None, dz, None, None, None, tr.perm, tr.ptr, n, D)       # d P_j: edges out of j
        wps = [_sub(Wm, 0), _sub(Wm, D)]
        dZx1, dx = k_pre_bwd(dP, d_x2, d_resx, Wx1, wps, 3 * D, Zx1)
        jobs = tail_jobs(dZ, x2, Z, R, x_out, gt[:10], gt[10:20])
        jobs += [(dZx1, D, x, D, 0, n, gWx1, D, gbx1),
                 (dP[0], D, Zx1, D, 1, n, _sub(gWm, 0), 3 * D, None),
                 (dP[1], D, Zx1, D, 1, n, _sub(gWm, D), 3 * D, None),
                 (dz, D, e, D, 0, m, _sub(gWm, 2 * D), 3 * D, gbm),
                 (dea, D, e, D, 0, m, gWea, D, None)]
        wgrad(jobs, x)
        return (dx, d_e, None, None) + _ret(direct, g)


def global_layer(layer, x, e, graph):
    lin_m = layer.mlp_m[0][0]
    params = [layer.mlp_x1[0][0].weight, layer.mlp_x1[0][0].bias, lin_m.weight, lin_m.bias,
              layer.W_edge_attr.weight] + tail_params(layer)
    return _GlobalLayer.apply(x, e, graph, params, *params)


# ---------------------------------------------------------------------------------------------------- local layer
class _LocalLayer(torch.autograd.Function):
    """Local_MessagePassing(_s).forward (layers/local_message_passing.py:36-66, 96-123).
    params = [Wx1, bx1, Wji, bji, Wkj, bkj, Ws1, bs1, Ws2, bs2, Wlr, Wlo] + tail(23)."""

    @staticmethod
    def forward(ctx, x, rbf, sbf, graph, plist, *params):
        Wx1, bx1, Wji, bji, Wkj, bkj, Ws1, bs1, Ws2, bs2, Wlr, Wlo = params[:12]
        tp = params[12:]
        x, rbf, sbf = x.contiguous(), rbf.contiguous(), sbf.contiguous()
        loc, tpc = graph.loc, graph.tp
        n, m, t = x.size(0), rbf.size(0), sbf.size(0)
        st = lib.stream_of(x)
        wps = [_sub(Wji, 0), _sub(Wkj, 0), _sub(Wji, D), _sub(Wkj, D)]
        Zx1, x1, P = k_pre_fwd(x, Wx1, bx1, wps, 3 * D)
        z_ji, z_kj, q2, q3, m_ji, m_nb = (_empty(m, D, like=x) for _ in range(6))
        wq = _parr([_sub(Wji, 2 * D), _sub(Wkj, 2 * D), Wlr, Wlo])
        lib.call('pamnet_local_edge_fwd_f32', lib.ptr(rbf), m, wq, _iarr([3 * D, 3 * D, D, D]), lib.ptr(bji),
                 lib.ptr(bkj), _parr([P[0], P[1], P[2], P[3]]), lib.ptr(loc.row_of), lib.ptr(loc.col), lib.ptr(z_ji),
                 lib.ptr(z_kj), lib.ptr(q2), lib.ptr(q3), lib.ptr(m_ji), lib.ptr(m_nb), st)
        z1, z2, s = _empty(t, D, like=x), _empty(t, D, like=x), _empty(t, D, like=x)
        lib.call('pamnet_mlp2_fwd_f32', lib.ptr(sbf), t, lib.ptr(Ws1), lib.ptr(bs1), lib.ptr(Ws2), lib.ptr(bs2),
                 lib.ptr(z1), lib.ptr(z2), lib.ptr(s), st)
        # m_t = m_ji + sum_{rows of e} m_nb[idx] * s;  x2 = x1 + sum_{e -> i} q3 * m_t     (local_message_passing.py:49-54)
        m_t, x2 = _empty(m, D, like=x), _empty(n, D, like=x)
        lib.call('pamnet_local_agg_fwd_f32', lib.ptr(m_ji), lib.ptr(m_nb), lib.ptr(s), lib.ptr(q3), lib.ptr(tpc.ptr),
                 lib.ptr(tpc.col), lib.ptr(loc.ptr), lib.ptr(x1), n, lib.ptr(m_t), lib.ptr(x2), st)
        Z, R, x_out, out, att = k_tail_fwd(x2, x, tp)
        ctx.save_for_backward(x, rbf, sbf, Zx1, z_ji, z_kj, q2, q3, m_nb, s, m_t, z1, z2, x2, Z, R, x_out, *params)
        ctx.graph, ctx.plist = graph, plist
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x, rbf, sbf, Zx1, z_ji, z_kj, q2, q3, m_nb, s, m_t, z1, z2, x2, Z, R, x_out = ctx.saved_tensors[:17]
        params = ctx.saved_tensors[17:]
        Wx1, bx1, Wji, bji, Wkj, bkj, Ws1, bs1, Ws2, bs2, Wlr, Wlo = params[:12]
        tp = params[12:]
        graph = ctx.graph
        loc, loc_T, tpc, tp_T = graph.loc, graph.loc_T, graph.tp, graph.tp_T
        n, m, t = x.size(0), rbf.size(0), sbf.size(0)
        st = lib.stream_of(x)
        direct, g = _grad_buffers(ctx.plist)
        gWx1, gbx1, gWji, gbji, gWkj, gbkj, gWs1, gbs1, gWs2, gbs2, gWlr, gWlo = g[:12]
        gt = g[12:]
        dZ, d_x2, d_resx = k_tail_bwd(g_x, g_out, g_att, tp, Z, gt[20], gt[21], gt[22])
        d_mt, d_q3 = _empty(m, D, like=x), _empty(m, D, like=x)
        d_s, d_mnb = _empty(t, D, like=x), _empty(m, D, like=x)
        # d m_t = d x2[i] * q3;  d q3 = d x2[i] * m_t;  d s[r] = m_nb[idx] * d m_t[edge];  d m_nb = transposed sum
        lib.call('pamnet_local_agg_bwd_f32', lib.ptr(d_x2), lib.ptr(loc.row_of), lib.ptr(q3), lib.ptr(m_t), lib.ptr(m_nb),
                 lib.ptr(s), lib.ptr(tpc.ptr), lib.ptr(tpc.col), lib.ptr(tpc.row_of), lib.ptr(tp_T.ptr),
                 lib.ptr(tp_T.perm), None, None, m, lib.ptr(d_mt), lib.ptr(d_q3), lib.ptr(d_s), lib.ptr(d_mnb), st)
        dz1, dz2, d_sbf = _empty(t, D, like=x), _empty(t, D, like=x), _empty(t, D, like=x)
        lib.call('pamnet_mlp2_bwd_f32', lib.ptr(d_s), t, lib.ptr(z1), lib.ptr(z2), lib.ptr(Ws1), lib.ptr(Ws2),
                 lib.ptr(dz1), lib.ptr(dz2), lib.ptr(d_sbf), 0, st)
        dz_ji, dz_kj, dq2, d_rbf = (_empty(m, D, like=x) for _ in range(4))
        wq = _parr([_sub(Wji, 2 * D), _sub(Wkj, 2 * D), Wlr, Wlo])
        lib.call('pamnet_local_edge_bwd_f32', lib.ptr(d_mt), lib.ptr(d_mnb), lib.ptr(d_q3), m, lib.ptr(z_ji),
                 lib.ptr(z_kj), lib.ptr(q2), wq, _iarr([3 * D, 3 * D, D, D]), lib.ptr(dz_ji), lib.ptr(dz_kj),
                 lib.ptr(dq2), lib.ptr(d_rbf), 0, st)
        dP = _empty(4, n, D, like=x)
        segment_sum_raw(dP[0], None, dz_ji, None, None, None, None, loc.ptr, n, D)
        segment_sum_raw(dP[1], None, dz_kj, None, None, None, None, loc.ptr, n, D)
        segment_sum_raw(dP[2], None, dz_ji, None, None, None, loc_T.perm, loc_T.ptr, n, D)
        segment_sum_raw(dP[3], None, dz_kj, None, None, None, loc_T.perm, loc_T.ptr, n, D)
        wps = [_sub(Wji, 0), _sub(Wkj, 0), _sub(Wji, D), _sub(Wkj, D)]
        dZx1, dx = k_pre_bwd(dP, d_x2, d_resx, Wx1, wps, 3 * D, Zx1)
        jobs = tail_jobs(dZ, x2, Z, R, x_out, gt[:10], gt[10:20])
        jobs += [(dZx1, D, x, D, 0, n, gWx1, D, gbx1),
                 (dP[0], D, Zx1, D, 1, n, _sub(gWji, 0), 3 * D, None),
                 (dP[1], D, Zx1, D, 1, n, _sub(gWkj, 0), 3 * D, None),
                 (dP[2], D, Zx1, D, 1, n, _sub(gWji, D), 3 * D, None),
                 (dP[3], D, Zx1, D, 1, n, _sub(gWkj, D), 3 * D, None),
                 (dz_ji, D, rbf, D, 0, m, _sub(gWji, 2 * D), 3 * D, gbji),
                 (dz_kj, D, rbf, D, 0, m, _sub(gWkj, 2 * D), 3 * D, gbkj),
                 (dq2, D, rbf, D, 0, m, gWlr, D, None),
                 (d_q3, D, rbf, D, 0, m, gWlo, D, None),
                 (dz2, D, z1, D, 1, t, gWs2, D, gbs2),
                 (dz1, D, sbf, D, 0, t, gWs1, D, gbs1)]
        wgrad(jobs, x)
        return (dx, d_rbf, d_sbf, None, None) + _ret(direct, g)


def local_layer(layer, x, rbf, sbf, graph):
    lin_ji = layer.mlp_m_ji[0][0]
    lin_kj = (layer.mlp_m_jj if layer.small else layer.mlp_m_kj)[0][0]
    s1, s2 = layer.mlp_sbf[0][0], layer.mlp_sbf[1][0]
    params = [layer.mlp_x1[0][0].weight, layer.mlp_x1[0][0].bias, lin_ji.weight, lin_ji.bias, lin_kj.weight,
              lin_kj.bias, s1.weight, s1.bias, s2.weight, s2.bias, layer.lin_rbf.weight,
              layer.lin_rbf_out.weight] + tail_params(layer)
    return _LocalLayer.apply(x, rbf, sbf, graph, params, *params)


# ---------------------------------------------------------------------------------------------------- whole layer stack
def global_params(layer):
    lin_m = layer.mlp_m[0][0]
    return [layer.mlp_x1[0][0].weight, layer.mlp_x1[0][0].bias, lin_m.weight, lin_m.bias,
            layer.W_edge_attr.weight] + tail_params(layer)


def local_params(layer):
    lin_ji = layer.mlp_m_ji[0][0]
    lin_kj = (layer.mlp_m_jj if layer.small else layer.mlp_m_kj)[0][0]
    s1, s2 = layer.mlp_sbf[0][0], layer.mlp_sbf[1][0]
    return [layer.mlp_x1[0][0].weight, layer.mlp_x1[0][0].bias, lin_ji.weight, lin_ji.bias, lin_kj.weight,
            lin_kj.bias, s1.weight, s1.bias, s2.weight, s2.bias, layer.lin_rbf.weight,
            layer.lin_rbf_out.weight] + tail_params(layer)


_AUX = {}
PACK_WEIGHTS = os.environ.get('PAMNET_PACK_WEIGHTS', '1') != '0'
AUX_FORK = os.environ.get('PAMNET_AUX_FWD', '0') != '0'      # measured: no gain at B=128 (host-side event cost, CU contention)


def _aux_fork(dev, n_layer):
    """(aux stream handle, event handle array) for the forward's x-independent branch; (None, None) when disabled."""
    if not AUX_FORK:
        return None, None
    key = (dev, n_layer)
    if key not in _AUX:
        stream = torch.cuda.Stream(device=dev)
        events = [torch.cuda.Event() for _ in range(n_layer + 1)]
        for e in events:
            e.record(torch.cuda.current_stream(dev))      # materialise the handles
        _AUX[key] = (stream, events, _parr([int(e.cuda_event) for e in events]))
    stream, _, arr = _AUX[key]
    return stream.cuda_stream, arr


def _graph_tables(graph):
    t = getattr(graph, '_tables', None)        # a graph built by the graph-construction engine carries its tables
    if t is not None:
        return t
    sizes = _iarr([graph.n, graph.glob.m, graph.loc.m, graph.tp.m])
    idx = _parr([graph.glob.ptr, graph.glob.row_of, graph.glob.col, graph.glob_T.ptr, graph.glob_T.perm,
                 graph.loc.ptr, graph.loc.row_of, graph.loc.col, graph.loc_T.ptr, graph.loc_T.perm,
                 graph.tp.ptr, graph.tp.row_of, graph.tp.col, graph.tp_T.ptr, graph.tp_T.perm,
                 getattr(graph, 'seg_cuts', None), getattr(graph, 'tT_edge', None), getattr(graph, 'tT_node', None)])
    return sizes, idx


class StackPlan(object):
    """Parameter tables of a layer stack, built once per model: walking the nn.Module tree (~400 attribute / Sequential
    lookups) and re-creating the pointer arrays cost ~0.5 ms of host time per step, comparable to enqueueing the kernels.
    The pointer arrays are rebuilt only when a parameter (or, in direct-gradient mode, a .grad) has moved."""

    def __init__(self, global_layers, local_layers):
        self.gl = [global_params(l) for l in global_layers]
        self.ll = [local_params(l) for l in local_layers]
        self.L = len(self.gl)
        self.gflat = [p for lay in self.gl for p in lay]
        self.lflat = [p for lay in self.ll for p in lay]
        self.flat = self.gflat + self.lflat
        self._probe = [self.flat[0], self.flat[len(self.flat) // 2], self.flat[-1]]
        self._pkey = self._gkey = None
        self._shape_groups = None           # {(shape, dtype, device): [indices into self.flat]} for grouped gradient allocation
        self._pack = self._temp = None
        self.ctx = stack_ctx(global_layers)

    def temp_arena(self, n_floats, dev):
        """Scratch arena of this model's engine calls (stream-ordered use; grown on demand)."""
        t = self._temp
        if t is None or t.numel() < n_floats or t.device != dev:
            t = self._temp = torch.empty(int(n_floats * 1.25) + 1024, dtype=torch.float32, device=dev)
        return t

    def pack_arena(self, dev):
        """Scratch for the fragment-ordered weight images the node chains read (re-packed by every engine call)."""
        if not PACK_WEIGHTS:
            return None
        if self._pack is None or self._pack.device != dev:
            need = ctypes.c_int64(0)
            lib.call('pamnet_stack_pack_floats', self.L, ctypes.addressof(need))
            self._pack = torch.empty(int(need.value), dtype=torch.float32, device=dev)
        return self._pack

    def param_tables(self):
        key = tuple(p.data_ptr() for p in self._probe)
        if key != self._pkey:
            self._gtab, self._ltab, self._pkey = _parr(self.gflat), _parr(self.lflat), key
        return self._gtab, self._ltab

    def direct(self):
        """True when every parameter owns a preallocated contiguous .grad handed out by train.FlatParams (which zeroes
        it every step: direct writes overwrite, they do not accumulate) with direct writes allowed."""
        if not all(getattr(p, '_pamnet_direct', False) for p in self._probe):
            return False
        grads = [p.grad for p in self._probe]
        if any(g is None for g in grads):
            return False
        key = tuple(g.data_ptr() for g in grads)
        if key != self._gkey:
            if not all(getattr(p, '_pamnet_direct', False) and getattr(p, 'grad', None) is not None
                       and p.grad.is_contiguous() for p in self.flat):
                return False
            self._ggrad, self._lgrad = _parr([p.grad for p in self.gflat]), _parr([p.grad for p in self.lflat])
            self._gkey = key
        return True


class _Stack(torch.autograd.Function):
    """The n_layer x (global, local) loop (models.py:196-204): x0, e_g, rbf_e, e_sbf -> outs [2L,N], atts [2L,N].
    One C call forward, one backward (csrc/engine.hip).  In direct-gradient mode the parameters are not autograd inputs
    (their gradients are written straight into the flat buffer): ~400 fewer edges for the autograd engine to walk."""

    @staticmethod
    def forward(ctx, x0, e_g, rbf_e, e_sbf, graph, plan, direct, save, *params):
        x0, e_g, rbf_e, e_sbf = x0.contiguous(), e_g.contiguous(), rbf_e.contiguous(), e_sbf.contiguous()
        L = plan.L
        n = x0.size(0)
        sizes, idx = _graph_tables(graph)
        need = (ctypes.c_int64 * 2)()
        lib.call('pamnet_stack_workspace', n, e_g.size(0), rbf_e.size(0), e_sbf.size(0), L,
                 ctypes.addressof(need), ctypes.addressof(need) + 8)
        saved = torch.empty(max(int(need[0]), 1), dtype=torch.float32, device=x0.device)
        temp = plan.temp_arena(int(need[1]), x0.device)
        outs, atts = _empty(2 * L, n, like=x0), _empty(2 * L, n, like=x0)
        gtab, ltab = plan.param_tables()
        aux, evs = _aux_fork(x0.device, L)
        lib.call('pamnet_stack_fwd_f32', sizes, idx, L, lib.ptr(x0), lib.ptr(e_g), lib.ptr(rbf_e), lib.ptr(e_sbf),
                 gtab, ltab, lib.ptr(saved), lib.ptr(temp), lib.ptr(outs), lib.ptr(atts),
                 1 if save else 0, lib.ptr(plan.pack_arena(x0.device)), aux, evs, lib.stream_of(x0))
        ctx.save_for_backward(x0, e_g, rbf_e, e_sbf, saved)
        ctx.graph, ctx.plan, ctx.direct, ctx.temp_floats = graph, plan, direct, int(need[1])
        ctx.mark_non_differentiable(saved)
        ctx.set_materialize_grads(False)       # else autograd zero-fills a gradient the size of `saved` every step
        return outs, atts, saved

    @staticmethod
    def backward(ctx, g_outs, g_atts, _g_saved):
        x0, e_g, rbf_e, e_sbf, saved = ctx.saved_tensors
        graph, plan, direct = ctx.graph, ctx.plan, ctx.direct
        L = plan.L
        sizes, idx = _graph_tables(graph)
        temp = plan.temp_arena(ctx.temp_floats, x0.device)
        d_x0, d_eg, d_rbf, d_sbf = (torch.empty_like(t) for t in (x0, e_g, rbf_e, e_sbf))
        gtab, ltab = plan.param_tables()
        evs = None
        if direct:
            ggrad, lgrad, g = plan._ggrad, plan._lgrad, ()
            sc = plan.ctx
            if sc.events is not None and len(sc.events) == L:
                evs = _parr([int(e.cuda_event) for e in sc.events])
                sc.recorded = True
        else:
            if plan._shape_groups is None or next(iter(plan._shape_groups))[2] != plan.flat[0].device:
                plan._shape_groups = {}                        # (first use, or the model moved)
                for i, p in enumerate(plan.flat):
                    plan._shape_groups.setdefault((tuple(p.shape), p.dtype, p.device), []).append(i)
            g, ptrs = alloc_like_grouped(plan.flat, plan._shape_groups)
            ng = len(plan.gflat)
            ggrad, lgrad = (ctypes.c_void_p * ng)(*ptrs[:ng]), (ctypes.c_void_p * (len(ptrs) - ng))(*ptrs[ng:])
        g_outs = torch.zeros(2 * L, x0.size(0), device=x0.device) if g_outs is None else g_outs.contiguous()
        g_atts = torch.zeros_like(g_outs) if g_atts is None else g_atts.contiguous()
        lib.call('pamnet_stack_bwd_f32', sizes, idx, L, lib.ptr(x0), lib.ptr(e_g), lib.ptr(rbf_e), lib.ptr(e_sbf),
                 gtab, ltab, lib.ptr(saved), lib.ptr(temp), lib.ptr(g_outs), lib.ptr(g_atts), ggrad, lgrad,
                 lib.ptr(d_x0), lib.ptr(d_eg), lib.ptr(d_rbf), lib.ptr(d_sbf), lib.ptr(plan.pack_arena(x0.device)), evs,
                 lib.stream_of(x0))
        return (d_x0, d_eg, d_rbf, d_sbf, None, None, None, None) + tuple(g)


def stack_plan(global_layers, local_layers):
    plan = getattr(global_layers, '_pamnet_plan', None)
    if plan is None or plan.L != len(global_layers):
        plan = StackPlan(global_layers, local_layers)
        global_layers._pamnet_plan = plan
    return plan


def layer_stack(global_layers, local_layers, x0, e_g, rbf_e, e_sbf, graph, tape=None):
    """Returns outs [2L,N], atts [2L,N] and the saved-activation arena (see stack_x_layers)."""
    plan = stack_plan(global_layers, local_layers)
    # inference (no gradient mode): the engine skips every store only the backward would read
    save = torch.is_grad_enabled()
    if tape is not None:                       # direct-gradient mode on the model's own tape (ops.Tape)
        return tape.call(_Stack, x0, e_g, rbf_e, e_sbf, graph, plan, True, True)
    if save and plan.direct():
        return _Stack.apply(x0, e_g, rbf_e, e_sbf, graph, plan, True, True)
    if not save:
        # forward-only: no autograd node (handing ~400 parameters to Function.apply costs ~0.1 ms of host time per
        # batch -- the forward-only loop is bound by the host, not by the 0.8 ms of kernels)
        return _apply(_Stack, x0, e_g, rbf_e, e_sbf, graph, plan, False, False)
    return _Stack.apply(x0, e_g, rbf_e, e_sbf, graph, plan, False, save, *plan.flat)


def stack_x_layers(saved, graph, n_layer):
    """Node features after every layer (global_0, local_0, ...) as views into the saved arena."""
    lay = (ctypes.c_int64 * 3)()
    lib.call('pamnet_stack_layout', graph.n, graph.glob.m, graph.loc.m, graph.tp.m, ctypes.addressof(lay))
    pair, og, ol = int(lay[0]), int(lay[1]), int(lay[2])
    n = graph.n
    xs = []
    for k in range(n_layer):
        for off in (og, ol):
            xs.append(saved[k * pair + off:k * pair + off + n * D].view(n, D))
    return xs

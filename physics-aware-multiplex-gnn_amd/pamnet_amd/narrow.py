"""Autograd glue for the narrow-width (dim 16 / 32 / 64) row kernels of csrc/narrow.hip.

The reference's RNA configurations (inference_rna_puzzles.py:29-30: dim 16, n_layer 1; main_rna_puzzles.py:52-53:
dim 64, n_layer 2) have ~10^6 global edges and triplet/pair rows per batch at a width where the per-row GEMMs are tiny:
the kernels stream the rows once, keep every intermediate in registers and recompute the forward in the backward, so
nothing of size [rows, dim] is saved besides the inputs.  No CPU fallback (lib.stream_of raises off-device).
"""
import ctypes
import os

import torch

from . import lib, ops

ENABLED = os.environ.get('PAMNET_NARROW', '1') != '0'
WIDTHS = (16, 32, 64)


def supported(x, dim):
    return ENABLED and x.is_cuda and x.dtype == torch.float32 and dim in WIDTHS


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _blocks(rows):
    n = ctypes.c_int64(0)
    lib.call('pamnet_narrow_blocks', rows, ctypes.addressof(n))
    return int(n.value)


def _empty(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


class _GlobalMessage(torch.autograd.Function):
    """x1 + sum_{q: tgt[q]=i} SiLU(P[i,:d] + P[src[q],d:] + e[q] We^T + b) * (e[q] Wea^T)
    (layers/global_message_passing.py:37-38,52-53).  `wm` is mlp_m's [d, 3d] weight: its last d columns are We."""

    @staticmethod
    def forward(ctx, x1, P, e, wm, bm, wea, csr, tr):
        d = x1.size(1)
        x1, P, e, wm, bm, wea = _c(x1), _c(P), _c(e), _c(wm), _c(bm), _c(wea)
        m = csr.m
        msg = _empty(m, d, like=e)
        we = wm[:, 2 * d:]                                   # view: row stride 3d
        lib.call('pamnet_narrow_global_fwd_f32', lib.ptr(e), m, d, lib.ptr(csr.row_of), lib.ptr(csr.col), lib.ptr(P),
                 we.data_ptr(), 3 * d, lib.ptr(bm), lib.ptr(wea), d, lib.ptr(msg), lib.stream_of(e))
        out = _empty(csr.rows, d, like=e)
        ops.segment_sum_raw(out, x1, msg, None, None, None, None, csr.ptr, csr.rows, d)
        ctx.save_for_backward(P, e, wm, bm, wea)
        ctx.csr, ctx.tr = csr, tr
        return out

    @staticmethod
    def backward(ctx, g):
        P, e, wm, bm, wea = ctx.saved_tensors
        csr, tr = ctx.csr, ctx.tr
        g = _c(g)
        d, m = g.size(1), csr.m
        if m == 0:
            return g, torch.zeros_like(P), torch.zeros_like(e), torch.zeros_like(wm), torch.zeros_like(bm), \
                torch.zeros_like(wea), None, None
        dz, de = _empty(m, d, like=g), _empty(m, d, like=g)
        partial = _empty(_blocks(m), 2 * d * d + d, like=g)
        dwe, dwea, db = _empty(d, d, like=g), _empty(d, d, like=g), _empty(d, like=g)
        we = wm[:, 2 * d:]
        lib.call('pamnet_narrow_global_bwd_f32', lib.ptr(e), m, d, lib.ptr(csr.row_of), lib.ptr(csr.col), lib.ptr(P),
                 we.data_ptr(), 3 * d, lib.ptr(bm), lib.ptr(wea), d, lib.ptr(g), lib.ptr(dz), lib.ptr(de),
                 lib.ptr(partial), lib.ptr(dwe), lib.ptr(dwea), lib.ptr(db), lib.stream_of(g))
        n = P.size(0)
        dpi, dpj = _empty(n, d, like=g), _empty(n, d, like=g)
        ops.segment_sum_raw(dpi, None, dz, None, None, None, None, csr.ptr, n, d)          # rows with tgt = i
        ops.segment_sum_raw(dpj, None, dz, None, None, None, tr.perm, tr.ptr, n, d)        # rows with src = j
        dP = torch.cat([dpi, dpj], 1)
        dwm = torch.zeros_like(wm)
        dwm[:, 2 * d:] = dwe
        return g, dP, de, dwm, db, dwea, None, None


def global_message(x1, P, e, wm, bm, wea, csr, tr):
    return ops.apply(_GlobalMessage, x1, P, e, wm, bm, wea, csr, tr)


class _LocalGate(torch.autograd.Function):
    """m_ji, m_nb of the local layer (layers/local_message_passing.py:46-48) from the node-side projections P [N, 4d] and
    the edge-side projections Q [E_l, 4d]: m_ji = SiLU(z_ji), m_nb = SiLU(z_kj) * lin_rbf(rbf)."""

    @staticmethod
    def forward(ctx, P, Q, b_ji, b_kj, csr, tr):
        P, Q, b_ji, b_kj = _c(P), _c(Q), _c(b_ji), _c(b_kj)
        m, d = csr.m, b_ji.numel()
        m_ji, m_nb = _empty(m, d, like=Q), _empty(m, d, like=Q)
        lib.call('pamnet_narrow_local_gate_fwd_f32', lib.ptr(P), lib.ptr(Q), lib.ptr(csr.row_of), lib.ptr(csr.col),
                 lib.ptr(b_ji), lib.ptr(b_kj), m, d, lib.ptr(m_ji), lib.ptr(m_nb), lib.stream_of(Q))
        ctx.save_for_backward(P, Q, b_ji, b_kj)
        ctx.csr, ctx.tr = csr, tr
        return m_ji, m_nb

    @staticmethod
    def backward(ctx, g_ji, g_nb):
        P, Q, b_ji, b_kj = ctx.saved_tensors
        csr, tr = ctx.csr, ctx.tr
        m, d, n = csr.m, b_ji.numel(), P.size(0)
        if m == 0:
            return torch.zeros_like(P), torch.zeros_like(Q), torch.zeros_like(b_ji), torch.zeros_like(b_kj), None, None
        g_ji = _c(g_ji) if g_ji is not None else torch.zeros(m, d, dtype=Q.dtype, device=Q.device)
        g_nb = _c(g_nb) if g_nb is not None else torch.zeros(m, d, dtype=Q.dtype, device=Q.device)
        dz, dQ = _empty(m, 2 * d, like=Q), _empty(m, 4 * d, like=Q)
        lib.call('pamnet_narrow_local_gate_bwd_f32', lib.ptr(P), lib.ptr(Q), lib.ptr(csr.row_of), lib.ptr(csr.col),
                 lib.ptr(b_ji), lib.ptr(b_kj), m, d, lib.ptr(g_ji), lib.ptr(g_nb), lib.ptr(dz), lib.ptr(dQ),
                 lib.stream_of(Q))
        dpi, dpj = _empty(n, 2 * d, like=Q), _empty(n, 2 * d, like=Q)
        ops.segment_sum_raw(dpi, None, dz, None, None, None, None, csr.ptr, n, 2 * d)          # edges with tgt = i
        ops.segment_sum_raw(dpj, None, dz, None, None, None, tr.perm, tr.ptr, n, 2 * d)        # edges with src = j
        db = dz.sum(0)
        return torch.cat([dpi, dpj], 1), dQ, db[:d], db[d:], None, None


def local_gate(P, Q, b_ji, b_kj, csr, tr):
    return ops.apply(_LocalGate, P, Q, b_ji, b_kj, csr, tr)


class _GateMul(torch.autograd.Function):
    """m = lin_rbf_out(rbf) * (m_ji + m_other) with lin_rbf_out(rbf) = the last d columns of the edge-side projections Q
    (layers/local_message_passing.py:53)."""

    @staticmethod
    def forward(ctx, Q, m_ji, m_other):
        d = m_ji.size(1)
        ctx.save_for_backward(Q, m_ji, m_other)
        return Q[:, 3 * d:] * (m_ji + m_other)

    @staticmethod
    def backward(ctx, g):
        Q, m_ji, m_other = ctx.saved_tensors
        d = m_ji.size(1)
        dQ = torch.zeros_like(Q)
        dQ[:, 3 * d:] = g * (m_ji + m_other)
        dm = g * Q[:, 3 * d:]
        return dQ, dm, dm


def gate_mul(Q, m_ji, m_other):
    return ops.apply(_GateMul, Q, m_ji, m_other)


class _Mlp2(torch.autograd.Function):
    """SiLU(W2 SiLU(W1 x + b1) + b2) [+ x] [+ r] on rows: mlp_sbf (layers/local_message_passing.py:24,49), the Res blocks
    (layers/basic.py:25-33) and the first two layers of mlp_out."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res_x, r):
        x, w1, b1, w2, b2 = _c(x), _c(w1), _c(b1), _c(w2), _c(b2)
        r = _c(r) if r is not None else None
        m, d = x.shape
        y = _empty(m, d, like=x)
        lib.call('pamnet_narrow_mlp2_fwd_f32', lib.ptr(x), m, d, lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2),
                 1 if res_x else 0, lib.ptr(r), lib.ptr(y), lib.stream_of(x))
        ctx.save_for_backward(x, w1, b1, w2, b2)
        ctx.res_x, ctx.has_r = bool(res_x), r is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w1, b1, w2, b2 = ctx.saved_tensors
        g = _c(g)
        m, d = x.shape
        gr = g if ctx.has_r else None
        if m == 0:
            return torch.zeros_like(x), torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), \
                torch.zeros_like(b2), None, gr
        dx = _empty(m, d, like=g) if ctx.needs_input_grad[0] else None
        partial = _empty(_blocks(m), 2 * d * d + 2 * d, like=g)
        dw, db = _empty(2, d, d, like=g), _empty(2, d, like=g)
        lib.call('pamnet_narrow_mlp2_bwd_f32', lib.ptr(x), m, d, lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2),
                 lib.ptr(g), 1 if ctx.res_x else 0, lib.ptr(dx), lib.ptr(partial), lib.ptr(dw), lib.ptr(db),
                 lib.stream_of(g))
        return dx, dw[0], db[0], dw[1], db[1], None, gr


def mlp2(x, seq, res_x=False, r=None):
    """seq = two Sequential(Linear, SiLU) blocks (MLP([d, d, d]) or the first two of a longer MLP)."""
    l1, l2 = seq[0][0], seq[1][0]
    return ops.apply(_Mlp2, x, l1.weight, l1.bias, l2.weight, l2.bias, res_x, r)


class _Project(torch.autograd.Function):
    """y[:, k d:(k+1) d] = act(x W_k^T (+ b)) for [d, d] blocks W_k = weights[wi][:, c0:c0 + d]: one dense layer
    (k = 1) or the node-side projections of the split message weights (global_message_passing.py:52,
    local_message_passing.py:46-48 after W[x_i | x_j | e] = W_i x_i + W_j x_j + W_e e)."""

    @staticmethod
    def forward(ctx, x, blocks, bias, act, *weights):
        x = _c(x)
        m, d = x.shape
        nb = len(blocks)
        y = _empty(m, nb * d, like=x)
        st = lib.stream_of(x)
        for k, (wi, c0) in enumerate(blocks):
            w = weights[wi]
            assert w.stride(1) == 1 and w.size(0) == d
            lib.call('pamnet_narrow_linear_fwd_f32', lib.ptr(x), m, d, w.data_ptr() + 4 * c0, w.stride(0),
                     lib.ptr(bias), 1 if act else 0, y.data_ptr() + 4 * k * d, nb * d, st)
        ctx.save_for_backward(x, bias, *weights)
        ctx.blocks, ctx.act = blocks, act
        return y

    @staticmethod
    def backward(ctx, g):
        x, bias = ctx.saved_tensors[:2]
        weights = ctx.saved_tensors[2:]
        g = _c(g)
        m, d = x.shape
        nb = len(ctx.blocks)
        dws = [torch.zeros_like(w) for w in weights]
        if m == 0:
            return (torch.zeros_like(x), None, (torch.zeros_like(bias) if bias is not None else None), None) + tuple(dws)
        st = lib.stream_of(g)
        dx = _empty(m, d, like=g) if ctx.needs_input_grad[0] else None
        partial = _empty(_blocks(m), d * d + d, like=g)
        db = _empty(d, like=g) if bias is not None else None
        for k, (wi, c0) in enumerate(ctx.blocks):
            w = weights[wi]
            dw = _empty(d, d, like=g)
            lib.call('pamnet_narrow_linear_bwd_f32', lib.ptr(x), m, d, w.data_ptr() + 4 * c0, w.stride(0),
                     lib.ptr(bias), 1 if ctx.act else 0, g.data_ptr() + 4 * k * d, nb * d, lib.ptr(dx),
                     1 if k > 0 else 0, lib.ptr(partial), lib.ptr(dw), lib.ptr(db), st)
            dws[wi][:, c0:c0 + d] += dw
        return (dx, None, db, None) + tuple(dws)


def linear(x, lin, act=True):
    """One Sequential(Linear, SiLU) block (act) or a bare Linear."""
    return ops.apply(_Project, x, ((0, 0),), lin.bias, act, lin.weight)


def project(x, blocks, *weights):
    """blocks: ((weight index, first column), ...) -> [rows, len(blocks) * d], no bias, no activation."""
    return ops.apply(_Project, x, tuple(blocks), None, False, *weights)


# ---- the node-update tail as ONE autograd node ------------------------------------------------------------------------
# Same six kernels per direction as linear() / mlp2() composed by hand, but a single autograd.Function: at these widths
# a training step is bound by the host (each Function.apply / backward hop costs more than its kernel), and the tail
# alone is twelve hops per layer.
def _lin_fwd(x, w, b, act, st):
    m, d = x.shape
    y = _empty(m, d, like=x)
    lib.call('pamnet_narrow_linear_fwd_f32', lib.ptr(x), m, d, lib.ptr(w), d, lib.ptr(b), 1 if act else 0, lib.ptr(y), d, st)
    return y


def _mlp2_fwd(x, w1, b1, w2, b2, res_x, r, st):
    m, d = x.shape
    y = _empty(m, d, like=x)
    lib.call('pamnet_narrow_mlp2_fwd_f32', lib.ptr(x), m, d, lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2),
             1 if res_x else 0, lib.ptr(r), lib.ptr(y), st)
    return y


def _lin_bwd(x, w, b, act, g, partial, st, need_dx=True):
    m, d = x.shape
    dx = _empty(m, d, like=x) if need_dx else None
    dw, db = _empty(d, d, like=x), _empty(d, like=x)
    lib.call('pamnet_narrow_linear_bwd_f32', lib.ptr(x), m, d, lib.ptr(w), d, lib.ptr(b), 1 if act else 0, lib.ptr(g), d,
             lib.ptr(dx), 0, lib.ptr(partial), lib.ptr(dw), lib.ptr(db), st)
    return dx, dw, db


def _mlp2_bwd(x, w1, b1, w2, b2, g, res_x, partial, st):
    m, d = x.shape
    dx = _empty(m, d, like=x)
    dw, db = _empty(2, d, d, like=x), _empty(2, d, like=x)
    lib.call('pamnet_narrow_mlp2_bwd_f32', lib.ptr(x), m, d, lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2),
             lib.ptr(g), 1 if res_x else 0, lib.ptr(dx), lib.ptr(partial), lib.ptr(dw), lib.ptr(db), st)
    return dx, dw[0], db[0], dw[1], db[1]


class _Tail(torch.autograd.Function):
    """(x_out, out, att) = tail(x, res_x): mlp_x2 -> Res1 (+ res_x) -> Res2 -> Res3 -> mlp_out -> the two heads
    (layers/global_message_passing.py:40-50, layers/local_message_passing.py:56-66).
    params: x2 (w, b), res1 / res2 / res3 (w1, b1, w2, b2 each), mlp_out (w1, b1, w2, b2, w3, b3),
    W_out.weight [1, d], W_out.bias [1], W [d, 1] = 23 tensors."""

    @staticmethod
    def forward(ctx, x, res_x, *params):
        x, res_x = _c(x), _c(res_x)
        pr = [_c(t) for t in params]
        st = lib.stream_of(x)
        h0 = _lin_fwd(x, pr[0], pr[1], True, st)
        r1 = _mlp2_fwd(h0, pr[2], pr[3], pr[4], pr[5], True, res_x, st)
        r2 = _mlp2_fwd(r1, pr[6], pr[7], pr[8], pr[9], True, None, st)
        r3 = _mlp2_fwd(r2, pr[10], pr[11], pr[12], pr[13], True, None, st)
        t = _mlp2_fwd(r3, pr[14], pr[15], pr[16], pr[17], False, None, st)
        o = _lin_fwd(t, pr[18], pr[19], True, st)
        m, d = x.shape
        out, att = _empty(m, like=x), _empty(m, like=x)
        lib.call('pamnet_narrow_heads_fwd_f32', lib.ptr(o), m, d, lib.ptr(pr[20]), lib.ptr(pr[21]), lib.ptr(pr[22]),
                 lib.ptr(out), lib.ptr(att), st)
        ctx.save_for_backward(x, h0, r1, r2, r3, t, o, *pr)
        return r3, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x, h0, r1, r2, r3, t, o = ctx.saved_tensors[:7]
        pr = ctx.saved_tensors[7:]
        m, d = x.shape
        st = lib.stream_of(x)
        if m == 0:
            return (torch.zeros_like(x), torch.zeros_like(x)) + tuple(torch.zeros_like(q) for q in pr)
        partial = _empty(_blocks(m), 2 * d * d + 2 * d, like=x)
        g_out = _c(g_out) if g_out is not None else torch.zeros(m, dtype=x.dtype, device=x.device)
        g_att = _c(g_att) if g_att is not None else torch.zeros(m, dtype=x.dtype, device=x.device)
        g_o, dvec = _empty(m, d, like=x), _empty(2 * d + 1, like=x)
        lib.call('pamnet_narrow_heads_bwd_f32', lib.ptr(o), m, d, lib.ptr(pr[20]), lib.ptr(pr[22]), lib.ptr(g_out),
                 lib.ptr(g_att), lib.ptr(g_o), lib.ptr(partial), lib.ptr(dvec), st)
        dt, dw3, db3 = _lin_bwd(t, pr[18], pr[19], True, g_o, partial, st)
        g3, ow1, ob1, ow2, ob2 = _mlp2_bwd(r3, pr[14], pr[15], pr[16], pr[17], dt, False, partial, st)
        if g_x is not None:
            g3 = g3 + g_x
        g2, cw1, cb1, cw2, cb2 = _mlp2_bwd(r2, pr[10], pr[11], pr[12], pr[13], g3, True, partial, st)
        g1, bw1, bb1, bw2, bb2 = _mlp2_bwd(r1, pr[6], pr[7], pr[8], pr[9], g2, True, partial, st)
        g0, aw1, ab1, aw2, ab2 = _mlp2_bwd(h0, pr[2], pr[3], pr[4], pr[5], g1, True, partial, st)
        dx, dwx, dbx = _lin_bwd(x, pr[0], pr[1], True, g0, partial, st)
        return (dx, g1, dwx, dbx, aw1, ab1, aw2, ab2, bw1, bb1, bw2, bb2, cw1, cb1, cw2, cb2,
                ow1, ob1, ow2, ob2, dw3, db3, dvec[:d].view(1, d), dvec[2 * d:], dvec[d:2 * d].view(d, 1))


def tail(layer, x, res_x):
    """mlp_x2 -> Res1 (+ the layer input) -> Res2 -> Res3 -> mlp_out (global_message_passing.py:40-50)."""
    ps = [layer.mlp_x2[0][0].weight, layer.mlp_x2[0][0].bias]
    for res in (layer.res1, layer.res2, layer.res3):
        ps += [res.mlp[0][0].weight, res.mlp[0][0].bias, res.mlp[1][0].weight, res.mlp[1][0].bias]
    for k in range(3):
        ps += [layer.mlp_out[k][0].weight, layer.mlp_out[k][0].bias]
    ps += [layer.W_out.weight, layer.W_out.bias, layer.W]
    return ops.apply(_Tail, x, res_x, *ps)


class _Embed(torch.autograd.Function):
    """SiLU(W f + b) on [rows, 16 | 42] basis rows (models.py:185-188); with `kind`, rows of kind 0 use (wa, ba) and
    the others (wb, bb)."""

    @staticmethod
    def forward(ctx, f, kind, wa, ba, wb, bb):
        f, wa, ba = _c(f), _c(wa), _c(ba)
        two = kind is not None
        if two:
            wb, bb = _c(wb), _c(bb)
        m, k = f.shape
        d = wa.size(0)
        y = _empty(m, d, like=f)
        lib.call('pamnet_narrow_embed_fwd_f32', lib.ptr(f), m, k, d, lib.ptr(kind) if two else None, lib.ptr(wa),
                 lib.ptr(ba), lib.ptr(wb) if two else None, lib.ptr(bb) if two else None, lib.ptr(y), lib.stream_of(f))
        ctx.save_for_backward(f, wa, ba, *((wb, bb) if two else ()))
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        f, wa, ba = saved[:3]
        kind = ctx.kind
        two = kind is not None
        wb, bb = (saved[3], saved[4]) if two else (None, None)
        g = _c(g)
        m, k = f.shape
        d = wa.size(0)
        sets = 2 if two else 1
        if m == 0:
            z = [torch.zeros_like(wa), torch.zeros_like(ba)] + ([torch.zeros_like(wb), torch.zeros_like(bb)] if two else [None, None])
            return (torch.zeros_like(f) if ctx.needs_input_grad[0] else None), None, z[0], z[1], z[2], z[3]
        need_df = ctx.needs_input_grad[0]
        if need_df and (k != 16 or two):
            raise RuntimeError('narrow embed: input gradient only for the 16-wide single-set embedding')
        df = _empty(m, k, like=g) if need_df else None
        kp = 16 if k == 16 else 48
        partial = _empty(_blocks(m), sets * (d * kp + d), like=g)
        dw, db = _empty(sets, d, k, like=g), _empty(sets, d, like=g)
        lib.call('pamnet_narrow_embed_bwd_f32', lib.ptr(f), m, k, d, lib.ptr(kind) if two else None, lib.ptr(wa),
                 lib.ptr(ba), lib.ptr(wb) if two else None, lib.ptr(bb) if two else None, lib.ptr(g), lib.ptr(df),
                 lib.ptr(partial), lib.ptr(dw), lib.ptr(db), lib.stream_of(g))
        return df, None, dw[0], db[0], (dw[1] if two else None), (db[1] if two else None)


class _EmbedRbf(torch.autograd.Function):
    """SiLU(W rbf(dist) + b) with the Bessel rows (layers/basic.py:74-76) formed inside the kernels: neither the [m, 16]
    rows nor their gradient exist; the backward returns the gradients of the 16 frequencies, W and b.  Forward: the same
    floats as ops.rbf followed by _Embed."""

    @staticmethod
    def forward(ctx, dist, freq, cutoff, w, b):
        dist, freq, w, b = _c(dist), _c(freq), _c(w), _c(b)
        m, d = int(dist.numel()), w.size(0)
        y = _empty(m, d, like=dist)
        lib.call('pamnet_narrow_embed_rbf_fwd_f32', lib.ptr(dist), lib.ptr(freq), float(cutoff), m, d, lib.ptr(w), lib.ptr(b),
                 lib.ptr(y), lib.stream_of(dist))
        ctx.save_for_backward(dist, freq, w, b)
        ctx.cutoff = float(cutoff)
        return y

    @staticmethod
    def backward(ctx, g):
        dist, freq, w, b = ctx.saved_tensors
        g = _c(g)
        m, d = int(dist.numel()), w.size(0)
        if m == 0:
            return None, torch.zeros_like(freq), None, torch.zeros_like(w), torch.zeros_like(b)
        partial = _empty(_blocks(m), d * 16 + d + 16, like=g)
        dw, dbf = _empty(d, 16, like=g), _empty(d + 16, like=g)
        lib.call('pamnet_narrow_embed_rbf_bwd_f32', lib.ptr(dist), lib.ptr(freq), ctx.cutoff, m, d, lib.ptr(w), lib.ptr(b),
                 lib.ptr(g), lib.ptr(partial), lib.ptr(dw), lib.ptr(dbf), lib.stream_of(g))
        return None, dbf[d:], None, dw, dbf[:d]


def embed_rbf(dist, freq, cutoff, lin):
    return ops.apply(_EmbedRbf, dist, freq, cutoff, lin.weight, lin.bias)


def embed(f, lin_a, lin_b=None, kind=None):
    if kind is None:
        return ops.apply(_Embed, f, None, lin_a.weight, lin_a.bias, None, None)
    return ops.apply(_Embed, f, kind, lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias)


# ---- the whole layer loop as ONE engine call per direction (csrc/narrow_engine.hip) -------------------------------------
ENGINE = os.environ.get('PAMNET_NARROW_ENGINE', '1') != '0'      # measurement aid: 0 = the per-operator path above


class _Stack(torch.autograd.Function):
    """The n_layer x (global, local) loop (models.py:196-204) at d = 16 / 32 / 64: x0, e_g, rbf_e, e_sbf ->
    outs [2L, N], atts [2L, N].  One C call forward, one backward; in direct-gradient mode the parameters are not autograd
    inputs (their gradients are written straight into the flat buffer, see fused._Stack)."""

    @staticmethod
    def forward(ctx, x0, e_g, rbf_e, e_sbf, graph, plan, direct, *params):
        from . import fused
        x0, e_g, rbf_e, e_sbf = _c(x0), _c(e_g), _c(rbf_e), _c(e_sbf)
        L, (n, d) = plan.L, x0.shape
        sizes, idx = fused._graph_tables(graph)
        need = (ctypes.c_int64 * 2)()
        lib.call('pamnet_narrow_stack_workspace', n, e_g.size(0), rbf_e.size(0), e_sbf.size(0), L, d,
                 ctypes.addressof(need), ctypes.addressof(need) + 8)
        saved = torch.empty(max(int(need[0]), 1), dtype=torch.float32, device=x0.device)
        temp = plan.temp_arena(int(need[1]), x0.device)
        outs, atts = _empty(2 * L, n, like=x0), _empty(2 * L, n, like=x0)
        gtab, ltab = plan.param_tables()
        lib.call('pamnet_narrow_stack_fwd_f32', sizes, idx, L, d, lib.ptr(x0), lib.ptr(e_g), lib.ptr(rbf_e), lib.ptr(e_sbf),
                 gtab, ltab, lib.ptr(saved), lib.ptr(temp), lib.ptr(outs), lib.ptr(atts), lib.stream_of(x0))
        ctx.save_for_backward(x0, e_g, rbf_e, e_sbf, saved)
        ctx.graph, ctx.plan, ctx.direct, ctx.temp_floats = graph, plan, direct, int(need[1])
        ctx.mark_non_differentiable(saved)
        ctx.set_materialize_grads(False)
        return outs, atts, saved

    @staticmethod
    def backward(ctx, g_outs, g_atts, _g_saved):
        from . import fused
        x0, e_g, rbf_e, e_sbf, saved = ctx.saved_tensors
        graph, plan, direct = ctx.graph, ctx.plan, ctx.direct
        L, (n, d) = plan.L, x0.shape
        sizes, idx = fused._graph_tables(graph)
        temp = plan.temp_arena(ctx.temp_floats, x0.device)
        d_x0, d_eg, d_rbf, d_sbf = (torch.empty_like(t) for t in (x0, e_g, rbf_e, e_sbf))
        gtab, ltab = plan.param_tables()
        evs = None
        if direct:
            ggrad, lgrad, g = plan._ggrad, plan._lgrad, ()
            sc = plan.ctx
            if sc.events is not None and len(sc.events) == L:
                evs = fused._parr([int(e.cuda_event) for e in sc.events])
                sc.recorded = True
        else:
            g = [torch.empty_like(p) for p in plan.flat]
            ggrad, lgrad = fused._parr(g[:len(plan.gflat)]), fused._parr(g[len(plan.gflat):])
        g_outs = torch.zeros(2 * L, n, device=x0.device) if g_outs is None else _c(g_outs)
        g_atts = torch.zeros_like(g_outs) if g_atts is None else _c(g_atts)
        lib.call('pamnet_narrow_stack_bwd_f32', sizes, idx, L, d, lib.ptr(x0), lib.ptr(e_g), lib.ptr(rbf_e), lib.ptr(e_sbf),
                 gtab, ltab, lib.ptr(saved), lib.ptr(temp), lib.ptr(g_outs), lib.ptr(g_atts), ggrad, lgrad,
                 lib.ptr(d_x0), lib.ptr(d_eg), lib.ptr(d_rbf), lib.ptr(d_sbf), evs, lib.stream_of(x0))
        return (d_x0, d_eg, d_rbf, d_sbf, None, None, None) + tuple(g)


def engine_supported(x, graph):
    """The engine needs every index list non-empty (its backward has no zero-row special cases; such batches take the
    per-operator path)."""
    return (ENGINE and supported(x, x.size(1)) and graph.n > 0 and graph.glob.m > 0 and graph.loc.m > 0
            and graph.tp.m > 0)


def layer_stack(global_layers, local_layers, x0, e_g, rbf_e, e_sbf, graph, tape=None):
    """Returns outs [2L, N], atts [2L, N] and the saved-activation arena (see stack_x_layers)."""
    from . import fused
    plan = fused.stack_plan(global_layers, local_layers)
    if tape is not None:                       # direct-gradient mode on the model's own tape (ops.Tape)
        return tape.call(_Stack, x0, e_g, rbf_e, e_sbf, graph, plan, True)
    if not torch.is_grad_enabled():
        return ops.apply(_Stack, x0, e_g, rbf_e, e_sbf, graph, plan, False)
    if plan.direct():
        return _Stack.apply(x0, e_g, rbf_e, e_sbf, graph, plan, True)
    return _Stack.apply(x0, e_g, rbf_e, e_sbf, graph, plan, False, *plan.flat)


def stack_x_layers(saved, graph, n_layer, d):
    """Node features after every layer (global_0, local_0, ...) as views into the saved arena."""
    lay = (ctypes.c_int64 * 3)()
    lib.call('pamnet_narrow_stack_layout', graph.n, graph.glob.m, graph.loc.m, graph.tp.m, d, ctypes.addressof(lay))
    pair, og, ol = int(lay[0]), int(lay[1]), int(lay[2])
    n = graph.n
    xs = []
    for k in range(n_layer):
        for off in (og, ol):
            xs.append(saved[k * pair + off:k * pair + off + n * d].view(n, d))
    return xs

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
    return _GlobalMessage.apply(x1, P, e, wm, bm, wea, csr, tr)


class _Mlp2(torch.autograd.Function):
    """SiLU(W2 SiLU(W1 x + b1) + b2) on rows (mlp_sbf, layers/local_message_passing.py:24,49)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x, w1, b1, w2, b2 = _c(x), _c(w1), _c(b1), _c(w2), _c(b2)
        m, d = x.shape
        y = _empty(m, d, like=x)
        lib.call('pamnet_narrow_mlp2_fwd_f32', lib.ptr(x), m, d, lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2),
                 lib.ptr(y), lib.stream_of(x))
        ctx.save_for_backward(x, w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w1, b1, w2, b2 = ctx.saved_tensors
        g = _c(g)
        m, d = x.shape
        if m == 0:
            return torch.zeros_like(x), torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), \
                torch.zeros_like(b2)
        dx = _empty(m, d, like=g) if ctx.needs_input_grad[0] else None
        partial = _empty(_blocks(m), 2 * d * d + 2 * d, like=g)
        dw, db = _empty(2, d, d, like=g), _empty(2, d, like=g)
        lib.call('pamnet_narrow_mlp2_bwd_f32', lib.ptr(x), m, d, lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2),
                 lib.ptr(g), lib.ptr(dx), lib.ptr(partial), lib.ptr(dw), lib.ptr(db), lib.stream_of(g))
        return dx, dw[0], db[0], dw[1], db[1]


def mlp2(x, seq):
    """seq = MLP([d, d, d]) (two Sequential(Linear, SiLU) blocks)."""
    l1, l2 = seq[0][0], seq[1][0]
    return _Mlp2.apply(x, l1.weight, l1.bias, l2.weight, l2.bias)


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


def embed(f, lin_a, lin_b=None, kind=None):
    if kind is None:
        return _Embed.apply(f, None, lin_a.weight, lin_a.bias, None, None)
    return _Embed.apply(f, kind, lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias)

"""Autograd wrappers of the fused fp32-MFMA chain kernels (csrc/node_chain.hip, csrc/wgrad.hip, csrc/edge_chain.hip).

dim = 128 only (the width the MFMA tiles are compiled for); other widths use the generic kernels of ops.py.
"""
import ctypes

import torch

from . import lib

D = 128


def _parr(tensors):
    """Host array of device pointers (NULL for None)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _iarr(vals, ctype=ctypes.c_int64):
    return (ctype * len(vals))(*vals)


def _empty(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def wgrad(jobs, stream_tensor):
    """jobs: list of (dZ, ld_dz, A, ld_a, a_mode, rows, dW, ld_dw, db).  One launch for all of them."""
    if not jobs:
        return
    rows_max = max(j[5] for j in jobs)
    split = int(min(64, max(1, (rows_max + 511) // 512)))
    n = len(jobs)
    partial = torch.empty(n * split * (D * D + D), dtype=torch.float32, device=stream_tensor.device)
    keep = [_parr([j[0] for j in jobs]), _iarr([j[1] for j in jobs]), _parr([j[2] for j in jobs]),
            _iarr([j[3] for j in jobs]), _iarr([j[4] for j in jobs], ctypes.c_int32), _iarr([j[5] for j in jobs]),
            _parr([j[6] for j in jobs]), _iarr([j[7] for j in jobs]), _parr([j[8] for j in jobs])]
    lib.call('pamnet_wgrad_batched_f32', n, keep[0], keep[1], keep[2], keep[3], keep[4], keep[5], keep[6], keep[7],
             keep[8], split, lib.ptr(partial), lib.stream_of(stream_tensor))


def tail_params(layer):
    """(10 weights, 10 biases, w_out, b_out, w_att) of a GlobalMP / LocalMP module in kernel order."""
    lins = [layer.mlp_x2[0][0], layer.res1.mlp[0][0], layer.res1.mlp[1][0], layer.res2.mlp[0][0],
            layer.res2.mlp[1][0], layer.res3.mlp[0][0], layer.res3.mlp[1][0], layer.mlp_out[0][0],
            layer.mlp_out[1][0], layer.mlp_out[2][0]]
    return [l.weight for l in lins] + [l.bias for l in lins] + [layer.W_out.weight, layer.W_out.bias, layer.W]


class _NodeTail(torch.autograd.Function):
    """x2, res_x -> x_out, out, att  (layers/global_message_passing.py:39-50 / local_message_passing.py:55-66)."""

    @staticmethod
    def forward(ctx, x2, res_x, *params):
        x2, res_x = x2.contiguous(), res_x.contiguous()
        n = x2.size(0)
        W, b = params[:10], params[10:20]
        w_out, b_out, w_att = params[20], params[21], params[22]
        Z, R = _empty(10, n, D, like=x2), _empty(2, n, D, like=x2)
        x_out, out, att = _empty(n, D, like=x2), _empty(n, like=x2), _empty(n, like=x2)
        lib.call('pamnet_node_tail_fwd_f32', lib.ptr(x2), lib.ptr(res_x), n, _parr(W), _parr(b), lib.ptr(w_out),
                 lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(x_out), lib.ptr(out), lib.ptr(att),
                 lib.stream_of(x2))
        ctx.save_for_backward(x2, Z, R, x_out, *params)
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x2, Z, R, x_out = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        W = params[:10]
        w_out, w_att = params[20], params[22]
        n = x2.size(0)
        dev = x2
        g_out = torch.zeros(n, device=x2.device) if g_out is None else g_out.contiguous()
        g_att = torch.zeros(n, device=x2.device) if g_att is None else g_att.contiguous()
        g_x = None if g_x is None else g_x.contiguous()
        dZ = _empty(10, n, D, like=dev)
        d_x2, d_resx = _empty(n, D, like=dev), _empty(n, D, like=dev)
        grid = (n + 15) // 16
        head_partial = _empty(grid * 257, like=dev)
        gW = [torch.empty_like(w) for w in W]
        gb = [_empty(D, like=dev) for _ in range(10)]
        d_wout, d_watt, d_bout = torch.empty_like(w_out), torch.empty_like(w_att), _empty(1, like=dev)
        lib.call('pamnet_node_tail_bwd_f32', lib.ptr(g_x), lib.ptr(g_out), lib.ptr(g_att), n, _parr(W), lib.ptr(w_out),
                 lib.ptr(w_att), lib.ptr(Z), lib.ptr(dZ), lib.ptr(d_x2), lib.ptr(d_resx), lib.ptr(head_partial),
                 lib.ptr(d_wout), lib.ptr(d_watt), lib.ptr(d_bout), lib.stream_of(dev))
        # layer inputs: x2 | SiLU(z0) | SiLU(z1) | r1 | SiLU(z3) | r2 | SiLU(z5) | r3 | SiLU(z7) | SiLU(z8)
        srcs = [(x2, 0), (Z[0], 1), (Z[1], 1), (R[0], 0), (Z[3], 1), (R[1], 0), (Z[5], 1), (x_out, 0), (Z[7], 1),
                (Z[8], 1)]
        jobs = [(dZ[k], D, srcs[k][0], D, srcs[k][1], n, gW[k], D, gb[k]) for k in range(10)]
        wgrad(jobs, dev)
        return (d_x2, d_resx) + tuple(gW) + tuple(gb) + (d_wout, d_bout, d_watt)


def node_tail(layer, x2, res_x):
    return _NodeTail.apply(x2, res_x, *tail_params(layer))


class _NodePre(torch.autograd.Function):
    """x -> x1 = SiLU(mlp_x1 x), P = x1 * [Wp_0; ..; Wp_{nblk-1}]^T  with Wp_b = 128-column blocks of the message MLPs."""

    @staticmethod
    def forward(ctx, x, Wx1, bx1, ldwp, *wps):
        x = x.contiguous()
        n, nblk = x.size(0), len(wps)
        Zx1, x1, P = _empty(n, D, like=x), _empty(n, D, like=x), _empty(n, nblk * D, like=x)
        lib.call('pamnet_node_pre_fwd_f32', lib.ptr(x), n, lib.ptr(Wx1), lib.ptr(bx1), _parr(wps), ldwp, nblk,
                 lib.ptr(Zx1), lib.ptr(x1), lib.ptr(P), lib.stream_of(x))
        ctx.save_for_backward(x, Zx1, Wx1, *wps)
        ctx.ldwp = ldwp
        return x1, P

    @staticmethod
    def backward(ctx, g_x1, g_P):
        x, Zx1, Wx1 = ctx.saved_tensors[:3]
        wps = ctx.saved_tensors[3:]
        n, nblk, ldwp = x.size(0), len(wps), ctx.ldwp
        g_P = g_P.contiguous() if g_P is not None else torch.zeros(n, nblk * D, device=x.device)
        g_x1 = None if g_x1 is None else g_x1.contiguous()
        dZ, dx = _empty(n, D, like=x), _empty(n, D, like=x)
        lib.call('pamnet_node_pre_bwd_f32', lib.ptr(g_P), lib.ptr(g_x1), None, n, lib.ptr(Wx1), _parr(wps), ldwp, nblk,
                 lib.ptr(Zx1), lib.ptr(dZ), lib.ptr(dx), lib.stream_of(x))
        gWx1, gbx1 = torch.empty_like(Wx1), _empty(D, like=x)
        # d Wp_b = dP_b^T * x1 (x1 = SiLU(z_x1)) written as dense [128,128] blocks, re-assembled by the caller's views
        gwp = [_empty(D, D, like=x) for _ in range(nblk)]
        jobs = [(dZ, D, x, D, 0, n, gWx1, D, gbx1)]
        for b in range(nblk):
            jobs.append((g_P[:, b * D:], nblk * D, Zx1, D, 1, n, gwp[b], D, None))
        wgrad(jobs, x)
        return (dx, gWx1, gbx1, None) + tuple(gwp)


def node_pre(x, lin_x1, weight_blocks, ldwp):
    """weight_blocks: list of [128, 128] views (row stride ldwp) of the message-MLP weights."""
    return _NodePre.apply(x, lin_x1.weight, lin_x1.bias, ldwp, *weight_blocks)

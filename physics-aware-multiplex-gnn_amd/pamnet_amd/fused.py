"""Per-layer autograd functions built from the fused fp32-MFMA kernels (csrc/node_chain.hip, csrc/edge_chain.hip,
csrc/wgrad.hip) and the sorted segment-sum kernels (csrc/segment.hip).  dim = 128 only.

One torch.autograd.Function per message-passing layer: its forward/backward are straight sequences of C-ABI kernel
launches (no torch ops, no autograd tape inside a layer), and every parameter gradient of the layer is produced by
ONE batched weight-gradient launch.  torch only provides the buffers and the tape between layers.

Kernel count per layer pair (forward): 2 x node_pre, global_edge, local_edge, mlp2, 3 x segment-sum, 2 x node_tail = 10
launches -- the reference issues ~150 for the same work.
"""
import ctypes

import torch

from . import lib
from .ops import gather_mul_raw, segment_sum_raw

D = 128


def _parr(tensors):
    """Host array of device pointers (NULL for None)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else (t if isinstance(t, int) else t.data_ptr())
    return arr


def _iarr(vals, ctype=ctypes.c_int64):
    return (ctype * len(vals))(*vals)


def _empty(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _sub(w, c0):
    """Pointer to the 128-column block starting at column c0 of a row-major weight."""
    return w.data_ptr() + 4 * c0


def wgrad(jobs, ref):
    """jobs: list of (dZ, ld_dz, A, ld_a, a_mode, rows, dW(ptr or tensor), ld_dw, db).  One launch for all of them."""
    if not jobs:
        return
    rows_max = max(j[5] for j in jobs)
    split = int(min(64, max(1, (rows_max + 511) // 512)))
    n = len(jobs)
    partial = torch.empty(n * split * (D * D + D), dtype=torch.float32, device=ref.device)
    args = [_parr([j[0] for j in jobs]), _iarr([j[1] for j in jobs]), _parr([j[2] for j in jobs]),
            _iarr([j[3] for j in jobs]), _iarr([j[4] for j in jobs], ctypes.c_int32), _iarr([j[5] for j in jobs]),
            _parr([j[6] for j in jobs]), _iarr([j[7] for j in jobs]), _parr([j[8] for j in jobs])]
    lib.call('pamnet_wgrad_batched_f32', n, args[0], args[1], args[2], args[3], args[4], args[5], args[6], args[7],
             args[8], split, lib.ptr(partial), lib.stream_of(ref))


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
             ldwp, nblk, lib.ptr(Zx1), lib.ptr(dZ), lib.ptr(dx), lib.stream_of(Zx1))
    return dZ, dx


def k_tail_fwd(x2, res_x, tp):
    n = x2.size(0)
    W, b, w_out, b_out, w_att = tp[:10], tp[10:20], tp[20], tp[21], tp[22]
    Z, R = _empty(10, n, D, like=x2), _empty(2, n, D, like=x2)
    x_out, out, att = _empty(n, D, like=x2), _empty(n, like=x2), _empty(n, like=x2)
    lib.call('pamnet_node_tail_fwd_f32', lib.ptr(x2), lib.ptr(res_x), n, _parr(W), _parr(b), lib.ptr(w_out),
             lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(x_out), lib.ptr(out), lib.ptr(att),
             lib.stream_of(x2))
    return Z, R, x_out, out, att


def k_tail_bwd(g_x, g_out, g_att, tp, Z):
    n = Z.size(1)
    W, w_out, w_att = tp[:10], tp[20], tp[22]
    g_out = torch.zeros(n, device=Z.device) if g_out is None else g_out.contiguous()
    g_att = torch.zeros(n, device=Z.device) if g_att is None else g_att.contiguous()
    g_x = None if g_x is None else g_x.contiguous()
    dZ = _empty(10, n, D, like=Z)
    d_x2, d_resx = _empty(n, D, like=Z), _empty(n, D, like=Z)
    head_partial = _empty(((n + 15) // 16) * 257, like=Z)
    d_wout, d_watt, d_bout = torch.empty_like(w_out), torch.empty_like(w_att), _empty(1, like=Z)
    lib.call('pamnet_node_tail_bwd_f32', lib.ptr(g_x), lib.ptr(g_out), lib.ptr(g_att), n, _parr(W), lib.ptr(w_out),
             lib.ptr(w_att), lib.ptr(Z), lib.ptr(dZ), lib.ptr(d_x2), lib.ptr(d_resx), lib.ptr(head_partial),
             lib.ptr(d_wout), lib.ptr(d_watt), lib.ptr(d_bout), lib.stream_of(Z))
    return dZ, d_x2, d_resx, d_wout, d_watt, d_bout


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


def _tail_grads(tp, dev):
    gW = [torch.empty_like(w) for w in tp[:10]]
    gb = [_empty(D, like=dev) for _ in range(10)]
    return gW, gb


# ---------------------------------------------------------------------------------------------------- node tail alone
class _NodeTail(torch.autograd.Function):
    """x2, res_x -> x_out, out, att  (layers/global_message_passing.py:39-50 / local_message_passing.py:55-66)."""

    @staticmethod
    def forward(ctx, x2, res_x, *tp):
        x2, res_x = x2.contiguous(), res_x.contiguous()
        Z, R, x_out, out, att = k_tail_fwd(x2, res_x, tp)
        ctx.save_for_backward(x2, Z, R, x_out, *tp)
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x2, Z, R, x_out = ctx.saved_tensors[:4]
        tp = ctx.saved_tensors[4:]
        dZ, d_x2, d_resx, d_wout, d_watt, d_bout = k_tail_bwd(g_x, g_out, g_att, tp, Z)
        gW, gb = _tail_grads(tp, x2)
        wgrad(tail_jobs(dZ, x2, Z, R, x_out, gW, gb), x2)
        return (d_x2, d_resx) + tuple(gW) + tuple(gb) + (d_wout, d_bout, d_watt)


def node_tail(layer, x2, res_x):
    return _NodeTail.apply(x2, res_x, *tail_params(layer))


# ---------------------------------------------------------------------------------------------------- global layer
class _GlobalLayer(torch.autograd.Function):
    """Global_MessagePassing.forward (layers/global_message_passing.py:33-56): x, e -> x_out, out, att."""

    @staticmethod
    def forward(ctx, x, e, graph, Wx1, bx1, Wm, bm, Wea, *tp):
        x, e = x.contiguous(), e.contiguous()
        csr = graph.glob
        n, m = x.size(0), e.size(0)
        st = lib.stream_of(x)
        wps = [_sub(Wm, 0), _sub(Wm, D)]
        Zx1, x1, P = k_pre_fwd(x, Wx1, bx1, wps, 3 * D)
        z, ea, msg = _empty(m, D, like=x), _empty(m, D, like=x), _empty(m, D, like=x)
        lib.call('pamnet_global_edge_fwd_f32', lib.ptr(e), m, _sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
                 lib.ptr(P[0]), lib.ptr(P[1]), lib.ptr(csr.row_of), lib.ptr(csr.col), lib.ptr(z), lib.ptr(ea),
                 lib.ptr(msg), st)
        x2 = _empty(n, D, like=x)
        segment_sum_raw(x2, x1, msg, None, None, None, None, csr.ptr, n, D)            # x1 + sum_{e -> i} msg_e
        Z, R, x_out, out, att = k_tail_fwd(x2, x, tp)
        ctx.save_for_backward(x, e, Zx1, z, ea, x2, Z, R, x_out, Wx1, Wm, Wea, *tp)
        ctx.graph = graph
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        x, e, Zx1, z, ea, x2, Z, R, x_out, Wx1, Wm, Wea = ctx.saved_tensors[:12]
        tp = ctx.saved_tensors[12:]
        graph = ctx.graph
        csr, tr = graph.glob, graph.glob_T
        n, m = x.size(0), e.size(0)
        st = lib.stream_of(x)
        dZ, d_x2, d_resx, d_wout, d_watt, d_bout = k_tail_bwd(g_x, g_out, g_att, tp, Z)
        dz, dea, d_e = _empty(m, D, like=x), _empty(m, D, like=x), _empty(m, D, like=x)
        lib.call('pamnet_global_edge_bwd_f32', lib.ptr(d_x2), lib.ptr(csr.row_of), m, lib.ptr(z), lib.ptr(ea),
                 _sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 0, st)
        dP = _empty(2, n, D, like=x)
        segment_sum_raw(dP[0], None, dz, None, None, None, None, csr.ptr, n, D)         # d P_i = sum over edges into i
        segment_sum_raw(dP[1], None, dz, None, None, None, tr.perm, tr.ptr, n, D)       # d P_j = sum over edges out of j
        wps = [_sub(Wm, 0), _sub(Wm, D)]
        dZx1, dx = k_pre_bwd(dP, d_x2, d_resx, Wx1, wps, 3 * D, Zx1)
        gWx1, gbx1 = torch.empty_like(Wx1), _empty(D, like=x)
        gWm, gbm, gWea = torch.empty_like(Wm), _empty(D, like=x), torch.empty_like(Wea)
        gW, gb = _tail_grads(tp, x)
        jobs = tail_jobs(dZ, x2, Z, R, x_out, gW, gb)
        jobs += [(dZx1, D, x, D, 0, n, gWx1, D, gbx1),
                 (dP[0], D, Zx1, D, 1, n, _sub(gWm, 0), 3 * D, None),
                 (dP[1], D, Zx1, D, 1, n, _sub(gWm, D), 3 * D, None),
                 (dz, D, e, D, 0, m, _sub(gWm, 2 * D), 3 * D, gbm),
                 (dea, D, e, D, 0, m, gWea, D, None)]
        wgrad(jobs, x)
        return (dx, d_e, None, gWx1, gbx1, gWm, gbm, gWea) + tuple(gW) + tuple(gb) + (d_wout, d_bout, d_watt)


def global_layer(layer, x, e, graph):
    lin_m = layer.mlp_m[0][0]
    return _GlobalLayer.apply(x, e, graph, layer.mlp_x1[0][0].weight, layer.mlp_x1[0][0].bias, lin_m.weight,
                              lin_m.bias, layer.W_edge_attr.weight, *tail_params(layer))


# ---------------------------------------------------------------------------------------------------- local layer
class _LocalLayer(torch.autograd.Function):
    """Local_MessagePassing(_s).forward (layers/local_message_passing.py:36-66, 96-123)."""

    @staticmethod
    def forward(ctx, x, rbf, sbf, graph, Wx1, bx1, Wji, bji, Wkj, bkj, Ws1, bs1, Ws2, bs2, Wlr, Wlo, *tp):
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
        m_t = _empty(m, D, like=x)          # m_ji + sum_{rows of e} m_nb[idx] * s        (local_message_passing.py:49-51)
        segment_sum_raw(m_t, m_ji, m_nb, tpc.col, s, None, None, tpc.ptr, m, D)
        x2 = _empty(n, D, like=x)           # x1 + sum_{e -> i} q3 * m_t                    (local_message_passing.py:53-54)
        segment_sum_raw(x2, x1, m_t, None, q3, None, None, loc.ptr, n, D)
        Z, R, x_out, out, att = k_tail_fwd(x2, x, tp)
        ctx.save_for_backward(x, rbf, sbf, Zx1, z_ji, z_kj, q2, q3, m_nb, s, m_t, z1, z2, x2, Z, R, x_out,
                              Wx1, Wji, Wkj, Ws1, Ws2, Wlr, Wlo, *tp)
        ctx.graph = graph
        return x_out, out, att

    @staticmethod
    def backward(ctx, g_x, g_out, g_att):
        (x, rbf, sbf, Zx1, z_ji, z_kj, q2, q3, m_nb, s, m_t, z1, z2, x2, Z, R, x_out,
         Wx1, Wji, Wkj, Ws1, Ws2, Wlr, Wlo) = ctx.saved_tensors[:24]
        tp = ctx.saved_tensors[24:]
        graph = ctx.graph
        loc, loc_T, tpc, tp_T = graph.loc, graph.loc_T, graph.tp, graph.tp_T
        n, m, t = x.size(0), rbf.size(0), sbf.size(0)
        st = lib.stream_of(x)
        dZ, d_x2, d_resx, d_wout, d_watt, d_bout = k_tail_bwd(g_x, g_out, g_att, tp, Z)
        d_mt, d_q3 = _empty(m, D, like=x), _empty(m, D, like=x)
        gather_mul_raw(d_mt, d_x2, loc.row_of, q3, None, m, D)                          # d m_t = d x2[i] * q3
        gather_mul_raw(d_q3, d_x2, loc.row_of, m_t, None, m, D)                         # d q3  = d x2[i] * m_t
        d_s, d_mnb = _empty(t, D, like=x), _empty(m, D, like=x)
        gather_mul_raw(d_s, m_nb, tpc.col, d_mt, tpc.row_of, t, D)                      # d s[r] = m_nb[idx] * d m_t[edge]
        segment_sum_raw(d_mnb, None, s, None, d_mt, tpc.row_of, tp_T.perm, tp_T.ptr, m, D)
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
        gWx1, gbx1 = torch.empty_like(Wx1), _empty(D, like=x)
        gWji, gbji, gWkj, gbkj = torch.empty_like(Wji), _empty(D, like=x), torch.empty_like(Wkj), _empty(D, like=x)
        gWs1, gbs1, gWs2, gbs2 = torch.empty_like(Ws1), _empty(D, like=x), torch.empty_like(Ws2), _empty(D, like=x)
        gWlr, gWlo = torch.empty_like(Wlr), torch.empty_like(Wlo)
        gW, gb = _tail_grads(tp, x)
        jobs = tail_jobs(dZ, x2, Z, R, x_out, gW, gb)
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
        return ((dx, d_rbf, d_sbf, None, gWx1, gbx1, gWji, gbji, gWkj, gbkj, gWs1, gbs1, gWs2, gbs2, gWlr, gWlo)
                + tuple(gW) + tuple(gb) + (d_wout, d_bout, d_watt))


def local_layer(layer, x, rbf, sbf, graph):
    lin_ji = layer.mlp_m_ji[0][0]
    lin_kj = (layer.mlp_m_jj if layer.small else layer.mlp_m_kj)[0][0]
    s1, s2 = layer.mlp_sbf[0][0], layer.mlp_sbf[1][0]
    return _LocalLayer.apply(x, rbf, sbf, graph, layer.mlp_x1[0][0].weight, layer.mlp_x1[0][0].bias, lin_ji.weight,
                             lin_ji.bias, lin_kj.weight, lin_kj.bias, s1.weight, s1.bias, s2.weight, s2.bias,
                             layer.lin_rbf.weight, layer.lin_rbf_out.weight, *tail_params(layer))

"""Parameter containers with the reference's state_dict layout + the layer forwards on top of pamnet_amd.ops.

Module / attribute names mirror the reference so that `state_dict()` keys and shapes are identical
(the shipped checkpoint save/pamnet_rna.pt loads with strict=True):
  MLP([a,b,..])  -> `<name>.<k>.0.{weight,bias}`      (layers/basic.py:19-22)
  Res(dim)       -> `<name>.mlp.<k>.0.{weight,bias}`  (layers/basic.py:25-33)
  BesselBasis    -> `<name>.freq`                      (layers/basic.py:59-72)
The compute, however, is organised for the MI355X kernels, not as a module-by-module translation: the edge MLPs on
[x_i | x_j | e] are split algebraically into node-level and edge-level projections (W[x_i|x_j|e] = W_i x_i + W_j x_j +
W_e e), so per-edge work is gather + add + SiLU + mul + sorted segment-sum.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F     # (elementwise SiLU between two kernels of the dim > 128 formulation only)

from . import fused, narrow, ops

# dim = 128: fp32-MFMA / bf16x6 chain kernels; dim = 16 / 32 / 64: row kernels; every other dim <= 128 is built zero-padded at
# the next of these widths (models._PAMNetBase), so it runs the same engines.  Only widths above 128 reach the generic
# formulation below: one launch per dense layer on the any-width GEMM kernels of csrc/dense.hip (ops.dense: bf16x6 MFMA, bias +
# SiLU in the epilogue, SiLU' while staging the backward's operand) between the HIP graph / basis / segment kernels -- no
# library GEMM, no torch dense op: every path of this module ends in libpamnet_hip.so and raises on anything that is not an fp32
# HIP tensor (lib.stream_of).  The plain-PyTorch statement of the same layers that the kernel tests compare against lives with
# the tests (tests/torch_formulation.py).


def linear(x, w, b=None):
    """x w^T + b on the any-width GEMM kernel."""
    return ops.dense(x, w, b, act=False)


def _fused(x):
    return x.is_cuda and x.size(-1) == fused.D


def _narrow(x):
    """dim 16 / 32 / 64 on an MI355X: row kernels of csrc/narrow.hip for everything of edge / triplet size."""
    return narrow.supported(x, x.size(-1))


class Act(nn.Module):
    """x * sigmoid(x) (layers/basic.py:11-16); parameter-free placeholder keeping the `.0` / `.1` key structure (the SiLU
    itself is computed in the epilogue of the kernel that runs the enclosing block)."""

    def forward(self, x):
        raise RuntimeError('pamnet_amd: the activation is fused into the dense kernel of its block; call the model / layer')


def MLP(channels):
    blocks = []
    for cin, cout in zip(channels[:-1], channels[1:]):
        blocks.append(nn.Sequential(nn.Linear(cin, cout), Act()))
    return nn.Sequential(*blocks)


class Res(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.mlp = MLP([dim, dim, dim])


class BesselBasis(nn.Module):
    """Holds the trainable frequencies n*pi, n=1..16 (layers/basic.py:65-72)."""

    def __init__(self, num_radial, cutoff, envelope_exponent=5):
        super().__init__()
        self.cutoff = cutoff
        self.envelope_exponent = int(envelope_exponent)
        self.freq = nn.Parameter(torch.arange(1, num_radial + 1, dtype=torch.float32) * math.pi)

    def forward(self, dist, tape=None):
        return ops.rbf(dist, self.freq, self.cutoff, tape=tape, exponent=self.envelope_exponent)


def glorot_(t):
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-bound, bound)


# ---------------------------------------------------------------------------------------------------- dense helpers
def dense(block, x):
    """One `Sequential(Linear, SiLU)` block."""
    lin = block[0]
    return ops.dense(x, lin.weight, lin.bias, act=True)


def mlp_apply(seq, x):
    for block in seq:
        x = dense(block, x)
    return x


def res_apply(res, x):
    return mlp_apply(res.mlp, x) + x


def update_and_heads(layer, x, res_x):
    """Shared tail of both layer kinds (global_message_passing.py:39-50 / local_message_passing.py:55-66)."""
    if _fused(x):
        return fused.node_tail(layer, x, res_x)
    if _narrow(x):
        return narrow.tail(layer, x, res_x)                   # one autograd node: chain + both heads
    x = mlp_apply(layer.mlp_x2, x)
    x = res_apply(layer.res1, x) + res_x
    x = res_apply(layer.res2, x)
    x = res_apply(layer.res3, x)
    o = mlp_apply(layer.mlp_out, x)
    b_out = layer.W_out.bias                                  # both heads as one [N, 2] product: W_out o + b | W^T o
    h = ops.dense(o, torch.cat([layer.W_out.weight, layer.W.t()], 0), torch.cat([b_out, torch.zeros_like(b_out)]))
    return x, h[:, 0].contiguous(), h[:, 1].contiguous()


class _LayerBase(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def _tail_params(self, dim):
        self.res1, self.res2, self.res3 = Res(dim), Res(dim), Res(dim)
        self.mlp_out = MLP([dim, dim, dim, dim])
        self.W_out = nn.Linear(dim, 1)
        self.W = nn.Parameter(torch.empty(dim, 1))
        glorot_(self.W)


class GlobalMP(_LayerBase):
    """layers/global_message_passing.py:9-60."""

    def __init__(self, dim):
        super().__init__(dim)
        self.mlp_x1 = MLP([dim, dim])
        self.mlp_x2 = MLP([dim, dim])
        self.res1, self.res2, self.res3 = Res(dim), Res(dim), Res(dim)
        self.mlp_m = MLP([dim * 3, dim])
        self.W_edge_attr = nn.Linear(dim, dim, bias=False)
        self.mlp_out = MLP([dim, dim, dim, dim])
        self.W_out = nn.Linear(dim, 1)
        self.W = nn.Parameter(torch.empty(dim, 1))
        glorot_(self.W)

    def forward(self, x, e, g):
        d = self.dim
        if _fused(x):
            return fused.global_layer(self, x, e, g)          # whole layer: 4 fused launches forward
        res_x = x
        wm, bm = self.mlp_m[0][0].weight, self.mlp_m[0][0].bias
        if _narrow(x):
            x = narrow.linear(x, self.mlp_x1[0][0])
            p = narrow.project(x, ((0, 0), (0, d)), wm)                          # [N, 2d]: W_i x | W_j x
            x = narrow.global_message(x, p, e, wm, bm, self.W_edge_attr.weight, g.glob, g.glob_T)
            return update_and_heads(self, x, res_x)
        x = mlp_apply(self.mlp_x1, x)
        p = linear(x, torch.cat([wm[:, :d], wm[:, d:2 * d]], 0))                 # [N, 2d]: W_i x | W_j x
        q = linear(e, torch.cat([wm[:, 2 * d:], self.W_edge_attr.weight], 0),   # [E_g, 2d]: W_e e + b | W_ea e
                     torch.cat([bm, torch.zeros_like(bm)]))
        csr = g.glob
        z = ops.gather(p[:, :d], csr.row_of, csr.ptr) + ops.gather(p[:, d:], csr.col, g.glob_T.ptr, g.glob_T.perm) \
            + q[:, :d]
        m = F.silu(z) * q[:, d:]
        x = ops.aggregate(m, csr, init=x)                                         # x + sum_{e -> i} m_e
        return update_and_heads(self, x, res_x)


class LocalMP(_LayerBase):
    """layers/local_message_passing.py:9-66 (and :69-123 with `small=True`: pairs only, `mlp_m_jj`)."""

    def __init__(self, dim, small=False):
        super().__init__(dim)
        self.small = small
        self.mlp_x1 = MLP([dim, dim])
        self.mlp_m_ji = MLP([3 * dim, dim])
        if small:
            self.mlp_m_jj = MLP([3 * dim, dim])
        else:
            self.mlp_m_kj = MLP([3 * dim, dim])
        self.mlp_sbf = MLP([dim, dim, dim])
        self.lin_rbf = nn.Linear(dim, dim, bias=False)
        self.res1, self.res2, self.res3 = Res(dim), Res(dim), Res(dim)
        self.lin_rbf_out = nn.Linear(dim, dim, bias=False)
        self.mlp_x2 = MLP([dim, dim])
        self.mlp_out = MLP([dim, dim, dim, dim])
        self.W_out = nn.Linear(dim, 1)
        self.W = nn.Parameter(torch.empty(dim, 1))
        glorot_(self.W)

    def forward(self, x, rbf, sbf, g):
        d = self.dim
        if _fused(x):
            return fused.local_layer(self, x, rbf, sbf, g)    # whole layer: 6 fused launches forward
        res_x = x
        lin_ji = self.mlp_m_ji[0][0]
        lin_kj = (self.mlp_m_jj if self.small else self.mlp_m_kj)[0][0]
        wj, wk = lin_ji.weight, lin_kj.weight
        # node-level projections [N, 4d]: ji_i | kj_i | ji_j | kj_j ; edge-level [E_l, 4d]: ji_e | kj_e | lin_rbf | lin_rbf_out
        if _narrow(x):
            x = narrow.linear(x, self.mlp_x1[0][0])
            p = narrow.project(x, ((0, 0), (1, 0), (0, d), (1, d)), wj, wk)
        else:
            x = mlp_apply(self.mlp_x1, x)
            p = linear(x, torch.cat([wj[:, :d], wk[:, :d], wj[:, d:2 * d], wk[:, d:2 * d]], 0))
        if _narrow(x):
            q = narrow.project(rbf, ((0, 2 * d), (1, 2 * d), (2, 0), (3, 0)), wj, wk, self.lin_rbf.weight,
                               self.lin_rbf_out.weight)
            zb = True                                         # the projection blocks carry no bias: added with the gates
        else:
            zero, zb = torch.zeros_like(lin_ji.bias), None
            q = linear(rbf, torch.cat([wj[:, 2 * d:], wk[:, 2 * d:], self.lin_rbf.weight, self.lin_rbf_out.weight], 0),
                         torch.cat([lin_ji.bias, lin_kj.bias, zero, zero]))
        csr = g.loc
        if zb is not None:                                    # narrow widths: gathers + gates in one kernel
            m_ji, m_nb = narrow.local_gate(p, q, lin_ji.bias, lin_kj.bias, csr, g.loc_T)
        else:
            z = ops.gather(p[:, :2 * d], csr.row_of, csr.ptr) + ops.gather(p[:, 2 * d:], csr.col, g.loc_T.ptr, g.loc_T.perm) \
                + q[:, :2 * d]
            a = F.silu(z)
            m_ji = a[:, :d]
            m_nb = a[:, d:] * q[:, 2 * d:3 * d]                                   # mlp_m_kj(m) * lin_rbf(rbf)
        s = narrow.mlp2(sbf, self.mlp_sbf) if _narrow(sbf) else mlp_apply(self.mlp_sbf, sbf)   # [T+P, d]
        m_other = ops.gather_mul_aggregate(m_nb, s, g.tp, g.tp_T)                 # -> [E_l, d]
        m = narrow.gate_mul(q, m_ji, m_other) if zb is not None else q[:, 3 * d:] * (m_ji + m_other)
        x = ops.aggregate(m, csr, init=x)
        return update_and_heads(self, x, res_x)

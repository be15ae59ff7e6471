"""Plain-PyTorch statement of the message-passing layers (test infrastructure: the fp32 / fp64 comparand of the kernel tests).

The reference's own formulation -- cat([x_i, x_j, e]) -> Linear(3 dim, dim), torch index ops for gather / scatter-add
(layers/global_message_passing.py:33-56, layers/local_message_passing.py:36-66, 96-123) -- evaluated on the parameters of a
pamnet_amd.modules layer object in whatever dtype / device those are.  Nothing in the product imports this file; until round 6
the same role was played by a module-level switch inside pamnet_amd/modules.py."""
import torch
import torch.nn.functional as F


def mlp_apply(seq, x):
    for block in seq:
        x = F.silu(F.linear(x, block[0].weight, block[0].bias))
    return x


def res_apply(res, x):
    return mlp_apply(res.mlp, x) + x


def update_and_heads(layer, x, res_x):
    """global_message_passing.py:39-50 / local_message_passing.py:55-66."""
    x = mlp_apply(layer.mlp_x2, x)
    x = res_apply(layer.res1, x) + res_x
    x = res_apply(layer.res2, x)
    x = res_apply(layer.res3, x)
    o = mlp_apply(layer.mlp_out, x)
    att = (o @ layer.W).view(-1)
    out = F.linear(o, layer.W_out.weight, layer.W_out.bias).view(-1)
    return x, out, att


def _scatter_add(src, index, rows):
    return torch.zeros(rows, src.size(1), dtype=src.dtype, device=src.device).index_add_(0, index, src)


def global_forward(layer, x, e, g):
    """GlobalMP.forward on the graph object of pamnet_amd.graph (edges in CSR order of their target: row_of = i, col = j)."""
    res_x = x
    x = mlp_apply(layer.mlp_x1, x)
    i, j = g.glob.row_of.long(), g.glob.col.long()
    m = mlp_apply(layer.mlp_m, torch.cat([x[i], x[j], e], 1)) * F.linear(e, layer.W_edge_attr.weight)
    x = x + _scatter_add(m, i, x.size(0))
    return update_and_heads(layer, x, res_x)


def local_forward(layer, x, rbf, sbf, g):
    """LocalMP.forward (triplets and pairs as one combined row list grouped by target edge: tp.row_of = target edge,
    tp.col = the edge whose message is gathered)."""
    res_x = x
    x = mlp_apply(layer.mlp_x1, x)
    i, j = g.loc.row_of.long(), g.loc.col.long()
    m = torch.cat([x[i], x[j], rbf], 1)
    m_ji = mlp_apply(layer.mlp_m_ji, m)
    m_nb = mlp_apply(layer.mlp_m_jj if layer.small else layer.mlp_m_kj, m) * F.linear(rbf, layer.lin_rbf.weight)
    s = mlp_apply(layer.mlp_sbf, sbf)
    m_other = _scatter_add(m_nb[g.tp.col.long()] * s, g.tp.row_of.long(), m.size(0))
    m = F.linear(rbf, layer.lin_rbf_out.weight) * (m_ji + m_other)
    x = x + _scatter_add(m, i, x.size(0))
    return update_and_heads(layer, x, res_x)

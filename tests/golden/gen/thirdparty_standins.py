"""Pure-torch stand-ins for the third-party packages the reference imports but this image lacks.

TEST INFRASTRUCTURE ONLY -- used by tests/golden/gen/gen_golden.py, in the build container, to import the
reference's own models.py / layers/*.py and dump golden vectors.  Nothing in the product imports this file.

These are NOT copies of any reference file: the packages below are un-vendored dependencies pinned in the
reference's requirements.txt:6-11 and absent from /root/reference.  Each stand-in restates the *documented*
behaviour of the pinned version at exactly the call sites the reference uses:

  torch_scatter 2.0.4   scatter(src, index, dim=0, dim_size, reduce='add')       layers/local_message_passing.py:50,54,107,111
  torch_sparse 0.6.0    SparseTensor(row,col,value,sparse_sizes); t[idx]; set_value(None).sum(1); storage.row/col/value
                                                                                  models.py:72-96,267-281
  torch_cluster 1.5.4   radius(x,y,r,batch_x,batch_y,max_num_neighbors), knn(x,y,k,batch_x,batch_y)
                                                                                  models.py:110,128,143,301
  torch_geometric 1.4.2 MessagePassing.propagate, remove_self_loops, global_add_pool, global_mean_pool, inits.glorot
                                                                                  layers/global_message_passing.py:3-4,38; models.py:6-7

Parity note: the reference has no tests, so results at these third-party boundaries are pinned only by
the documented semantics restated here ("parity unpinned" at the third-party level).  Tie handling of radius
(<= r here, as torch_cluster 1.5.4 `dist <= radius`) and neighbour order (ascending index here) are not pinned;
neighbour order only changes fp32 summation order.
"""
import inspect
import math
import sys
import types

import numpy as np
import torch


# ----------------------------------------------------------------------------- torch_scatter
def scatter(src, index, dim=0, out=None, dim_size=None, reduce='add'):
    assert dim == 0 and reduce in ('add', 'sum', 'mean')
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, index, src)
    if reduce == 'mean':
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        res = res / cnt.clamp(min=1).view((-1,) + (1,) * (src.dim() - 1))
    return res


# ----------------------------------------------------------------------------- torch_sparse
class _Storage:
    def __init__(self, row, col, value):
        self._row, self._col, self._value = row, col, value

    def row(self):
        return self._row

    def col(self):
        return self._col

    def value(self):
        return self._value


class SparseTensor:
    """CSR-ordered COO: entries sorted by (row, col) as torch_sparse does on construction."""

    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, _sorted=False):
        n_rows, n_cols = sparse_sizes
        if not _sorted:
            perm = (row * n_cols + col).argsort(stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.sizes = (n_rows, n_cols)
        self.storage = _Storage(row, col, value)
        cnt = torch.bincount(row, minlength=n_rows)
        self.rowptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])

    def __getitem__(self, idx):
        assert idx.dtype == torch.long and idx.dim() == 1
        start = self.rowptr[idx]
        cnt = self.rowptr[idx + 1] - start
        new_row = torch.arange(idx.numel(), device=idx.device).repeat_interleave(cnt)
        off = torch.arange(int(cnt.sum()), device=idx.device) - (cnt.cumsum(0) - cnt).repeat_interleave(cnt)
        src = start.repeat_interleave(cnt) + off
        value = self.storage._value[src] if self.storage._value is not None else None
        return SparseTensor(row=new_row, col=self.storage._col[src], value=value,
                            sparse_sizes=(idx.numel(), self.sizes[1]), _sorted=True)

    def set_value(self, value, layout=None):
        return SparseTensor(row=self.storage._row, col=self.storage._col, value=value,
                            sparse_sizes=self.sizes, _sorted=True)

    def sum(self, dim):
        assert dim == 1
        if self.storage._value is None:
            return (self.rowptr[1:] - self.rowptr[:-1])
        res = torch.zeros(self.sizes[0], dtype=self.storage._value.dtype)
        return res.index_add_(0, self.storage._row, self.storage._value)


# ----------------------------------------------------------------------------- torch_cluster
def _per_graph(batch_x, batch_y):
    nb = int(max(batch_x.max(), batch_y.max())) + 1
    for b in range(nb):
        yield (batch_y == b).nonzero().view(-1), (batch_x == b).nonzero().view(-1)


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    """For each element of y, all points of x (same example) within distance r.  Returns [2,M]=(y idx, x idx)."""
    if batch_x is None:
        batch_x = x.new_zeros(x.size(0), dtype=torch.long)
    if batch_y is None:
        batch_y = y.new_zeros(y.size(0), dtype=torch.long)
    rows, cols = [], []
    for iy, ix in _per_graph(batch_x, batch_y):
        d = (y[iy].double().unsqueeze(1) - x[ix].double().unsqueeze(0)).pow(2).sum(-1).sqrt()
        m = d <= r
        assert int(m.sum(1).max()) <= max_num_neighbors
        q, n = m.nonzero(as_tuple=True)
        rows.append(iy[q])
        cols.append(ix[n])
    return torch.stack([torch.cat(rows), torch.cat(cols)], 0)


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False):
    """For each element of y, its k nearest points of x (same example).  Returns [2,M]=(y idx, x idx)."""
    if batch_x is None:
        batch_x = x.new_zeros(x.size(0), dtype=torch.long)
    if batch_y is None:
        batch_y = y.new_zeros(y.size(0), dtype=torch.long)
    rows, cols = [], []
    for iy, ix in _per_graph(batch_x, batch_y):
        d = (y[iy].double().unsqueeze(1) - x[ix].double().unsqueeze(0)).pow(2).sum(-1)
        kk = min(k, ix.numel())
        nn_idx = d.topk(kk, dim=1, largest=False, sorted=True).indices
        rows.append(iy.repeat_interleave(kk))
        cols.append(ix[nn_idx.reshape(-1)])
    return torch.stack([torch.cat(rows), torch.cat(cols)], 0)


# ----------------------------------------------------------------------------- torch_geometric
def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], (edge_attr[mask] if edge_attr is not None else None)


def global_add_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return scatter(x, batch, dim=0, dim_size=size, reduce='add')


def global_mean_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return scatter(x, batch, dim=0, dim_size=size, reduce='mean')


def glorot(tensor):
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


class MessagePassing(torch.nn.Module):
    """PyG 1.4.2 propagate: (i, j) = (0, 1) for flow='target_to_source' else (1, 0); gathers `<name>_i`,
    `<name>_j` from kwargs[<name>][edge_index[i or j]]; add-aggregates message() at edge_index[i]."""

    def __init__(self, aggr='add', flow='source_to_target', node_dim=0):
        super().__init__()
        assert aggr == 'add' and flow in ('source_to_target', 'target_to_source')
        self.aggr, self.flow = aggr, flow
        self.__msg_args__ = [n for n in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        i, j = (0, 1) if self.flow == 'target_to_source' else (1, 0)
        ij = {'_i': i, '_j': j}
        kwargs['edge_index'] = edge_index
        n_nodes = None
        args = []
        for name in self.__msg_args__:
            if name[-2:] in ij:
                src = kwargs[name[:-2]]
                n_nodes = src.size(0)
                args.append(src.index_select(0, edge_index[ij[name[-2:]]]))
            else:
                args.append(kwargs[name])
        out = self.message(*args)
        out = scatter(out, edge_index[i], dim=0, dim_size=n_nodes, reduce='add')
        return self.update(out)

    def update(self, aggr_out):
        return aggr_out


def install():
    """Register the stand-ins in sys.modules and patch np.math (removed in numpy 2; used by utils/sbf.py:65)."""
    if not hasattr(np, 'math'):
        np.math = math

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('torch_scatter', scatter=scatter)
    mod('torch_sparse', SparseTensor=SparseTensor)
    mod('torch_cluster', radius=radius, knn=knn)
    inits = mod('torch_geometric.nn.inits', glorot=glorot)
    nn = mod('torch_geometric.nn', radius=radius, knn=knn, global_add_pool=global_add_pool,
             global_mean_pool=global_mean_pool, MessagePassing=MessagePassing, inits=inits)
    utils = mod('torch_geometric.utils', remove_self_loops=remove_self_loops)
    mod('torch_geometric', nn=nn, utils=utils)

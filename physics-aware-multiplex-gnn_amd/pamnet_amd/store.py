"""Resident dataset + device-side batch collation ("device-side batch collation + graph construction
with zero host syncs").

The reference's loop builds every batch on the host (PyG `DataLoader` / `Batch.from_data_list`, main_qm9.py:74-77) and
its forward reads data-dependent sizes back from the device (boolean masks, `repeat_interleave`: models.py:62-98).  With
288 GB of HBM the whole dataset stays on the device (QM9: 134 k molecules, ~60 MB), a batch is ONE gather launch over
the concatenated arrays (`pamnet_collate_f32`), and the sizes graph construction needs are per-molecule constants --
the number of atom pairs within the global cutoff, the number of triplet / pair rows of the bond graph -- counted once
per dataset on the device (with the same kernels the forward uses) and summed on the host per batch.  A collated batch
carries them as `batch.sizes`; PAMNet.forward then needs no host round trip at all, and the kernels' own counts are
checked against them on the device (`PAMNet.verify`, called by `Trainer` / `train.predict` when they synchronise
anyway).  The graph itself is still rebuilt every forward from positions and bonds, as models.py:104-177 does: only
integers are cached, no structure.
"""
import numpy as np
import torch

from . import graph as G
from . import lib

I32 = torch.int32


KNN_K = G.KNN_K       # neighbours of the RNA kNN graph (models.py:143): build_graph's default, the only k the sizes are counted for


def size_key(model):
    """Everything the per-graph sizes of a forward depend on: the schema (which graph construction runs), both cutoffs,
    the layer kind (triplets or pairs only), the flow and the kNN neighbour count.  Two models that share a store but
    differ in any of these get their own count tables."""
    ds = model.dataset
    schema = 'rna' if ds[:3].lower() == 'rna' else ds
    return (schema, float(model.cutoff_g), float(model.cutoff_l), not model.small, str(model.flow), KNN_K,
            int(getattr(model, 'max_num_neighbors', 0) or 0))


class Batch(object):
    """Duck-typed `data` of PAMNet.forward: x, pos, edge_index ([2, E] int32), batch (int32), y, num_graphs, sizes."""

    def __init__(self):
        self._pamnet_prepared = None
        self.sizes = None

    def to(self, device):
        return self


class MoleculeStore(object):
    """Dataset resident on one device, any of the three schemas of models.py:104-157.

    data_list: objects / dicts with x ([n] atom types for QM9; [n, w] rows of coordinates + features for PDBbind / RNA),
    optionally pos [n, 3] and edge_index [2, e] (QM9: node ids local to the molecule, both directions present as in
    qm9_dataset.py:243) and y (scalar target)."""

    def __init__(self, data_list, device):
        self.device = torch.device(device)

        class _View(object):                                   # dicts (pamnet_amd.synth) and attribute objects (PyG Data) alike
            def __init__(self, d):
                get = d.get if isinstance(d, dict) else (lambda k, default=None: getattr(d, k, default))
                self.x, self.pos, self.edge_index, self.y = get('x'), get('pos'), get('edge_index'), get('y', None)

        data_list = [_View(d) for d in data_list]
        n = np.array([int(d.x.shape[0]) for d in data_list], dtype=np.int64)
        self.has_edges = data_list[0].edge_index is not None
        self.has_pos = data_list[0].pos is not None
        if self.has_edges:
            # remove_self_loops (models.py:63) once, at ingestion: resident bond lists are loop-free by construction, so
            # a batch never trips the forward's self-loop flag and the per-molecule counts describe what the forward uses
            for d in data_list:
                ei = torch.as_tensor(d.edge_index).to(torch.int64)
                keep = ei[0] != ei[1]
                d.edge_index = ei if bool(keep.all()) else ei[:, keep]
        e = np.array([int(d.edge_index.shape[1]) if self.has_edges else 0 for d in data_list], dtype=np.int64)
        self.n_nodes, self.n_edges = n, e                      # host copies: the batch sizes are sums of these
        self.nptr = np.concatenate([[0], np.cumsum(n)])
        self.eptr = np.concatenate([[0], np.cumsum(e)])
        dev = self.device
        cat = lambda ts, dt: torch.cat([torch.as_tensor(t).reshape(-1, *torch.as_tensor(t).shape[1:]) for t in ts]).to(dt)
        x0 = torch.as_tensor(data_list[0].x)
        self.x_width = 1 if x0.dim() == 1 else int(x0.shape[1])
        self.x = cat([d.x for d in data_list], torch.float32).reshape(-1, self.x_width).contiguous().to(dev)
        self.pos = cat([d.pos for d in data_list], torch.float32).contiguous().to(dev) if self.has_pos else None
        if self.has_edges:
            ei = torch.cat([torch.as_tensor(d.edge_index).to(torch.int64) for d in data_list], dim=1)
            self.esrc = ei[0].to(I32).contiguous().to(dev)     # graph-local endpoints
            self.edst = ei[1].to(I32).contiguous().to(dev)
        else:
            self.esrc = self.edst = None
        ys = [getattr(d, 'y', None) for d in data_list]
        self.y = None if any(v is None for v in ys) else torch.as_tensor(
            np.array([float(torch.as_tensor(v).reshape(-1)[0]) for v in ys], dtype=np.float32)).to(dev)
        self.nptr_d = torch.from_numpy(self.nptr.astype(np.int32)).to(dev)
        self.eptr_d = torch.from_numpy(self.eptr.astype(np.int32)).to(dev)
        self._counts = {}                  # size_key(model) -> per-graph (E_g, E_l, T+P) arrays
        self._mol_local = {}               # size_key(model) -> every molecule fits the molecule-local graph builder
        self._capped = {}                  # size_key(model) -> max_num_neighbors binds in this dataset (no sizes handed out)

    def __len__(self):
        return len(self.n_nodes)

    _RING = 8

    def _staging(self, ints):
        """Pinned staging for a batch's selection + prefix sums: a ring of buffers allocated once (a pinned allocation per
        batch cost ~30 us of host time; the ring is deeper than the trainer's two steps in flight, and a slot is only
        reused after the copy queued from it has completed)."""
        ring = self.__dict__.setdefault('_ring', [])
        k = self.__dict__.get('_ring_at', 0)
        self.__dict__['_ring_at'] = (k + 1) % self._RING
        if len(ring) <= k:
            ring.append([None, None, None])
        slot = ring[k]
        if slot[0] is None or slot[0].numel() < ints:
            slot[0] = torch.empty(max(int(ints), 4096), dtype=I32, pin_memory=True)
            slot[1] = slot[0].numpy()
            slot[2] = None
        if slot[2] is not None:
            slot[2].synchronize()                              # the upload that last read this slot (long done)
        slot[2] = torch.cuda.Event()
        self._last_slot = slot
        return slot[0][:ints], slot[1][:ints]

    # -------------------------------------------------------------------------------------------------- per-molecule sizes
    def counts_for(self, model, chunk=None):
        """Per-graph (global edges, local edges, triplet + pair rows) for `model` (its dataset schema, cutoffs, flow and
        layer kind): counted on the device by the forward's own graph construction, `chunk` graphs per pass, read back
        ONCE per dataset.  The local edges of a graph are contiguous in the batch's CSR (sorted by target node), so every
        count is a difference of CSR pointers at the graph's node range."""
        key = size_key(model)
        if key in self._counts:
            return self._counts[key]
        m = len(self)
        if chunk is None:
            chunk = 4096 if self.n_nodes.mean() < 100 else 8
        eg, el, tp = (np.zeros(m, dtype=np.int64) for _ in range(3))
        capped = False                     # max_num_neighbors binds somewhere in this dataset (graph.build_graph)
        for a in range(0, m, chunk):
            b = min(m, a + chunk)
            bt = self.collate(np.arange(a, b), with_sizes=False)
            g = G.build_graph(model.dataset, model.cutoff_l, model.cutoff_g, model.flow, bt.x, bt.batch, bt.pos, bt.edge_index,
                              num_graphs=b - a, need_grad=False, with_triplets=not model.small,
                              n_types=model.embeddings.size(0) if hasattr(model, 'embeddings') else None,
                              max_num_neighbors=getattr(model, 'max_num_neighbors', None))
            capped = capped or bool(getattr(g, 'capped', False))
            nodes = torch.from_numpy((self.nptr[a:b + 1] - self.nptr[a]).astype(np.int64)).to(self.device)
            pg, pl = g.glob.ptr.long()[nodes], g.loc.ptr.long()[nodes]
            pt = g.tp.ptr.long()[pl]
            eg[a:b] = (pg[1:] - pg[:-1]).cpu().numpy()
            el[a:b] = (pl[1:] - pl[:-1]).cpu().numpy()
            tp[a:b] = (pt[1:] - pt[:-1]).cpu().numpy()
        self._counts[key] = (eg, el, tp)
        # A dataset in which the radius search's neighbour cap binds is handed over without sizes: its capped global graphs are
        # not symmetric, which the one-call graph assumes -- those batches take the plain path (one host round trip each).
        self._capped[key] = capped
        # QM9 schema: every molecule inside the molecule-local graph builder's limits (csrc/graph_mol.hip)?  Collation keeps a
        # molecule's bonds together and in batch order, self loops were stripped at ingestion: the rest of its contract.
        self._mol_local[key] = bool(key[0] == 'QM9' and m > 0 and self.n_nodes.max() <= G.MOL_ATOMS
                                    and el.max() <= G.MOL_BONDS)
        return eg, el, tp

    def prepare_for(self, *models):
        """Count the sizes the given models' forwards need (one device pass per distinct cutoff / layer kind)."""
        for mdl in models:
            self.counts_for(mdl)
        return self

    # ------------------------------------------------------------------------------------------------------- collation
    def collate(self, idx, with_sizes=True):
        """Batch of the molecules `idx` (host integers, any order): one gather launch on the device, no host sync.  The
        prefix sums of the batch are computed on the host from the per-molecule counts and travel with `idx` in one
        small asynchronous copy."""
        idx = np.asarray(idx, dtype=np.int64)
        b = int(idx.size)
        nn, ne = self.n_nodes[idx], self.n_edges[idx]
        n_out, e_out = int(nn.sum()), int(ne.sum())
        # pinned staging: the upload is a true asynchronous copy (from pageable memory the runtime parks the host
        # behind everything already queued on the stream -- the whole previous step)
        meta_h, meta = self._staging(3 * b + 2)
        meta[:b] = idx
        meta[b] = 0
        np.cumsum(nn, out=meta[b + 1:2 * b + 1])
        meta[2 * b + 1] = 0
        np.cumsum(ne, out=meta[2 * b + 2:3 * b + 2])
        dev = self.device
        meta_d = meta_h.to(dev, non_blocking=True)
        self._last_slot[2].record(torch.cuda.current_stream(dev))
        sel, out_nptr, out_eptr = meta_d[:b], meta_d[b:2 * b + 1], meta_d[2 * b + 1:]
        bt = Batch()
        w = self.x_width
        bt.x = torch.empty(n_out if w == 1 else (n_out, w), dtype=torch.float32, device=dev)
        bt.pos = torch.empty((n_out, 3), dtype=torch.float32, device=dev) if self.has_pos else None
        bt.batch = torch.empty(n_out, dtype=I32, device=dev)
        bt.edge_index = torch.empty((2, e_out), dtype=I32, device=dev) if self.has_edges else None
        bt.y = None if self.y is None else torch.empty(b, dtype=torch.float32, device=dev)
        lib.call('pamnet_collate_f32', b, lib.ptr(sel), lib.ptr(out_nptr), lib.ptr(out_eptr), lib.ptr(self.nptr_d),
                 lib.ptr(self.eptr_d), lib.ptr(self.x), w, lib.ptr(self.pos), lib.ptr(self.esrc), lib.ptr(self.edst), n_out,
                 e_out, lib.ptr(bt.x), lib.ptr(bt.pos), lib.ptr(bt.batch),
                 bt.edge_index.data_ptr() if e_out else None, (bt.edge_index.data_ptr() + 4 * e_out) if e_out else None,
                 lib.ptr(self.y), lib.ptr(bt.y), lib.stream_of(bt.x))       # (the targets ride in the same launch)
        bt.num_graphs = b
        # the batch is produced by work queued on this stream: a consumer on another stream (the input pipeline's side
        # stream, train.Prefetcher) waits for exactly this event
        bt.inputs_ready = torch.cuda.Event()
        bt.inputs_ready.record(torch.cuda.current_stream(dev))
        if with_sizes:
            bt.sizes = {key: (int(eg[idx].sum()), int(el[idx].sum()), int(tp[idx].sum()))
                        for key, (eg, el, tp) in self._counts.items() if not self._capped.get(key)}
            bt.mol_local = self._mol_local
        return bt

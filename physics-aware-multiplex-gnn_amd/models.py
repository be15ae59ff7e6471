"""Drop-in for the reference's `models` module: `from models import PAMNet, PAMNet_s, Config`
(reference main_qm9.py:13, main_pdbbind.py, main_rna_puzzles.py, inference_rna_puzzles.py:10).

Same constructor arguments, attribute names, `state_dict()` keys/shapes and `forward(data)` contract as the reference
(models.py:12-224, 227-353), but the forward runs on hand-written gfx950 kernels through the C ABI of libpamnet_hip.so
(include/pamnet_hip.h).  Put this directory on sys.path in place of the reference checkout's root.

forward(data): duck-typed `data` with `.x`, `.batch` (sorted) and, for QM9, `.pos` [N,3] and `.edge_index` [2,E]
(optionally `.num_graphs`).  Returns fp32 [num_graphs], differentiable w.r.t. every parameter.  MI355X only: tensors must
live on a HIP device -- there is no CPU path (the CPU oracle in oracle/ is test infrastructure and is never imported
from here).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from pamnet_amd import fused, modules, narrow
from pamnet_amd import graph as G
from pamnet_amd import ops
from pamnet_amd.modules import MLP, BesselBasis, GlobalMP, LocalMP, mlp_apply


class Config(object):
    def __init__(self, dataset, dim, n_layer, cutoff_l, cutoff_g, flow='source_to_target'):
        self.dataset = dataset
        self.dim = dim
        self.n_layer = n_layer
        self.cutoff_l = cutoff_l
        self.cutoff_g = cutoff_g
        self.flow = flow


def _sph_jn(l, x):
    """Spherical Bessel j_l(x), float64, for the host-side table construction: power series below the turning point,
    upward recurrence above it (the same split the kernels use, csrc/basis.hip)."""
    if x < l + 1.0:
        h, term, total = -0.5 * x * x, 1.0, 1.0
        for k in range(1, 60):
            term *= h / (k * (2 * l + 2 * k + 1))
            total += term
        pref = 1.0
        for m in range(1, l + 1):
            pref *= x / (2 * m + 1)
        return pref * total
    jm = math.sin(x) / x
    if l == 0:
        return jm
    j = (jm - math.cos(x)) / x
    for m in range(1, l):
        jm, j = j, (2 * m + 1) / x * j - jm
    return j


def basis_tables(num_spherical, num_radial):
    """(zeros float32 [ns, nr], normalisers float64 [ns, nr]) of the spherical-Bessel basis: the first nr positive zeros of
    j_0 .. j_{ns-1}, ROUNDED TO FLOAT32 as utils/sbf.py:15-26 stores them (zeros of j_l interlace those of j_{l-1}: the
    same brackets as the reference's brentq sweep, bisected to the last float64 bit), and N_ln = 1 / sqrt(0.5 j_{l+1}(z_ln)^2)
    evaluated on the rounded zeros (utils/sbf.py:43-49).  Pure float64 arithmetic: no scipy / sympy at run time (the
    reference spends 12-16 s of sympy here per model)."""
    import numpy as np
    ns, nr = int(num_spherical), int(num_radial)
    zeros = np.zeros((ns, nr), dtype=np.float64)
    zeros[0] = np.arange(1, nr + 1) * math.pi
    points = [k * math.pi for k in range(1, nr + ns)]
    for l in range(1, ns):
        roots = []
        for j in range(nr + ns - 1 - l):
            a, b = points[j], points[j + 1]
            fa = _sph_jn(l, a)
            for _ in range(200):
                mid = 0.5 * (a + b)
                if mid == a or mid == b:
                    break
                fm = _sph_jn(l, mid)
                if (fm > 0) == (fa > 0):
                    a, fa = mid, fm
                else:
                    b = mid
            roots.append(0.5 * (a + b))
        points = roots
        zeros[l] = roots[:nr]
    z32 = zeros.astype(np.float32)
    z = z32.astype(np.float64)
    norm = np.array([[1.0 / math.sqrt(0.5 * _sph_jn(l + 1, z[l, i]) ** 2) for i in range(nr)] for l in range(ns)])
    return z32, norm


class SphericalBasis(nn.Module):
    """Parameter-free stand-in for `self.sbf` (layers/basic.py:79-116).  The reference spends 12-16 s of sympy here.  The
    default sizes (7, 6, envelope exponent 5 -- what every script of the reference constructs) ship as compile-time
    constants inside the library (csrc/basis_constants.h); any other (num_spherical <= 16, num_radial <= 64,
    envelope_exponent >= 1) gets its zeros / normalisers computed here once and handed to the table-driven kernels
    (pamnet_sbf_radial_tab_f32 / pamnet_sbf_combine_tab_f32)."""

    def __init__(self, num_spherical, num_radial, cutoff, envelope_exponent):
        super().__init__()
        ns, nr, p = int(num_spherical), int(num_radial), int(envelope_exponent)
        if not (1 <= ns <= 16 and 1 <= nr <= 64 and 1 <= p <= 64):
            raise ValueError('spherical basis: num_spherical in 1..16, num_radial in 1..64 (layers/basic.py:82), '
                             'envelope_exponent in 1..64')
        self.cutoff = cutoff
        self.num_spherical, self.num_radial, self.envelope_exponent = ns, nr, p
        self.default = (ns, nr, p) == (7, 6, 5)
        if not self.default:
            z32, norm = basis_tables(ns, nr)
            # Constants of the basis, not module state: host masters (float32 zeros / float64 normalisers, exactly the types
            # the table-driven kernels read through raw pointers) plus one device copy per device, made on first use.  They are
            # NOT buffers: `model.half()` / `.double()` / `.to(dtype)` must not recast them (a recast table would be read
            # with the wrong element size), and `state_dict()` keeps the reference's keys (its basis holds no tensors).
            self.__dict__['_host_tables'] = (torch.from_numpy(z32).reshape(-1).contiguous(),
                                             torch.from_numpy(norm).reshape(-1).contiguous())
            self.__dict__['_device_tables'] = {}

    def tables(self, device):
        """(zeros float32 [ns*nr], normalisers float64 [ns*nr]) resident on `device`."""
        t = self._device_tables.get(device)
        if t is None:
            z, n = self._host_tables
            t = self._device_tables[device] = (z.to(device), n.to(device))
        return t

    def forward(self, graph):
        if self.default:
            return G.spherical_basis(graph, self.cutoff)
        zeros, norm = self.tables(graph.dist_l.device)
        return G.spherical_basis_tab(graph, self.cutoff, self.num_spherical, self.num_radial, self.envelope_exponent,
                                     zeros, norm)


class _LazyLayers(object):
    """x after every layer, materialised from the engine's saved-activation arena only when somebody looks."""

    def __init__(self, saved, graph, n_layer, dim=None):
        self._args = (saved, graph, n_layer)
        self._dim = dim

    def _views(self):
        if self._dim is not None:
            return narrow.stack_x_layers(*self._args, self._dim)
        return fused.stack_x_layers(*self._args)

    def __iter__(self):
        return iter(self._views())

    def __len__(self):
        return 2 * self._args[2]

    def __getitem__(self, i):
        return self._views()[i]


ENGINE_WIDTHS = (16, 32, 64, 128)       # widths with hand-written kernel families (narrow engine: 16 / 32 / 64; fused: 128)


FLAT_NAME = 'flat_parameters'


class _FlatView(object):
    """State of the PAMNET_FLAT_PARAMS=1 interface of one model: the flat buffers (train.FlatParams, gradients written in
    place by the kernels) and the ONE nn.Parameter handed to the caller's optimiser."""

    def __init__(self, model):
        from pamnet_amd.train import FlatParams
        self.fp = FlatParams(model, direct=True)
        self.param = nn.Parameter(self.fp.flat)           # shares the buffer's storage
        self.param.grad = self.fp.grad
        self._pending = None

    def valid(self):
        """Every parameter is still a view of the buffer (model.to(...) / .cpu() re-allocate them: rebuilt on next use)."""
        fp = self.fp
        first, last = fp.params[0], fp.params[-1]
        return (first.is_cuda and first.data_ptr() == fp.flat.data_ptr() + 4 * fp.offsets[fp.names[0]]
                and last.data_ptr() == fp.flat.data_ptr() + 4 * fp.offsets[fp.names[-1]])

    def before_forward(self):
        """The kernels OVERWRITE their gradients and a backward starts from a zeroed buffer.  optimizer.zero_grad() (torch
        >= 2.0: set_to_none) left .grad = None: zero the buffer, nothing else.  A .grad that is still there may hold gradients
        the caller wants accumulated into (or zeros: zero_grad(set_to_none=False)): set aside, added back after the backward."""
        if self._pending is not None:                     # the previous forward was never differentiated: undo its zeroing
            if self._pending[0] == 'own':
                self.fp.grad.copy_(self._pending[1])
            self._pending = None
        g = self.param.grad
        if g is None:
            self._pending = ('none', None)
        elif g.data_ptr() == self.fp.grad.data_ptr():
            self._pending = ('own', g.clone())
        else:
            self._pending = ('other', g)
        self.fp.grad.zero_()

    def after_backward(self):
        if self._pending is not None and self._pending[0] != 'none':
            self.fp.grad.add_(self._pending[1].view(-1))
        self._pending = None
        self.param.grad = self.fp.grad


def _rng_snapshot():
    """(cpu_state, [device states]) of torch's global generators; the device part only when a HIP context exists already
    (asking for it would create one)."""
    dev = None
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        dev = [torch.cuda.get_rng_state(i) for i in range(torch.cuda.device_count())]
    return torch.get_rng_state(), dev


def _rng_restore(snap):
    cpu, dev = snap
    torch.set_rng_state(cpu)
    if dev is not None:
        for i, st in enumerate(dev):
            torch.cuda.set_rng_state(st, i)


def engine_width(dim):
    """The kernel width a model of hidden size `dim` runs at: the smallest engine width >= dim; above 128 (the layer-by-layer
    path on csrc/dense.hip) the next multiple of 4 -- 16-byte vector lanes of the gather / segment kernels -- with the same
    zero-padding (models.py:25 takes any dim)."""
    for w in ENGINE_WIDTHS:
        if dim <= w:
            return w
    return (int(dim) + 3) // 4 * 4


class _PAMNetBase(nn.Module):
    """Any `config.dim` (models.py:25: the reference takes whatever it is given).  Widths 16 / 32 / 64 / 128 have kernel
    families of their own; every other dim <= 128 runs on the next one up with ZERO-PADDED parameters: the model is built at
    the engine width `self.dim`, the padded rows / columns of every weight, bias and embedding are zero, and they stay
    exactly zero under training -- a padded output channel is SiLU(0) = 0 and 0 * gate = 0 in every message, so its
    activations, its gradients (dz = dy * SiLU'(0) with dy = W_next[:, pad]^T dz = 0; dW[:, pad] = dz x_pad = 0) and hence
    its Adam / EMA updates are zeros: arithmetic on the logical `config_dim` channels is untouched.  `state_dict()` /
    `load_state_dict()` speak the reference's shapes (hooks below slice / pad); `parameters()` are the padded tensors.
    Above 128 every dense layer is a launch of the any-width GEMM kernels of csrc/dense.hip (bf16x6 on the matrix pipe, no
    library GEMM) between the HIP graph / basis / gather / segment kernels; such a dim is padded to the next multiple of 4."""
    small = False
    max_num_neighbors = 1000            # radius(..., max_num_neighbors=1000), models.py:110,128

    def __init__(self, config, num_spherical=7, num_radial=6, envelope_exponent=5, _pad=True):
        super().__init__()
        self.dataset = config.dataset
        self.config_dim = int(config.dim)
        w = engine_width(self.config_dim) if _pad else None
        self.dim = w if w is not None else self.config_dim           # the width the kernels run at
        self.n_layer = config.n_layer
        self.cutoff_l = config.cutoff_l
        self.cutoff_g = config.cutoff_g
        self.flow = getattr(config, 'flow', 'source_to_target')
        if self.config_dim < 1:
            raise ValueError('dim must be positive')
        # (RNG state at construction: _finish_padding rewinds to it so that a padded model draws what the unpadded one would)
        # CPU generator and, when a HIP context is live, the generators of every visible device: under
        # torch.set_default_device('cuda') / a device context the parameters are drawn from the device's generator.
        # Caveat: draws a SUBCLASS constructor makes between this point and _finish_padding are replayed as well.
        self.__dict__['_rng_at_ctor'] = _rng_snapshot() if (_pad and self.dim != self.config_dim) else None
        self._rna = self.dataset[:3].lower() == 'rna'
        self.__dict__['_pending_checks'] = []            # device flag words of forwards that ran without a host round trip
        self.__dict__['_ctor'] = (config, num_spherical, num_radial, envelope_exponent)

    # ---- parameter walks -------------------------------------------------------------------------------------------
    # `model.parameters()` / `named_parameters()` walk ~250 sub-modules for ~390 tensors: 1.4 ms of host time per walk, and
    # the reference's loop walks the tree three times per step (clip_grad_norm_(model.parameters()), utils.EMA, and the
    # optimiser at construction) -- a third of its 8-9 ms step on this path.  The model's structure is fixed after
    # construction, so the full walk is cached; anything that could change which Parameter objects hang off the tree
    # (attribute assignment on the model, `_apply` under the overwrite-on-conversion future flag, adding modules)
    # drops the cache.
    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        if prefix != '' or not recurse or not remove_duplicate:
            yield from super().named_parameters(prefix, recurse, remove_duplicate)
            return
        flat = self._flat_view()
        if flat is not None:                             # PAMNET_FLAT_PARAMS=1: ONE parameter for optimiser / clip / EMA
            yield (FLAT_NAME, flat.param)
            return
        yield from self._real_named_parameters()

    def _real_named_parameters(self):
        """The ~390 reference-named parameters (the cached walk); what state_dict() and every kernel work on."""
        cache = self.__dict__.get('_named_param_cache')
        if cache is not None and not self.__dict__.get('_params_checked') and not self._param_cache_valid(cache):
            self._drop_param_cache()                     # (the derived lists go with it)
            cache = None
        if cache is None:
            cache = list(super().named_parameters('', True, True))
            mods = dict(self.named_modules())
            self.__dict__['_named_param_cache'] = cache
            self.__dict__['_named_param_owners'] = [(mods[n.rpartition('.')[0]], n.rpartition('.')[2]) for n, _ in cache]
            # the live tree the walk saw: every module's link from its parent and its own entry counts
            self.__dict__['_named_param_links'] = (
                [(mods[n.rpartition('.')[0]], n.rpartition('.')[2], m, len(m._parameters), len(m._modules))
                 for n, m in mods.items() if n != ''], len(self._parameters), len(self._modules))
        yield from cache

    def _param_cache_valid(self, cache):
        """A sub-module cannot tell the root that something was replaced, added or removed inside it, so the cache is checked
        against the LIVE tree before it is used (~250 modules + ~390 tensors of dictionary look-ups: ~0.1 ms, not a walk):
        every module must still hang off its parent under its name (a replaced module -- `layer.W_out = nn.Linear(...)` --
        still holds its own old tensors, so checking owners alone is not enough: ADVICE r4), hold as many parameters and
        children as when the walk saw it (register_parameter / add_module inside a child), and every cached tensor must
        still be what its owner holds under that name."""
        links, n_par, n_mod = self.__dict__['_named_param_links']
        if len(self._parameters) != n_par or len(self._modules) != n_mod:
            return False
        for parent, leaf, m, np_, nm_ in links:
            if parent._modules.get(leaf) is not m or len(m._parameters) != np_ or len(m._modules) != nm_:
                return False
        owners = self.__dict__['_named_param_owners']
        return all(o._parameters.get(leaf) is p for (o, leaf), (_, p) in zip(owners, cache))

    def _drop_param_cache(self):
        self.__dict__.pop('_named_param_cache', None)
        self.__dict__.pop('_named_param_owners', None)
        self.__dict__.pop('_named_param_links', None)
        self.__dict__.pop('_all_param_list', None)
        self.__dict__.pop('_top_param_list', None)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Parameter, nn.Module)) or name in self.__dict__.get('_parameters', ()):
            self._drop_param_cache()
        super().__setattr__(name, value)

    def register_parameter(self, name, param):
        self._drop_param_cache()
        super().register_parameter(name, param)

    def add_module(self, name, module):
        self._drop_param_cache()
        super().add_module(name, module)

    def _apply(self, fn, recurse=True):
        self._drop_param_cache()
        out = super()._apply(fn, recurse)
        self._drop_param_cache()
        return out

    # ---- PAMNET_FLAT_PARAMS=1: the reference's loop, unchanged, at a fraction of its host cost -------------------------
    # main_qm9.py:90-116 hands `model.parameters()` to torch.optim.Adam, clip_grad_norm_ and utils.EMA.  Those are elementwise
    # (Adam, EMA) or one global norm (clip): over ~390 tensors they cost the HOST 3-4 ms per step (multi-tensor launches,
    # 390 gradient accumulators in the autograd graph) against a 2 ms step.  With the switch on, parameters() /
    # named_parameters() yield ONE nn.Parameter -- a flat buffer every reference-named parameter is a view of
    # (pamnet_amd.train.FlatParams; 256-byte aligned, zero padding that stays zero under Adam) -- with the flat gradient the
    # kernels write in place as its .grad.  The same numbers come out (Adam / EMA element by element; the norm over the
    # same elements), state_dict() / load_state_dict() still speak the ~390 reference keys (they walk the modules, not
    # parameters()), and the whole forward is ONE autograd node.  Opt-in because two things differ from a plain
    # nn.Module: `sum(p.numel() for p in model.parameters())` counts the padding, and per-tensor options
    # (parameter groups by name) have nothing to hold on to.
    def _flat_view(self):
        if os.environ.get('PAMNET_FLAT_PARAMS', '0') != '1' or self.__dict__.get('_flat_view_off'):
            return None
        st = self.__dict__.get('_flat_view_state')
        if st is not None and st.valid():
            return st
        self.__dict__.pop('_flat_view_state', None)
        real = [p for _, p in self._real_named_parameters()]
        if not real or not real[0].is_cuda or real[0].dtype != torch.float32 or not all(p.requires_grad for p in real):
            return None
        if not (self.dim == fused.D or (self.dim in narrow.WIDTHS and narrow.ENABLED)):
            return None                                   # (widths without a one-node forward keep the per-tensor interface)
        st = self.__dict__['_flat_view_state'] = _FlatView(self)
        return st

    def _disable_flat_view(self):
        """pamnet_amd.train.Trainer re-homes the parameters in flat buffers of its own."""
        self.__dict__['_flat_view_off'] = True
        self.__dict__.pop('_flat_view_state', None)

    def _run_one_node(self, run):
        """The forward as ONE autograd node (ops.run_whole) on the flat-view parameter when there is one."""
        st = self.__dict__.get('_flat_view_state')
        if st is None or not st.valid():
            return ops.run_whole(self.rbf_g.freq, run)
        return ops.run_whole(st.param, run, before=st.before_forward, after=st.after_backward)

    def _finish_padding(self):
        """Called at the end of the subclass constructors.  A model whose engine width differs from its configured dim gets
        (a) the reference's shapes for the state_dict interface, taken from an unpadded twin, (b) that twin's initial
        values (the same initialisation law on the LOGICAL fan-in / fan-out) in the top-left blocks, zeros elsewhere."""
        if self.dim == self.config_dim:
            return
        # The twin takes the FIRST draw of the random stream -- exactly what a model of the reference's shapes draws from this
        # seed (the padded model's own init() above drew values that are overwritten below): seed-for-seed the same initial
        # weights as at an engine width, and the stream is left where the unpadded construction leaves it.
        if self.__dict__.get('_rng_at_ctor') is not None:
            _rng_restore(self.__dict__.pop('_rng_at_ctor'))
        base = PAMNet_s if self.small else PAMNet         # (explicit class: a subclass may have another constructor)
        twin = base(*self._ctor, _pad=False)
        tsd = twin.state_dict()
        self.__dict__['_logical_shapes'] = {k: tuple(v.shape) for k, v in tsd.items()}
        self._register_load_state_dict_pre_hook(self._pad_incoming)
        self._register_state_dict_hook(_slice_outgoing)
        self.load_state_dict(tsd, strict=True)

    def _pad_index(self, logical, padded):
        """Per dimension: where the logical entries sit in the padded tensor.  A dimension is either untouched (16 Bessel /
        42 spherical / 18 feature inputs, 1-wide heads), one channel block (dim -> engine width), or the three channel
        blocks [x_i | x_j | e] of the message MLPs' inputs (3 dim -> 3 width: block k starts at k * width, models.py
        global_message_passing.py:55, local_message_passing.py:46)."""
        d, w = self.config_dim, self.dim
        out = []
        for n, m in zip(logical, padded):
            if n == m:
                out.append(None)
            else:
                nb = n // d
                assert nb * d == n and nb * w == m, (logical, padded)
                out.append(torch.cat([torch.arange(k * w, k * w + d) for k in range(nb)]))
        return out

    def logical_mask(self, name, like=None):
        """Bool tensor of parameter `name`'s padded shape: True where a logical (reference-shaped) entry lives."""
        p = dict(self._real_named_parameters())[name] if like is None else like
        shape = self.__dict__.get('_logical_shapes', {}).get(name, tuple(p.shape))
        mask = torch.zeros(p.shape, dtype=torch.bool, device=p.device)
        idx = self._pad_index(shape, tuple(p.shape))
        grid = [torch.arange(n, device=p.device) if i is None else i.to(p.device) for n, i in zip(shape, idx)]
        if grid:
            mask[tuple(torch.meshgrid(*grid, indexing='ij'))] = True
        else:
            mask[...] = True
        return mask

    def _pad_incoming(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        own = dict(self._real_named_parameters())
        own.update(dict(self.named_buffers()))
        for k, shape in self._logical_shapes.items():
            t = state_dict.get(prefix + k)
            if t is None or k not in own or tuple(t.shape) != shape or tuple(own[k].shape) == shape:
                continue                                  # (absent, or already at the padded shape)
            full = torch.zeros(own[k].shape, dtype=t.dtype, device=t.device)
            idx = [None if i is None else i.to(t.device) for i in self._pad_index(shape, tuple(own[k].shape))]
            grid = [torch.arange(n, device=t.device) if i is None else i for n, i in zip(shape, idx)]
            full[tuple(torch.meshgrid(*grid, indexing='ij'))] = t
            state_dict[prefix + k] = full

    def _build_common(self, num_spherical, num_radial, envelope_exponent):
        d = self.dim
        self.rbf_g = BesselBasis(16, self.cutoff_g, envelope_exponent)          # models.py:26-28
        self.rbf_l = BesselBasis(16, self.cutoff_l, envelope_exponent)
        self.sbf = SphericalBasis(num_spherical, num_radial, self.cutoff_l, envelope_exponent)
        self.mlp_rbf_g = MLP([16, d])
        self.mlp_rbf_l = MLP([16, d])

    def init(self):
        stdv = math.sqrt(3)
        self.embeddings.data.uniform_(-stdv, stdv)

    # ------------------------------------------------------------------------------------------------------------
    def _graph(self, data):
        ev = getattr(data, 'inputs_ready', None)
        if ev is not None and data.batch.is_cuda:
            # a batch produced on another stream (asynchronous upload, device-side collation) says so: this stream waits for
            # exactly that event (a no-op when it is the producing stream)
            torch.cuda.current_stream(data.batch.device).wait_event(ev)
        pre = getattr(data, '_pamnet_prepared', None)
        if pre is not None:
            data._pamnet_prepared = None                 # a prepared graph serves exactly one forward: nothing is cached
            if pre.need_grad or not torch.is_grad_enabled():
                return pre                               # built ahead of time by prepare() (possibly on a side stream)
        ng = getattr(data, 'num_graphs', None)
        g = G.build_graph(self.dataset, self.cutoff_l, self.cutoff_g, self.flow, data.x, data.batch,
                          getattr(data, 'pos', None), getattr(data, 'edge_index', None), num_graphs=ng,
                          need_grad=torch.is_grad_enabled(), with_triplets=not self.small,
                          n_types=self.embeddings.size(0) if hasattr(self, 'embeddings') else None,
                          sizes=self._sizes_of(data), default_basis=self.sbf.default, mol_local=self._mol_local_of(data),
                          max_num_neighbors=self.max_num_neighbors, aux_tables=self.dim == fused.D)
        if g.check is not None:                          # zero-host-sync path: the flag word waits for verify()
            self._pending_checks.append(g.check)
        g.need_grad = torch.is_grad_enabled()
        g.sbf = self.sbf(g)                              # [T+P, 42]; geometry only, no parameters
        return g

    def _sizes_of(self, data):
        """Host-side sizes of a batch collated by pamnet_amd.store.MoleculeStore: (global edges, local edges, triplet +
        pair rows) for THIS model's cutoffs / layer kind, or None (sizes are then read back from the device)."""
        sz = getattr(data, 'sizes', None)
        if sz is None:
            return None
        if isinstance(sz, dict):
            from pamnet_amd.store import size_key
            return sz.get(size_key(self))
        return sz

    def _mol_local_of(self, data):
        """True: a store vouches that this batch is inside the molecule-local graph builder's contract (graph.build_graph);
        None: unknown (plain tensors -- found out on the device); False: no."""
        ml = getattr(data, 'mol_local', None)
        if isinstance(ml, dict):
            from pamnet_amd.store import size_key
            return ml.get(size_key(self))
        return ml

    def verify(self):
        """Check the device-side flag words of the forwards since the last call that ran without a host round trip
        (batches carrying `sizes`): one readback.  Raises IndexError (invalid index inputs, as the reference would have)
        or graph.GraphCheckError (sizes that do not belong to the batch).  Call it wherever the host synchronises anyway
        -- train.predict does; train.Trainer takes the flag words over step by step (take_pending_checks) and reads each
        step's own words once that step has completed."""
        pend = self.take_pending_checks()
        if pend:
            G.raise_for_flag(G.read_flags(pend))

    def take_pending_checks(self):
        """Hand the flag words of the forwards since the last verify() / take to the caller, who then owns checking them
        (graph.read_flags + graph.raise_for_flag) once the forwards that wrote them have completed."""
        pend = self._pending_checks
        self.__dict__['_pending_checks'] = []
        return pend

    def prepare(self, data, need_grad=True):
        """Parameter-independent part of forward(data): graph construction (models.py:104-177) and the spherical
        basis.  It depends on the batch only, so an input pipeline can run it ahead of time -- e.g. on a side stream
        while the previous step is still executing (pamnet_amd.train.Trainer.step(..., next_data=...)).  The result
        is attached to `data` and picked up by forward()."""
        self._check_dataset(params=False)                 # (no parameter is read here: forward() checks their dtype)
        with torch.set_grad_enabled(need_grad):
            data._pamnet_prepared = None
            data._pamnet_prepared = self._graph(data)
        return data

    def _embed(self, data, g, tape=None):
        x_raw = data.x
        if self.dataset == 'PDBbind':
            xr = x_raw.unsqueeze(-1) if x_raw.dim() == 1 else x_raw
            feats = xr[:, 3:].to(torch.float32)
            if fused.embed_supported(feats, self.init_linear):
                return fused.embed(feats, self.init_linear, act=False, tape=tape)          # models.py:119
            return ops.plain_linear(feats.contiguous(), self.init_linear.weight, tape=tape)
        col = x_raw if self.dataset == 'QM9' else (x_raw.unsqueeze(-1) if x_raw.dim() == 1 else x_raw)[:, -1]
        idx = g.types if getattr(g, 'types', None) is not None else col.to(torch.int32).contiguous()
        if ops.type_rows_supported(self.embeddings):                                    # models.py:107,140
            direct = self.embeddings.grad if ((tape is not None or torch.is_grad_enabled()) and self.embeddings.grad is not None
                                              and getattr(self.embeddings, '_pamnet_direct', False)) else None
            return ops.type_rows(self.embeddings, idx, direct, tape=tape)
        tr = G.Transpose(idx, self.embeddings.size(0)) if torch.is_grad_enabled() else None
        return ops.gather(self.embeddings, idx, tr.ptr if tr else None, tr.perm if tr else None)

    def _edge_embeddings(self, g, tape=None):
        sbf = g.sbf                                                                          # [T+P, 42], no grad
        if self._narrow(g.dist_g) and g.loc.m > 0 and g.glob.m > 0 and self.sbf.envelope_exponent == 5:
            # dim 16 / 32 / 64: the Bessel rows are formed inside the embedding kernels (no [E, 16] tensor either way)
            e_l = narrow.embed_rbf(g.dist_l, self.rbf_l.freq, self.cutoff_l, self.mlp_rbf_l[0][0])
            e_g = narrow.embed_rbf(g.dist_g, self.rbf_g.freq, self.cutoff_g, self.mlp_rbf_g[0][0])
            return e_l, e_g, sbf
        rbf_l = self.rbf_l(g.dist_l, tape=tape)
        rbf_g = self.rbf_g(g.dist_g, tape=tape)
        if self._embed_fused(rbf_l, self.mlp_rbf_l):
            e_l = fused.embed(rbf_l, self.mlp_rbf_l[0][0], tape=tape, need_dx=True)          # models.py:186
            e_g = fused.embed(rbf_g, self.mlp_rbf_g[0][0], tape=tape, need_dx=True)          # models.py:185
        elif self._narrow(rbf_l):
            e_l = narrow.embed(rbf_l, self.mlp_rbf_l[0][0])
            e_g = narrow.embed(rbf_g, self.mlp_rbf_g[0][0])
        else:
            e_l = mlp_apply(self.mlp_rbf_l, rbf_l)
            e_g = mlp_apply(self.mlp_rbf_g, rbf_g)
        return e_l, e_g, sbf

    def _narrow(self, x):
        return narrow.supported(x, self.dim)

    def _input_stage(self, data, g, tape, lin_a, lin_b=None):
        """(x, e_l, e_g, e_sbf) of models.py:107/119/140 and 185-188 as ONE launch (two in the backward) on the dim = 128
        path: the Bessel rows are formed inside the embedding kernel from the edge lengths, the type-table rows ride along.
        lin_a (, lin_b): the sbf embedding(s) -- with lin_b, rows of g.tp_kind == 1 use lin_b.  None when not applicable."""
        if not (self.dim == fused.D and g.sbf.is_cuda and self.sbf.default):
            return None                               # (the one-launch input stage is built for the default basis)
        lin_l, lin_g = self.mlp_rbf_l[0][0], self.mlp_rbf_g[0][0]
        layers = [(None, g.dist_l, self.cutoff_l, None, True, True), (None, g.dist_g, self.cutoff_g, None, True, True),
                  (g.sbf, None, None, g.tp_kind if lin_b is not None else None, True, True)]
        params = [self.rbf_l.freq, lin_l.weight, lin_l.bias, self.rbf_g.freq, lin_g.weight, lin_g.bias,
                  lin_a.weight, lin_a.bias]
        if lin_b is not None:
            params += [lin_b.weight, lin_b.bias]
        x_raw, types = data.x, None
        if self.dataset == 'PDBbind':
            xr = x_raw.unsqueeze(-1) if x_raw.dim() == 1 else x_raw
            feats = xr[:, 3:].to(torch.float32).contiguous()
            if feats.size(1) != 18:
                return None
            layers.append((feats, None, None, None, False, False))                          # models.py:119
            params.append(self.init_linear.weight)
        else:
            if not ops.type_rows_supported(self.embeddings):
                return None
            col = x_raw if self.dataset == 'QM9' else (x_raw.unsqueeze(-1) if x_raw.dim() == 1 else x_raw)[:, -1]
            types = g.types if getattr(g, 'types', None) is not None else col.to(torch.int32).contiguous()
            params.append(self.embeddings)                                                  # models.py:107,140
        outs = fused.input_stage(fused.InputSpec(layers, types), params, tape=tape)
        return outs[3], outs[0], outs[1], outs[2]

    @staticmethod
    def _embed_fused(x, seq):
        return len(seq) == 1 and fused.embed_supported(x, seq[0][0])

    def _run_layers(self, x, e_l, e_g, e_sbf, g, tape=None):
        if modules._fused(x):                      # dim = 128 on an MI355X: the whole loop is one engine call
            outs, atts, saved = fused.layer_stack(self.global_layer, self.local_layer, x, e_g, e_l, e_sbf, g, tape=tape)
            self._x_layers = _LazyLayers(saved, g, self.n_layer)
            return outs, atts
        if modules._narrow(x) and narrow.engine_supported(x, g):       # dim 16 / 32 / 64: one engine call as well
            outs, atts, saved = narrow.layer_stack(self.global_layer, self.local_layer, x, e_g, e_l, e_sbf, g, tape=tape)
            self._x_layers = _LazyLayers(saved, g, self.n_layer, self.dim)
            return outs, atts
        outs, atts = [], []
        self._x_layers = []
        for k in range(self.n_layer):
            x, o, a = self.global_layer[k](x, e_g, g)
            outs.append(o), atts.append(a), self._x_layers.append(x)
            x, o, a = self.local_layer[k](x, e_l, e_sbf, g)
            outs.append(o), atts.append(a), self._x_layers.append(x)
        return ops.stack_rows(outs), ops.stack_rows(atts)                                    # [2L, N]

    def _layers_and_pool(self, x, e_l, e_g, e_sbf, g, tape, mean):
        outs, atts = self._run_layers(x, e_l, e_g, e_sbf, g, tape)
        out, node_out = ops.fuse_pool(outs, atts, g, mean=mean, tape=tape)                  # models.py:206-224
        self._graph_cache, self._node_out = g, node_out
        return out if tape is not None else out.view(-1)        # (ops._Whole hands autograd its own view)

    def _one_node(self):
        """Training forward with preallocated gradients (train.FlatParams) on the fused dim = 128 path or the narrow-width
        row kernels: the whole forward is recorded on the model's own tape and handed to autograd as ONE node (ops.Tape)."""
        if not (ops.TAPE and torch.is_grad_enabled() and self.rbf_g.freq.is_cuda):
            return False
        if self.dim == fused.D:
            if not all(getattr(p, '_pamnet_direct', False) and p.grad is not None for p in self._top_params()):
                return False
            return fused.stack_plan(self.global_layer, self.local_layer).direct()
        if self.dim in narrow.WIDTHS and narrow.ENABLED:
            if not all(getattr(p, '_pamnet_direct', False) and p.grad is not None for p in self._all_params()):
                return False
            return fused.stack_plan(self.global_layer, self.local_layer).direct()      # (the engine's gradient tables)
        return False

    def _checked_forward(self, data):
        """The dtype check just walked the cache against the live tree (~0.05 ms of host time); nothing re-hangs parameters
        during a forward, so the other users of the cached lists inside it (the one-node test, the engine plan) take that
        result instead of repeating the walk -- three walks per training step were 5 % of the host-bound RNA step."""
        self.__dict__['_params_checked'] = True
        try:
            return self._on_own_device(self._forward, data)
        finally:
            self.__dict__['_params_checked'] = False

    def _derived_lists_valid(self):
        cache = self.__dict__.get('_named_param_cache')
        if cache is not None and self.__dict__.get('_params_checked'):
            return True
        if cache is None or not self._param_cache_valid(cache):
            self._drop_param_cache()
            return False
        return True

    def _all_params(self):
        ap = self.__dict__.get('_all_param_list') if self._derived_lists_valid() else None
        if ap is None:
            ap = self.__dict__['_all_param_list'] = [p for _, p in self._real_named_parameters() if p.requires_grad]
        return ap

    def _top_params(self):
        tp = self.__dict__.get('_top_param_list') if self._derived_lists_valid() else None
        if tp is None:                                   # walking the module tree costs ~1 ms: once per model
            tp = [p for n, p in self._real_named_parameters() if not n.startswith(('global_layer.', 'local_layer.'))]
            self.__dict__['_top_param_list'] = tp
        return tp

    def _release_inspection(self):
        """`_x_layers` / `_graph_cache` / `_node_out` (inspection hooks of the parity tests) reference the previous
        forward's saved-activation arena and graph: dropped before the next forward allocates its own, so they never
        double the peak activation memory (PDBbind / RNA sized batches)."""
        self._x_layers = self._graph_cache = self._node_out = None

    def _check_dataset(self, params=True):
        if params:
            self._check_dtype()
        if not (self.dataset in ('QM9', 'PDBbind') or self._rna):
            raise ValueError("Invalid dataset. If you are using any dataset related to RNA 3D structure prediction, "
                             "be sure to use 'rna' as the first 3 characters of the dataset name.")

    def _on_own_device(self, fn, data):
        """The library launches on the raw stream of the tensors' device; HIP wants that device current.  A model that lives
        on another device than the thread's current one (model.to('cuda:1') without torch.cuda.set_device) runs under a
        device guard instead of failing in the first launch."""
        dev = self.rbf_g.freq.device
        if dev.type == 'cuda' and dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                return fn(data)
        return fn(data)

    def _check_dtype(self):
        # the kernels read the parameters through raw pointers as fp32: a model cast with .double() / .half() / .bfloat16()
        # would be read with the wrong element size -- refused instead (first and last parameter: a cast touches all)
        ps = self._all_params()
        if ps and (ps[0].dtype != torch.float32 or ps[-1].dtype != torch.float32):
            raise TypeError('PAMNet on the MI355X kernels computes in float32 (the precision the 1e-5 parity bound needs): '
                            'parameters are %s -- call model.float()' % ps[0].dtype)


def _slice_outgoing(module, state_dict, prefix, local_metadata):
    """state_dict hook of a padded model: every tensor in the reference's shape -- a view of the padded parameter, or (the
    message MLPs' [dim, 3 dim] weights, whose three input blocks are not adjacent in the padded tensor) a gathered copy."""
    for k, shape in module._logical_shapes.items():
        t = state_dict.get(prefix + k)
        if t is None or tuple(t.shape) == shape:
            continue
        for axis, idx in enumerate(module._pad_index(shape, tuple(t.shape))):
            if idx is None:
                continue
            n = shape[axis]
            if int(idx[-1]) == n - 1:                     # one block at the origin: a plain slice keeps the view
                t = t.narrow(axis, 0, n)
            else:
                t = t.index_select(axis, idx.to(t.device))
        state_dict[prefix + k] = t
    return state_dict


class PAMNet(_PAMNetBase):
    """models.py:21-224."""

    def __init__(self, config, num_spherical=7, num_radial=6, envelope_exponent=5, _pad=True):
        super().__init__(config, num_spherical, num_radial, envelope_exponent, _pad)
        d = self.dim
        if self._rna:
            self.embeddings = nn.Parameter(torch.ones((3, d)))       # C, N, O
        else:
            self.embeddings = nn.Parameter(torch.ones((5, d)))
            self.init_linear = nn.Linear(18, d, bias=False)
        self._build_common(num_spherical, num_radial, envelope_exponent)
        self.mlp_sbf1 = MLP([num_spherical * num_radial, d])
        self.mlp_sbf2 = MLP([num_spherical * num_radial, d])
        self.global_layer = nn.ModuleList([GlobalMP(d) for _ in range(self.n_layer)])
        self.local_layer = nn.ModuleList([LocalMP(d) for _ in range(self.n_layer)])
        self.softmax = nn.Softmax(dim=-1)
        self.init()
        self._finish_padding()

    def forward(self, data):
        self._check_dataset()
        return self._checked_forward(data)

    def _forward(self, data):
        self._release_inspection()
        g = self._graph(data)
        if self._one_node():
            return self._run_one_node(lambda tape: self._forward_on(data, g, tape))
        return self._forward_on(data, g, None)

    def _forward_on(self, data, g, tape):
        staged = self._input_stage(data, g, tape, self.mlp_sbf2[0][0], self.mlp_sbf1[0][0])
        if staged is not None:
            x, e_l, e_g, e_sbf = staged
            return self._layers_and_pool(x, e_l, e_g, e_sbf, g, tape, self._rna)
        x = self._embed(data, g, tape)
        e_l, e_g, sbf = self._edge_embeddings(g, tape)
        # mlp_sbf2 on triplet rows, mlp_sbf1 on pair rows (models.py:187-188), rows grouped by target edge
        if self._embed_fused(sbf, self.mlp_sbf2):
            e_sbf = fused.embed(sbf, self.mlp_sbf2[0][0], self.mlp_sbf1[0][0], kind=g.tp_kind, tape=tape)
        elif self._narrow(sbf) and sbf.size(1) == 42:
            e_sbf = narrow.embed(sbf, self.mlp_sbf2[0][0], self.mlp_sbf1[0][0], kind=g.tp_kind)
        else:                                      # a basis width without a kernel of its own: tape-aware dense layers
            e_sbf = ops.dense_act(sbf, self.mlp_sbf2[0][0], self.mlp_sbf1[0][0], kind=g.tp_kind, tape=tape)
        return self._layers_and_pool(x, e_l, e_g, e_sbf, g, tape, self._rna)


class PAMNet_s(_PAMNetBase):
    """models.py:227-353 (QM9 only; one-hop pairs only)."""
    small = True
    max_num_neighbors = 500             # radius(..., max_num_neighbors=500), models.py:301

    def __init__(self, config, num_spherical=7, num_radial=6, envelope_exponent=5, _pad=True):
        super().__init__(config, num_spherical, num_radial, envelope_exponent, _pad)
        d = self.dim
        self.embeddings = nn.Parameter(torch.ones((5, d)))
        self._build_common(num_spherical, num_radial, envelope_exponent)
        self.mlp_sbf = MLP([num_spherical * num_radial, d])
        self.global_layer = nn.ModuleList([GlobalMP(d) for _ in range(self.n_layer)])
        self.local_layer = nn.ModuleList([LocalMP(d, small=True) for _ in range(self.n_layer)])
        self.softmax = nn.Softmax(dim=-1)
        self.init()
        self._finish_padding()

    def forward(self, data):
        if self.dataset != "QM9":
            raise ValueError("Invalid dataset. The current PAMNet_s is only for QM9 experiments.")
        self._check_dtype()
        return self._checked_forward(data)

    def _forward(self, data):
        self._release_inspection()
        g = self._graph(data)
        if self._one_node():
            return self._run_one_node(lambda tape: self._forward_on(data, g, tape))
        return self._forward_on(data, g, None)

    def _forward_on(self, data, g, tape):
        staged = self._input_stage(data, g, tape, self.mlp_sbf[0][0])
        if staged is not None:
            x, e_l, e_g, e_sbf = staged
            return self._layers_and_pool(x, e_l, e_g, e_sbf, g, tape, False)
        x = self._embed(data, g, tape)
        e_l, e_g, sbf = self._edge_embeddings(g, tape)
        if self._embed_fused(sbf, self.mlp_sbf):
            e_sbf = fused.embed(sbf, self.mlp_sbf[0][0], tape=tape)
        elif self._narrow(sbf) and sbf.size(1) == 42:
            e_sbf = narrow.embed(sbf, self.mlp_sbf[0][0])
        else:
            e_sbf = ops.dense_act(sbf, self.mlp_sbf[0][0], tape=tape)
        return self._layers_and_pool(x, e_l, e_g, e_sbf, g, tape, False)

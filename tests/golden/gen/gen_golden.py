#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own code in the build container.

Run from anywhere:   python tests/golden/gen/gen_golden.py
Needs /root/reference (read-only) -- which does NOT exist on the GPU box; only the emitted .npz fixtures travel.
The reference's models.py / layers/*.py / utils/sbf.py are imported unmodified; its four absent third-party wheels are
replaced by the documented-semantics stand-ins in thirdparty_standins.py (see that file's header).

Fixtures hold DATA only: inputs, expected outputs (fp32 and fp64 reference runs), a few intermediates, constants.
Random-init weights are not stored: they are regenerated from a seed by oracle.init_state_dict (torch CPU generator,
same image on every box) and the fixture carries a checksum of them.
"""
import os
import sys
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..', '..'))
OUT = os.path.join(REPO, 'tests', 'golden')
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
import thirdparty_standins  # noqa: E402

thirdparty_standins.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import models as ref_models  # noqa: E402  (the reference's models.py)
from utils import sbf as ref_sbf  # noqa: E402

# builder-side helpers (inputs + seeded weights); loaded by path to avoid name clashes with the reference's modules
import importlib.util  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load('pamnet_synth', os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'pamnet_amd', 'synth.py'))
oracle = _load('pamnet_oracle', os.path.join(REPO, 'oracle', 'pamnet_oracle.py'))

torch.set_num_threads(8)


class Data(object):
    pass


def make_data(batch, dtype):
    d = Data()
    d.x = batch.x.to(dtype) if batch.x.is_floating_point() else batch.x
    d.batch = batch.batch
    if hasattr(batch, 'pos'):
        d.pos = batch.pos.to(dtype)
    if hasattr(batch, 'edge_index'):
        d.edge_index = batch.edge_index
    return d


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def run_reference(model, data, capture):
    """One forward of the reference model with hooks recording intermediates."""
    rec = {}
    hooks = []
    xs = []

    def layer_hook(mod, inp, out):
        xs.append(out[0].detach().clone())

    for k in range(model.n_layer):
        hooks.append(model.global_layer[k].register_forward_hook(layer_hook))
        hooks.append(model.local_layer[k].register_forward_hook(layer_hook))
    orig_add, orig_mean = ref_models.global_add_pool, ref_models.global_mean_pool

    def rec_pool(fn):
        def f(x, batch, size=None):
            rec['node_out'] = x.detach().clone().view(-1)
            return fn(x, batch, size)
        return f

    ref_models.global_add_pool, ref_models.global_mean_pool = rec_pool(orig_add), rec_pool(orig_mean)
    orig_indices = model.indices

    def rec_indices(edge_index, num_nodes):
        res = orig_indices(edge_index, num_nodes)
        rec['edge_index_l'] = edge_index.clone()
        names = ['idx_i', 'idx_j', 'idx_k', 'idx_kj', 'idx_ji', 'idx_i_pair', 'idx_j1_pair', 'idx_j2_pair',
                 'idx_jj_pair', 'idx_ji_pair']
        if len(res) == 5:
            names = names[5:]
        for n, r in zip(names, res):
            rec[n] = r.clone()
        return res

    model.indices = rec_indices
    if capture:
        def basis_hook(name):
            def f(mod, inp, out):
                rec.setdefault(name, []).append(out.detach().clone())
            return f
        hooks.append(model.rbf_l.register_forward_hook(basis_hook('rbf_l')))
        hooks.append(model.rbf_g.register_forward_hook(basis_hook('rbf_g')))
        hooks.append(model.sbf.register_forward_hook(basis_hook('sbf')))
    try:
        out = model(data)
    finally:
        for h in hooks:
            h.remove()
        ref_models.global_add_pool, ref_models.global_mean_pool = orig_add, orig_mean
        model.indices = orig_indices
    rec['out'] = out.detach().clone()
    rec['x_layers'] = torch.stack(xs)
    return rec


def to_np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, list):
            for i, t in enumerate(v):
                out['%s_%d' % (k, i)] = t.numpy()
        elif isinstance(v, torch.Tensor):
            out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    return out


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote %-32s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------- constructors
_SBF_CACHE = {}


def build_model(cls, cfg):
    """Reference model; the 15 s sympy construction of SphericalBasisLayer is shared through a cache."""
    basis = getattr(cfg, 'basis', None)
    if basis is not None:                 # PAMNet(config, num_spherical, num_radial, envelope_exponent): models.py:22
        key = (cfg.cutoff_l,) + tuple(basis)
        if key in _SBF_CACHE:
            orig = ref_models.SphericalBasisLayer
            ref_models.SphericalBasisLayer = lambda *a, **k: _SBF_CACHE[key]
            try:
                return cls(cfg, *basis)
            finally:
                ref_models.SphericalBasisLayer = orig
        m = cls(cfg, *basis)
        _SBF_CACHE[key] = m.sbf
        return m
    key = cfg.cutoff_l
    if key in _SBF_CACHE:
        orig = ref_models.SphericalBasisLayer
        ref_models.SphericalBasisLayer = lambda *a, **k: _SBF_CACHE[key]
        try:
            m = cls(cfg)
        finally:
            ref_models.SphericalBasisLayer = orig
    else:
        m = cls(cfg)
        _SBF_CACHE[key] = m.sbf
    return m


# ------------------------------------------------------------------------------------------------- fixtures
def fixture_star():
    """indices() on a 4-node star, bonds 0-1, 1-2, 1-3."""
    ei = torch.tensor([[0, 1, 1, 1, 2, 3], [1, 0, 2, 3, 1, 1]])
    cfg = ref_models.Config(dataset='QM9', dim=8, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)
    m = build_model(ref_models.PAMNet, cfg)
    names = ['idx_i', 'idx_j', 'idx_k', 'idx_kj', 'idx_ji', 'idx_i_pair', 'idx_j1_pair', 'idx_j2_pair',
             'idx_jj_pair', 'idx_ji_pair']
    res = m.indices(ei, num_nodes=4)
    save('star_indices', edge_index=ei.numpy(), **{n: r.numpy() for n, r in zip(names, res)})


def fixture_basis():
    """Zeros / normalisers (utils/sbf.py:14-49) and dense tables of the reference's basis layers in fp64 and fp32."""
    zeros = ref_sbf.Jn_zeros(7, 6)
    norm = np.zeros((7, 6), dtype=np.float64)
    for l in range(7):
        for i in range(6):
            norm[l, i] = 1.0 / np.sqrt(np.float64(0.5 * ref_sbf.Jn(zeros[l, i], l + 1) ** 2))
    cfg = ref_models.Config(dataset='QM9', dim=8, n_layer=1, cutoff_l=5.0, cutoff_g=5.0)
    m = build_model(ref_models.PAMNet, cfg)
    dist = torch.linspace(0.75, 5.25, 96, dtype=torch.float64)            # x = d/5 in [0.15, 1.05]
    ang = torch.linspace(0.0, np.pi, 96, dtype=torch.float64)
    idx = torch.arange(96)
    sbf64 = m.sbf(dist, ang, idx)
    sbf32 = m.sbf(dist.float(), ang.float(), idx)
    rad64 = m.sbf(dist, torch.zeros_like(dist), idx)                       # angle 0: pure radial x Y_l0(0)
    rbf64 = m.rbf_l(dist)
    rbf32 = m.rbf_l(dist.float())
    save('basis_tables', zeros=zeros, norm=norm, dist=dist.numpy(), angle=ang.numpy(), sbf64=sbf64.detach().numpy(),
         sbf32=sbf32.detach().numpy(), rad64=rad64.detach().numpy(), rbf64=rbf64.detach().numpy(),
         rbf32=rbf32.detach().numpy(), cutoff=np.float64(5.0))


def fixture_rna():
    """Real shipped data + checkpoint (inference_rna_puzzles.py:46-66): graphs 6, 4, 17 (841 / 932 / 1152 nodes)."""
    raw = os.path.join(REF, 'data', 'RNA-Puzzles', 'rna_native', 'raw', 'rna_native_')
    gi = np.loadtxt(raw + 'graph_indicator.txt', dtype=np.int64) - 1
    na = np.loadtxt(raw + 'node_attributes.txt', delimiter=',', dtype=np.float32)
    nl = np.loadtxt(raw + 'node_labels.txt', dtype=np.float32)
    x_all = np.concatenate([na, nl[:, None]], 1)
    cfg = ref_models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0,
                            flow='target_to_source')
    model = build_model(ref_models.PAMNet, cfg)
    sd = torch.load(os.path.join(REF, 'save', 'pamnet_rna.pt'), map_location='cpu')
    print(model.load_state_dict(sd))
    model.eval()
    arrs = {'ckpt_keys': np.array(list(sd.keys()))}
    for k, v in sd.items():
        arrs['ckpt/' + k] = v.numpy()
    all32 = []
    for g in range(int(gi.max()) + 1):
        d = Data()
        d.x = torch.from_numpy(x_all[gi == g])
        d.batch = torch.zeros(d.x.size(0), dtype=torch.long)
        all32.append(float(model(d)))
    arrs['all_out32'] = np.asarray(all32, dtype=np.float32)
    arrs['all_num_nodes'] = np.bincount(gi)
    model64 = build_model(ref_models.PAMNet, cfg)
    model64.load_state_dict(sd)
    model64 = model64.double().eval()
    for g in (6, 4, 17):
        x = torch.from_numpy(x_all[gi == g])
        d = Data()
        d.x, d.batch = x, torch.zeros(x.size(0), dtype=torch.long)
        r32 = run_reference(model, d, capture=False)
        d64 = Data()
        d64.x, d64.batch = x.double(), d.batch
        r64 = run_reference(model64, d64, capture=False)
        arrs['g%d/x' % g] = x.numpy()
        arrs['g%d/out32' % g] = r32['out'].numpy()
        arrs['g%d/out64' % g] = r64['out'].numpy()
        arrs['g%d/num_edges_l' % g] = np.int64(r32['edge_index_l'].shape[1])
        arrs['g%d/num_triplets' % g] = np.int64(r32['idx_kj'].numel())
        arrs['g%d/num_pairs' % g] = np.int64(r32['idx_jj_pair'].numel())
        if g == 6:
            arrs['g6/x_layers32'] = r32['x_layers'].numpy()
            arrs['g6/x_layers64'] = r64['x_layers'].numpy()
            arrs['g6/node_out32'] = r32['node_out'].numpy()
            arrs['g6/node_out64'] = r64['node_out'].numpy()
    # graphs 4+6 batched together reproduce the per-graph outputs (graphs are independent units)
    xb = torch.from_numpy(np.concatenate([x_all[gi == 4], x_all[gi == 6]]))
    d = Data()
    d.x = xb
    d.batch = torch.cat([torch.zeros(int((gi == 4).sum()), dtype=torch.long),
                         torch.ones(int((gi == 6).sum()), dtype=torch.long)])
    arrs['batched_4_6_out32'] = model(d).detach().numpy()
    save('rna_native', **arrs)


def fixture_rna_all():
    """ALL 21 shipped RNA-Puzzles native structures as inference_rna_puzzles.py:46-66 scores them: TUDataset order, DataLoader
    batch_size=16, shuffle=False -> a batch of 16 graphs and one of 5, the shipped checkpoint, model.eval().  Inputs of every
    graph (xyz + label), the reference's outputs fp32 and fp64, and the integer sizes of both batches' graphs.  (The checkpoint
    tensors are in rna_native.npz.)"""
    raw = os.path.join(REF, 'data', 'RNA-Puzzles', 'rna_native', 'raw', 'rna_native_')
    gi = np.loadtxt(raw + 'graph_indicator.txt', dtype=np.int64) - 1
    na = np.loadtxt(raw + 'node_attributes.txt', delimiter=',', dtype=np.float32)
    nl = np.loadtxt(raw + 'node_labels.txt', dtype=np.float32)
    gl = np.loadtxt(raw + 'graph_labels.txt', dtype=np.float32)
    x_all = np.concatenate([na, nl[:, None]], 1)
    assert (np.diff(gi) >= 0).all()
    counts = np.bincount(gi)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    cfg = ref_models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0,
                            flow='target_to_source')
    sd = torch.load(os.path.join(REF, 'save', 'pamnet_rna.pt'), map_location='cpu')
    model = build_model(ref_models.PAMNet, cfg)
    model.load_state_dict(sd)
    model.eval()
    model64 = build_model(ref_models.PAMNet, cfg)
    model64.load_state_dict(sd)
    model64 = model64.double().eval()
    arrs = {'x_all': x_all, 'node_ptr': ptr.astype(np.int64), 'graph_labels': gl, 'batch_size': np.int64(16)}
    out32, out64, sizes = [], [], []
    n_graphs = len(counts)
    for b0 in range(0, n_graphs, 16):
        b1 = min(b0 + 16, n_graphs)
        x = torch.from_numpy(x_all[ptr[b0]:ptr[b1]])
        batch = torch.from_numpy(np.repeat(np.arange(b1 - b0), counts[b0:b1]))
        d = Data()
        d.x, d.batch = x, batch
        r32 = run_reference(model, d, capture=False)
        d64 = Data()
        d64.x, d64.batch = x.double(), batch
        r64 = run_reference(model64, d64, capture=False)
        out32.append(r32['out'].numpy().reshape(-1)), out64.append(r64['out'].numpy().reshape(-1))
        sizes.append([r32['edge_index_g'].shape[1] if 'edge_index_g' in r32 else -1, r32['edge_index_l'].shape[1],
                      r32['idx_kj'].numel(), r32['idx_jj_pair'].numel()])
        print('batch of %d graphs (%d nodes): done' % (b1 - b0, x.size(0)), flush=True)
    arrs['out32'] = np.concatenate(out32).astype(np.float32)
    arrs['out64'] = np.concatenate(out64).astype(np.float64)
    arrs['batch_sizes'] = np.asarray(sizes, np.int64)        # per batch: global edges (-1: not captured), local edges, triplets, pairs
    save('rna_native_all', **arrs)


def fixture_random(name, cls, cfg_kw, batch, seed, capture, small=False, basis=None):
    """Seeded random-init reference model on a synthetic batch; fp32 and fp64 runs.  basis: (num_spherical, num_radial,
    envelope_exponent) other than the default (7, 6, 5)."""
    cfg = ref_models.Config(**cfg_kw)
    if basis is not None:
        cfg.basis = tuple(basis)
    sd = oracle.init_state_dict(cfg, seed=seed, small=small)
    arrs = dict(seed=np.int64(seed), weights_checksum=np.float64(checksum(sd)),
                cfg_dataset=np.array(cfg_kw['dataset']), cfg_dim=np.int64(cfg_kw['dim']),
                cfg_n_layer=np.int64(cfg_kw['n_layer']), cfg_cutoff_l=np.float64(cfg_kw['cutoff_l']),
                cfg_cutoff_g=np.float64(cfg_kw['cutoff_g']), cfg_flow=np.array(cfg_kw.get('flow', 'source_to_target')))
    if basis is not None:
        arrs['cfg_basis'] = np.asarray(basis, np.int64)
    for k in ('x', 'batch', 'pos', 'edge_index', 'y'):
        if hasattr(batch, k):
            arrs['in/' + k] = getattr(batch, k).numpy()
    for tag, dtype in (('32', torch.float32), ('64', torch.float64)):
        model = build_model(cls, cfg)
        print(name, tag, model.load_state_dict(sd))
        if dtype == torch.float64:
            model = model.double()
        rec = run_reference(model, make_data(batch, dtype), capture=capture and tag == '32')
        arrs['out' + tag] = rec['out'].numpy()
        arrs['node_out' + tag] = rec['node_out'].numpy()
        if capture:
            arrs['x_layers' + tag] = rec['x_layers'].numpy()
        if tag == '32':
            arrs['num_edges_l'] = np.int64(rec['edge_index_l'].shape[1])
            arrs['num_pairs'] = np.int64(rec['idx_jj_pair'].numel())
            if 'idx_kj' in rec:
                arrs['num_triplets'] = np.int64(rec['idx_kj'].numel())
            if capture:
                arrs.update({'ref/' + k: v for k, v in to_np(
                    {k: rec[k] for k in rec if k.startswith('idx_') or k in ('edge_index_l', 'rbf_l', 'rbf_g', 'sbf')}
                ).items()})
    # loss-gradient golden (fp64): d mean|out - y| / d params, a few named tensors + global norm
    model = build_model(cls, cfg)
    model.load_state_dict(sd)
    model = model.double()
    out = model(make_data(batch, torch.float64))
    loss = torch.nn.functional.l1_loss(out, batch.y.double())
    loss.backward()
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters() if p.grad is not None))
    arrs['loss64'] = np.float64(loss.item())
    arrs['grad_norm64'] = np.float64(gn.item())
    for k, p in model.named_parameters():
        if p.grad is not None and k in ('embeddings', 'rbf_g.freq', 'rbf_l.freq', 'mlp_sbf1.0.0.weight', 'mlp_sbf.0.0.weight',
                 'global_layer.0.mlp_m.0.0.weight', 'local_layer.0.mlp_sbf.1.0.bias', 'local_layer.0.lin_rbf.weight',
                 'global_layer.0.W', 'init_linear.weight'):
            arrs['grad64/' + k] = p.grad.numpy()
    save(name, **arrs)


def fixture_train(name, cfg_kw, batch, seed, lrs, max_norm, loss='l1', ema=True):
    """K optimiser steps of the reference training loop body (main_qm9.py:103-118) on one fixed batch: reference model,
    torch Adam (wd=0, amsgrad=False), clip_grad_norm_, the reference's own EMA class (utils/ema.py); per-step learning
    rates are given explicitly (the warm-up scheduler wheel is absent).  fp32 and fp64.
    loss / max_norm=None / ema=False select the other two drivers' loop bodies: main_pdbbind.py:88-95 (F.mse_loss, no clip,
    no EMA; MultiStepLR stepped per epoch -> the per-step rates are handed in) and main_rna_puzzles.py:86-93
    (F.smooth_l1_loss, no clip, no EMA, constant rate)."""
    from torch.nn.utils import clip_grad_norm_
    from utils.ema import EMA as RefEMA
    cfg = ref_models.Config(**cfg_kw)
    sd = oracle.init_state_dict(cfg, seed=seed)
    arrs = dict(seed=np.int64(seed), weights_checksum=np.float64(checksum(sd)), lrs=np.asarray(lrs, np.float64),
                max_norm=np.float64(-1.0 if max_norm is None else max_norm), cfg_dataset=np.array(cfg_kw['dataset']),
                cfg_dim=np.int64(cfg_kw['dim']),
                cfg_n_layer=np.int64(cfg_kw['n_layer']), cfg_cutoff_l=np.float64(cfg_kw['cutoff_l']),
                cfg_cutoff_g=np.float64(cfg_kw['cutoff_g']), cfg_flow=np.array(cfg_kw.get('flow', 'source_to_target')),
                loss_kind=np.array(loss), ema=np.int64(1 if ema else 0))
    loss_fn = {'l1': torch.nn.functional.l1_loss, 'mse': torch.nn.functional.mse_loss,
               'smooth_l1': torch.nn.functional.smooth_l1_loss}[loss]
    for k in ('x', 'batch', 'pos', 'edge_index', 'y'):
        if hasattr(batch, k):
            arrs['in/' + k] = getattr(batch, k).numpy()
    for tag, dtype in (('32', torch.float32), ('64', torch.float64)):
        model = build_model(ref_models.PAMNet, cfg)
        model.load_state_dict(sd)
        if dtype == torch.float64:
            model = model.double()
        opt = torch.optim.Adam(model.parameters(), lr=lrs[0], weight_decay=0, amsgrad=False)
        ema_obj = RefEMA(model, decay=0.999) if ema else None
        data, y = make_data(batch, dtype), batch.y.to(dtype)
        losses, norms = [], []
        for lr in lrs:
            for grp in opt.param_groups:
                grp['lr'] = lr
            opt.zero_grad()
            out = model(data)
            loss_v = loss_fn(out, y)
            loss_v.backward()
            if max_norm is not None:
                norms.append(float(clip_grad_norm_(model.parameters(), max_norm=max_norm, norm_type=2)))
            else:                                  # recorded, not applied
                norms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()
                                                  if p.grad is not None))))
            opt.step()
            if ema_obj is not None:
                ema_obj(model)
            losses.append(float(loss_v))
        arrs['loss' + tag] = np.asarray(losses, np.float64)
        arrs['grad_norm' + tag] = np.asarray(norms, np.float64)
        arrs['param_l2_' + tag] = np.float64(torch.sqrt(sum((p.double() ** 2).sum() for p in model.parameters())))
        if ema_obj is not None:
            arrs['shadow_l2_' + tag] = np.float64(torch.sqrt(sum((v.double() ** 2).sum() for v in ema_obj.shadow.values())))
        arrs['delta_l2_' + tag] = np.float64(torch.sqrt(sum(((p.double() - sd[k].double()) ** 2).sum()
                                                           for k, p in model.named_parameters())))
        with torch.no_grad():
            arrs['out_final' + tag] = model(data).numpy()
            if ema_obj is not None:
                ema_obj.assign(model)
                arrs['out_ema' + tag] = model(data).numpy()
                ema_obj.resume(model)
    save(name, **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    if '--train-only' in sys.argv:
        return main_train()
    if '--train-other-only' in sys.argv:
        return main_train_other()
    if '--ragged-only' in sys.argv:
        return main_ragged()
    if '--d128-only' in sys.argv:
        return main_d128()
    if '--baseline-only' in sys.argv:
        return main_baseline()
    if '--basis-only' in sys.argv:
        return main_basis()
    if '--baseline-s-only' in sys.argv:
        return main_baseline_s()
    if '--wide-only' in sys.argv:
        return main_wide()
    if '--rna-all-only' in sys.argv:
        return fixture_rna_all()
    if '--flow-only' in sys.argv:
        return main_flow()
    main_forward()
    main_train()
    main_baseline()
    main_baseline_s()
    main_basis()
    main_wide()


def fixture_baseline(name, cfg_kw, batch, seed, grads=False, small=False):
    """The reference itself at a BASELINE.json batch size: graph outputs and pooled node values (fp32 and fp64 runs) plus
    the integer sizes of its graphs.  Inputs are not stored -- tests regenerate them from pamnet_amd.synth (deterministic
    per graph index) and check the checksum kept here."""
    cfg = ref_models.Config(**cfg_kw)
    cls = ref_models.PAMNet_s if small else ref_models.PAMNet
    sd = oracle.init_state_dict(cfg, seed=seed, small=small)
    arrs = dict(seed=np.int64(seed), weights_checksum=np.float64(checksum(sd)),
                cfg_dataset=np.array(cfg_kw['dataset']), cfg_dim=np.int64(cfg_kw['dim']),
                cfg_n_layer=np.int64(cfg_kw['n_layer']), cfg_cutoff_l=np.float64(cfg_kw['cutoff_l']),
                cfg_cutoff_g=np.float64(cfg_kw['cutoff_g']), cfg_flow=np.array(cfg_kw.get('flow', 'source_to_target')),
                x_checksum=np.float64(batch.x.double().abs().sum()), num_nodes=np.int64(batch.x.size(0)),
                num_graphs=np.int64(int(batch.batch.max()) + 1))
    for tag, dtype in (('32', torch.float32), ('64', torch.float64)):
        model = build_model(cls, cfg)
        print(name, tag, model.load_state_dict(sd), flush=True)
        if dtype == torch.float64:
            model = model.double()
        with torch.no_grad():
            rec = run_reference(model, make_data(batch, dtype), capture=False)
        arrs['out' + tag] = rec['out'].numpy()
        arrs['node_out' + tag] = rec['node_out'].numpy()
        if tag == '32':
            arrs['num_edges_l'] = np.int64(rec['edge_index_l'].shape[1])
            arrs['num_pairs'] = np.int64(rec['idx_jj_pair'].numel())
            arrs['num_triplets'] = np.int64(rec['idx_kj'].numel() if 'idx_kj' in rec else 0)
    if grads:
        # the reference's own fp64 autograd of the mean L1 loss at this batch: loss, global gradient norm, every parameter
        # gradient's max-magnitude and L2 norm (checks of the whole backward) and a few full tensors
        model = build_model(cls, cfg)
        model.load_state_dict(sd)
        model = model.double()
        out = model(make_data(batch, torch.float64))
        loss = torch.nn.functional.l1_loss(out, batch.y.double())
        loss.backward()
        named = [(k, p) for k, p in model.named_parameters() if p.grad is not None]
        arrs['loss64'] = np.float64(loss.item())
        arrs['grad_norm64'] = np.float64(torch.sqrt(sum((p.grad ** 2).sum() for _, p in named)).item())
        arrs['grad_keys'] = np.array([k for k, _ in named])
        arrs['grad_l2_64'] = np.array([float(p.grad.norm()) for _, p in named], np.float64)
        arrs['grad_max_64'] = np.array([float(p.grad.abs().max()) for _, p in named], np.float64)
        for k, p in named:
            if k in ('embeddings', 'rbf_g.freq', 'rbf_l.freq', 'mlp_sbf1.0.0.weight', 'mlp_rbf_g.0.0.weight',
                     'global_layer.0.mlp_m.0.0.weight', 'global_layer.5.mlp_m.0.0.weight', 'global_layer.3.W_edge_attr.weight',
                     'local_layer.0.mlp_sbf.1.0.bias', 'local_layer.0.lin_rbf.weight', 'local_layer.5.mlp_m_kj.0.0.weight',
                     'local_layer.2.res2.mlp.1.0.weight', 'global_layer.0.W', 'local_layer.5.W_out.weight'):
                arrs['grad64/' + k] = p.grad.numpy().astype(np.float32 if p.numel() > 4096 else np.float64)
    save(name, **arrs)


def main_baseline():
    """Reference runs at the BASELINE.json batch sizes (outputs only): configs[0] QM9 B=32 and configs[1] QM9 B=128 (the
    exact batch bench.py times; with the reference's fp64 loss gradient), the PDBbind B=32 batch and its first 8-complex
    shard, the RNA B=8 batch."""
    qm9 = dict(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    fixture_baseline('baseline_qm9_b32', qm9, synth.qm9_batch(0, 0, 32), seed=0)
    fixture_baseline('baseline_qm9_b128', qm9, synth.qm9_batch(0, 0, 128), seed=0, grads=True)
    pdb = dict(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    fixture_baseline('baseline_pdbbind_b8', pdb, synth.collate([synth.pdbbind_complex(1, i) for i in range(8)]), seed=3)
    fixture_baseline('baseline_pdbbind_b32', pdb, synth.collate([synth.pdbbind_complex(1, i) for i in range(32)]), seed=3)
    rna = dict(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    fixture_baseline('baseline_rna_b8', rna, synth.rna_batch(2, 0, 8), seed=5)


def _wide_batches():
    return {'wide_qm9_d192_l2': synth.qm9_batch(21, 0, 16),
            'wide_qm9s_d136_l2': synth.qm9_batch(22, 0, 16),
            'wide_pdbbind_d160_l2': synth.pdbbind_batch(7, 0, 2, n_pocket=70, n_ligand=14),
            'wide_rna_d144_l1': synth.rna_batch(5, 0, 2, n_nodes=150)}


def main_wide():
    """Hidden sizes above 128 (models.py:25) from the reference itself: the widths whose dense layers run on csrc/dense.hip.
    Outputs, pooled node values, graph sizes (fp32 and fp64 runs) and the reference's fp64 gradients of the mean L1 loss."""
    b = _wide_batches()
    fixture_baseline('wide_qm9_d192_l2', dict(dataset='QM9', dim=192, n_layer=2, cutoff_l=5.0, cutoff_g=5.0),
                     b['wide_qm9_d192_l2'], seed=41, grads=True)
    fixture_baseline('wide_qm9s_d136_l2', dict(dataset='QM9', dim=136, n_layer=2, cutoff_l=5.0, cutoff_g=5.0),
                     b['wide_qm9s_d136_l2'], seed=42, grads=True, small=True)
    fixture_baseline('wide_pdbbind_d160_l2', dict(dataset='PDBbind', dim=160, n_layer=2, cutoff_l=2.0, cutoff_g=6.0),
                     b['wide_pdbbind_d160_l2'], seed=43, grads=True)
    fixture_baseline('wide_rna_d144_l1', dict(dataset='rna_native', dim=144, n_layer=1, cutoff_l=2.6, cutoff_g=20.0,
                                              flow='target_to_source'), b['wide_rna_d144_l1'], seed=44, grads=True)


def main_baseline_s():
    """PAMNet_s (pairs only, models.py:283-353) at the batch bench.py times: the reference itself, fp32 and fp64."""
    qm9 = dict(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    fixture_baseline('baseline_qm9s_b128', qm9, synth.qm9_batch(0, 0, 128), seed=0, small=True)


def main_basis():
    """Non-default basis sizes straight from the reference: PAMNet(config, num_spherical, num_radial, envelope_exponent)
    (models.py:22) -- a smaller basis with another envelope at a narrow width, a larger one at dim = 128."""
    qm9 = dict(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    fixture_random('qm9_basis_5x4_p6_d32_l2', ref_models.PAMNet, qm9, synth.qm9_batch(11, 0, 6), seed=3, capture=True,
                   basis=(5, 4, 6))
    big = dict(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    fixture_random('qm9_basis_8x7_p4_d128_l2', ref_models.PAMNet, big, synth.qm9_batch(12, 0, 5), seed=4, capture=False,
                   basis=(8, 7, 4))


def main_train():
    qm9 = dict(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    lrs = [1e-3, 2e-3, 3e-3, 3e-3, 2.5e-3]
    fixture_train('train_qm9_d32_l2', qm9, synth.qm9_batch(11, 0, 6), seed=3, lrs=lrs, max_norm=1000.0)
    big = dict(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    # max_norm below the first gradient norm so that the clip is exercised
    fixture_train('train_qm9_d128_l2', big, synth.qm9_batch(0, 0, 8), seed=6, lrs=lrs, max_norm=0.5)
    main_train_other()


def main_train_other():
    """The loop bodies of the other two drivers (the data sets themselves are not available offline: schema-true synthetic
    batches).  PDBbind: MultiStepLR(gamma=0.2) crossing two milestones inside the five steps; targets ~ pK values so that
    |out - y| > 1 (the quadratic and the linear branch of smooth-L1 / the magnitude of the MSE gradient are both live)."""
    pdb = dict(dataset='PDBbind', dim=32, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)
    b = synth.pdbbind_batch(5, 0, 3, n_pocket=70, n_ligand=14)
    fixture_train('train_pdbbind_d32_l2', pdb, b, seed=5, lrs=[1e-3, 1e-3, 2e-4, 2e-4, 4e-5], max_norm=None, loss='mse',
                  ema=False)
    pdb128 = dict(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    b = synth.pdbbind_batch(9, 0, 2, n_pocket=90, n_ligand=16)
    # (rates ten times smaller than at d = 32: at 1e-3 this model's loss falls 43 -> 0.04 within the five steps and the
    # sequence amplifies rounding differences -- the reference's own fp32 run is 1e-4 off its fp64 run by step 5)
    fixture_train('train_pdbbind_d128_l3', pdb128, b, seed=31, lrs=[1e-4, 1e-4, 2e-5, 2e-5, 4e-6], max_norm=None, loss='mse',
                  ema=False)
    rna = dict(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    b = synth.rna_batch(2, 0, 3, n_nodes=260)
    b.y = torch.tensor([0.4, 2.5, 7.0])            # RMSD-like targets: |out - y| on both sides of 1
    fixture_train('train_rna_d16_l1', rna, b, seed=12, lrs=[5e-4] * 5, max_norm=None, loss='smooth_l1', ema=False)


def main_ragged():
    qm9 = dict(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    fixture_random('qm9_ragged_d32_l2', ref_models.PAMNet, qm9, synth.ragged_qm9_batch(), seed=8, capture=True)
    fixture_random('qm9s_ragged_d32_l2', ref_models.PAMNet_s, qm9, synth.ragged_qm9_batch(), seed=9, capture=True,
                   small=True)


def main_d128():
    """dim = 128 (the fused MI355X engine's width) straight from the reference: PDBbind branch and PAMNet_s; outputs and
    pooled-node values only (small files)."""
    pdb = dict(dataset='PDBbind', dim=128, n_layer=3, cutoff_l=2.0, cutoff_g=6.0)
    fixture_random('pdbbind_d128_l3', ref_models.PAMNet, pdb, synth.pdbbind_batch(9, 0, 2, n_pocket=90, n_ligand=16),
                   seed=31, capture=False)
    qm9 = dict(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    fixture_random('qm9s_d128_l2', ref_models.PAMNet_s, qm9, synth.qm9_batch(13, 0, 12), seed=31, capture=False, small=True)


def main_flow():
    """Round 6 (verdict item 9d): flow = 'target_to_source' on the QM9 and the PDBbind branch straight from the reference (until
    now reference runs used the non-default flow on the RNA branch only; layers/global_message_passing.py:11, models.py:19)."""
    qm9 = dict(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0, flow='target_to_source')
    fixture_random('qm9_flow_t2s_d32_l2', ref_models.PAMNet, qm9, synth.qm9_batch(21, 0, 6), seed=13, capture=True)
    pdb = dict(dataset='PDBbind', dim=32, n_layer=2, cutoff_l=2.0, cutoff_g=6.0, flow='target_to_source')
    fixture_random('pdbbind_flow_t2s_d32_l2', ref_models.PAMNet, pdb, synth.pdbbind_batch(15, 0, 2, n_pocket=70, n_ligand=14),
                   seed=15, capture=True)


def main_forward():
    main_flow()
    main_ragged()
    main_d128()
    fixture_star()
    fixture_basis()
    fixture_rna()
    qm9 = dict(dataset='QM9', dim=32, n_layer=2, cutoff_l=5.0, cutoff_g=5.0)
    fixture_random('qm9_d32_l2', ref_models.PAMNet, qm9, synth.qm9_batch(11, 0, 6), seed=3, capture=True)
    fixture_random('qm9s_d32_l2', ref_models.PAMNet_s, qm9, synth.qm9_batch(11, 0, 6), seed=4, capture=True, small=True)
    pdb = dict(dataset='PDBbind', dim=32, n_layer=2, cutoff_l=2.0, cutoff_g=6.0)
    fixture_random('pdbbind_d32_l2', ref_models.PAMNet, pdb, synth.pdbbind_batch(5, 0, 2, n_pocket=70, n_ligand=14),
                   seed=5, capture=True)
    big = dict(dataset='QM9', dim=128, n_layer=6, cutoff_l=5.0, cutoff_g=5.0)
    fixture_random('qm9_d128_l6', ref_models.PAMNet, big, synth.qm9_batch(0, 0, 8), seed=1, capture=False)


if __name__ == '__main__':
    main()

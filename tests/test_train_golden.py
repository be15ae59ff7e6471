"""Training-loop parity: K optimiser steps on one fixed batch against sequences produced by the
reference's own model + torch Adam + clip_grad_norm_ + the reference's EMA class (tests/golden/gen/gen_golden.py
fixture_train): per-step loss, pre-clip gradient norm, L2 norm of the parameter displacement / parameters / EMA shadow,
and the outputs under the final and the EMA weights.

  * CPU: the oracle's restatement of the loop body (main_qm9.py:103-118, utils/ema.py:13-20) in fp64 vs the fp64 fixture.
  * GPU: pamnet_amd.train.Trainer (flat buffers, direct gradient writes, fused Adam) vs the fixtures, with
    err(hip, ref64) <= max(tol, 2 * err(ref32, ref64)).
"""
import numpy as np
import pytest
import torch

from test_hip_model import _batch_from

# train_qm9_*: main_qm9.py:103-118 (L1, clip, EMA).  train_pdbbind_*: main_pdbbind.py:88-95 (MSE, no clip, no EMA, per-step
# rates of a MultiStepLR crossing two milestones).  train_rna_*: main_rna_puzzles.py:86-93 (smooth-L1, no clip, no EMA).
NAMES = ['train_qm9_d32_l2', 'train_qm9_d128_l2', 'train_pdbbind_d32_l2', 'train_pdbbind_d128_l3', 'train_rna_d16_l1']


def _cfg(g, Config):
    flow = str(g['cfg_flow']) if 'cfg_flow' in g.files else 'source_to_target'
    return Config(dataset=str(g['cfg_dataset']), dim=int(g['cfg_dim']), n_layer=int(g['cfg_n_layer']),
                  cutoff_l=float(g['cfg_cutoff_l']), cutoff_g=float(g['cfg_cutoff_g']), flow=flow)


def _loop_kind(g):
    """(loss kind, max_norm or None, EMA on) of a training fixture."""
    kind = str(g['loss_kind']) if 'loss_kind' in g.files else 'l1'
    max_norm = float(g['max_norm'])
    ema = bool(int(g['ema'])) if 'ema' in g.files else True
    return kind, (None if max_norm < 0 else max_norm), ema


LOSSES = {'l1': torch.nn.functional.l1_loss, 'mse': torch.nn.functional.mse_loss,
          'smooth_l1': torch.nn.functional.smooth_l1_loss}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize('name', NAMES)
def test_oracle_training_loop_vs_reference(golden, name):
    from oracle import pamnet_oracle as O
    g = golden(name)
    cfg = _cfg(g, O.Config)
    sd0 = {k: v.double() for k, v in O.init_state_dict(cfg, seed=int(g['seed'])).items()}
    assert abs(sum(float(v.abs().sum()) for v in sd0.values()) - float(g['weights_checksum'])) < 1e-6 * float(g['weights_checksum'])
    params = O.as_params(sd0)
    names = list(params.keys())
    opt = torch.optim.Adam([params[k] for k in names], lr=float(g['lrs'][0]), weight_decay=0, amsgrad=False)
    shadow = {k: params[k].data.clone() for k in names}                                   # utils/ema.py:9-11
    kind, max_norm, ema = _loop_kind(g)
    x, batch, y = torch.from_numpy(g['in/x']), torch.from_numpy(g['in/batch']), torch.from_numpy(g['in/y']).double()
    pos = torch.from_numpy(g['in/pos']).double() if 'in/pos' in g.files else None
    ei = torch.from_numpy(g['in/edge_index']) if 'in/edge_index' in g.files else None
    if x.dim() == 2:
        x = x.double()                                   # PDBbind / RNA rows carry the coordinates
    losses, norms = [], []
    for lr in g['lrs']:
        opt.param_groups[0]['lr'] = float(lr)
        opt.zero_grad()
        out = O.pamnet_forward(params, cfg, x, batch, pos, ei, dtype=torch.float64)
        loss = LOSSES[kind](out, y)                      # main_qm9.py:108 / main_pdbbind.py:93 / main_rna_puzzles.py:92
        loss.backward()
        grads = [params[k].grad for k in names if params[k].grad is not None]
        norm = torch.sqrt(sum((gr ** 2).sum() for gr in grads))                           # clip_grad_norm_, main_qm9.py:111
        if max_norm is not None:
            coef = min(1.0, max_norm / (float(norm) + 1e-6))
            for gr in grads:
                gr.mul_(coef)
        opt.step()
        decay = min(0.999, (1.0 + 99999) / (10.0 + 99999))                                # utils/ema.py:14
        for k in names:
            shadow[k] = (1.0 - decay) * params[k].data + decay * shadow[k]
        losses.append(float(loss.detach())), norms.append(float(norm))
    # fp64 on both sides; Adam's g/|g| updates amplify last-bit differences of the first step, hence 2e-7 and not 1e-12
    assert _rel(losses, g['loss64']) < 2e-7 and _rel(norms, g['grad_norm64']) < 2e-7
    l2 = lambda d: float(torch.sqrt(sum((v.double() ** 2).sum() for v in d.values())))
    assert abs(l2({k: params[k].data for k in names}) - float(g['param_l2_64'])) < 1e-9 * float(g['param_l2_64'])
    assert abs(l2({k: params[k].data - sd0[k] for k in names}) - float(g['delta_l2_64'])) < 1e-7 * float(g['delta_l2_64'])
    with torch.no_grad():
        out_fin = O.pamnet_forward({k: params[k].data for k in names}, cfg, x, batch, pos, ei, dtype=torch.float64)
        assert _rel(out_fin.numpy(), g['out_final64']) < 2e-7
        if ema:
            assert abs(l2(shadow) - float(g['shadow_l2_64'])) < 1e-9 * float(g['shadow_l2_64'])
            out_ema = O.pamnet_forward(shadow, cfg, x, batch, pos, ei, dtype=torch.float64)
            assert _rel(out_ema.numpy(), g['out_ema64']) < 2e-7


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_trainer_vs_reference_training_loop(golden, name):
    import models
    from oracle import pamnet_oracle as O
    from pamnet_amd.train import Trainer
    dev = torch.device('cuda:0')
    g = golden(name)
    cfg = _cfg(g, models.Config)
    model = models.PAMNet(cfg)
    sd0 = O.init_state_dict(cfg, seed=int(g['seed']))
    model.load_state_dict(sd0, strict=True)
    model = model.to(dev)
    kind, max_norm, ema = _loop_kind(g)
    tr = Trainer(model, lr=float(g['lrs'][0]), weight_decay=0.0, ema_decay=0.999 if ema else None, max_grad_norm=max_norm,
                 loss=kind)
    data = _batch_from(g, dev)
    losses, norms = [], []
    for lr in g['lrs']:
        loss = tr.step(data, lr=float(lr))
        losses.append(float(loss.detach())), norms.append(float(tr.last_grad_norm))

    def ok(a, k, tol):
        e, floor = _rel(a, g[k + '64']), _rel(g[k + '32'], g[k + '64'])
        return e <= max(tol, 2 * floor), (k, e, floor)

    # losses / norms of later steps inherit the fp32 noise of the earlier Adam updates (update ~ lr * g/|g|)
    for a, k, tol in ((losses, 'loss', 1e-5), (norms, 'grad_norm', 1e-5)):
        good, info = ok(a, k, tol)
        assert good, info
    cur = {k: p.detach().cpu().double() for k, p in model.state_dict().items()}
    assert set(cur) == set(sd0)
    l2 = lambda d: float(torch.sqrt(sum((v ** 2).sum() for v in d.values())))
    checks = [(l2(cur), 'param_l2_'), (l2({k: cur[k] - sd0[k].double() for k in cur}), 'delta_l2_')]
    if ema:
        checks.append((float(torch.linalg.vector_norm(tr.shadow.double())), 'shadow_l2_'))
    for val, k in checks:
        good, info = ok([val], k, 1e-5)
        assert good, info
    scale = None
    if cfg.dataset == 'PDBbind':                  # complex - pocket - ligand: relative to the targets' magnitude once trained
        scale = float(np.abs(g['in/y']).max())
    with torch.no_grad():
        out_fin = model(data).cpu().numpy()
    if scale is None:
        good, info = ok(out_fin, 'out_final', 1e-5)
    else:
        e = float(np.abs(out_fin - g['out_final64']).max()) / scale
        floor = float(np.abs(g['out_final32'] - g['out_final64']).max()) / scale
        good, info = e <= max(1e-5, 2 * floor), ('out_final', e, floor)
    assert good, info
    if ema:
        with torch.no_grad():
            tr.ema_assign()
            out_ema = model(data).cpu().numpy()
            tr.ema_resume()
        good, info = ok(out_ema, 'out_ema', 1e-5)
        assert good, info
        # evaluate() = MAE under EMA weights (main_qm9.py:29-37)
        mae = tr.evaluate([data])
        assert abs(mae - float(np.abs(g['out_ema64'] - g['in/y']).mean())) < 1e-5 * max(1.0, float(np.abs(g['in/y']).mean()))
    else:
        # test() of main_pdbbind.py:25-39 / main_rna_puzzles.py:26-46: predictions under the weights themselves
        pred, y = tr.predictions([data])
        assert np.array_equal(y, g['in/y']) and np.allclose(pred, out_fin, rtol=0, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
@pytest.mark.parametrize('flat', [False, True])
def test_reference_loop_unchanged_on_the_hip_model(golden, name, flat, monkeypatch):
    """"Drops into main_qm9.py unchanged": the reference's OWN loop bodies -- torch.optim.Adam + loss.backward() +
    clip_grad_norm_ + utils.EMA (main_qm9.py:99-118), F.mse_loss + Adam (main_pdbbind.py:88-95), F.smooth_l1_loss + Adam
    (main_rna_puzzles.py:86-93) -- on the HIP model through plain autograd, no pamnet_amd.train.Trainer anywhere, against the
    same reference-run sequences."""
    import models
    from oracle import pamnet_oracle as O
    from torch.nn.utils import clip_grad_norm_
    from utils import EMA
    # flat: PAMNET_FLAT_PARAMS=1 -- model.parameters() is ONE flat tensor (models._FlatView); the loop text is the same, the
    # numbers must be (Adam / EMA are elementwise, the clip norm runs over the same elements), state_dict() keeps its keys
    monkeypatch.setenv('PAMNET_FLAT_PARAMS', '1' if flat else '0')
    dev = torch.device('cuda:0')
    g = golden(name)
    cfg = _cfg(g, models.Config)
    kind, max_norm, use_ema = _loop_kind(g)
    model = models.PAMNet(cfg)
    sd0 = O.init_state_dict(cfg, seed=int(g['seed']))
    model.load_state_dict(sd0, strict=True)
    model = model.to(dev)
    optimizer = torch.optim.Adam(model.parameters(), lr=float(g['lrs'][0]), weight_decay=0, amsgrad=False)
    n_seen = len(optimizer.param_groups[0]['params'])
    one_node = cfg.dim in (16, 32, 64, 128)
    assert n_seen == (1 if (flat and one_node) else len(sd0)), n_seen
    assert set(model.state_dict().keys()) == set(sd0.keys())
    ema = EMA(model, decay=0.999) if use_ema else None
    data = _batch_from(g, dev)
    losses, norms = [], []
    model.train()
    for lr in g['lrs']:
        for grp in optimizer.param_groups:
            grp['lr'] = float(lr)
        optimizer.zero_grad()
        output = model(data)
        loss = LOSSES[kind](output, data.y)
        losses.append(loss.item())
        loss.backward()
        if max_norm is not None:
            norms.append(float(clip_grad_norm_(model.parameters(), max_norm=max_norm, norm_type=2)))
        else:
            norms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))))
        optimizer.step()
        if ema is not None:
            ema(model)

    def ok(a, k, tol):
        e, floor = _rel(a, g[k + '64']), _rel(g[k + '32'], g[k + '64'])
        return e <= max(tol, 2 * floor), (k, e, floor)

    for a, k in ((losses, 'loss'), (norms, 'grad_norm')):
        good, info = ok(a, k, 1e-5)
        assert good, info
    cur = {k: p.detach().cpu().double() for k, p in model.state_dict().items()}
    assert set(cur) == set(sd0)
    l2 = lambda d: float(torch.sqrt(sum((v ** 2).sum() for v in d.values())))
    checks = [(l2(cur), 'param_l2_'), (l2({k: cur[k] - sd0[k].double() for k in cur}), 'delta_l2_')]
    if ema is not None:
        checks.append((l2({k: v.cpu().double() for k, v in ema.shadow.items()}), 'shadow_l2_'))
    for val, k in checks:
        good, info = ok([val], k, 1e-5)
        assert good, info
    if ema is not None:
        model.eval()
        with torch.no_grad():
            ema.assign(model)
            out_ema = model(data).cpu().numpy()
            ema.resume(model)
        good, info = ok(out_ema, 'out_ema', 1e-5)
        assert good, info

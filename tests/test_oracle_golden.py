"""Pin the CPU oracle (oracle/pamnet_oracle.py) against golden vectors produced by the reference's own code
(tests/golden/gen/gen_golden.py).  CPU only.

Tolerances (max-normalised error, max|a-b|/max|b|):
  fp64 oracle vs fp64 reference : 1e-9   (same algorithm, different summation order / basis evaluation route)
  fp32 oracle vs fp32 reference : 1e-5   (the north-star tolerance; the reference's own fp32 noise floor vs its fp64
                                          run is ~2e-6)
Index lists: bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import maxnorm_err
from oracle import pamnet_oracle as O


def _cfg(g):
    cfg = O.Config(dataset=str(g['cfg_dataset']), dim=int(g['cfg_dim']), n_layer=int(g['cfg_n_layer']),
                   cutoff_l=float(g['cfg_cutoff_l']), cutoff_g=float(g['cfg_cutoff_g']), flow=str(g['cfg_flow']))
    if 'cfg_basis' in g.files:             # PAMNet(config, num_spherical, num_radial, envelope_exponent), models.py:22
        cfg.basis = tuple(int(v) for v in g['cfg_basis'])
    return cfg


def _inputs(g):
    t = lambda k: torch.from_numpy(g['in/' + k]) if ('in/' + k) in g.files else None
    return t('x'), t('batch'), t('pos'), t('edge_index'), t('y')


def test_star_indices_bit_exact(golden):
    g = golden('star_indices')
    res = O.indices(torch.from_numpy(g['edge_index']), 4)
    names = ['idx_i', 'idx_j', 'idx_k', 'idx_kj', 'idx_ji', 'idx_i_pair', 'idx_j1_pair', 'idx_j2_pair',
             'idx_jj_pair', 'idx_ji_pair']
    for n, r in zip(names, res):
        assert np.array_equal(r.numpy(), g[n]), n
    # the worked example of SURVEY.md section 4
    assert g['idx_kj'].tolist() == [4, 5, 0, 5, 0, 4]
    assert g['idx_ji_pair'].tolist() == [0, 0, 0, 1, 2, 3, 4, 4, 4, 5, 5, 5]


def test_basis_constants(golden):
    g = golden('basis_tables')
    k = O.basis_constants()
    assert np.array_equal(k['zeros'], g['zeros'])                  # fp32-rounded zeros: bit-exact
    # the reference evaluates the normaliser on float32 scalars under numpy 2 (utils/sbf.py:47) -> 1e-7-class noise
    assert np.max(np.abs(k['norm'] / g['norm'] - 1)) < 3e-7


def test_basis_tables(golden):
    g = golden('basis_tables')
    dist, ang = torch.from_numpy(g['dist']), torch.from_numpy(g['angle'])
    idx = torch.arange(dist.numel())
    c = float(g['cutoff'])
    sbf = O.spherical_basis(dist, ang, idx, c)
    # compare per l-block: magnitudes differ by orders between blocks
    for l in range(7):
        blk = slice(6 * l, 6 * l + 6)
        assert maxnorm_err(sbf[:, blk], g['sbf64'][:, blk]) < 1e-6, l      # bounded by the 3e-7 normaliser noise
    freq = torch.arange(1, 17, dtype=torch.float32) * np.pi        # the layer's fp32 parameter (basic.py:69-72)
    assert maxnorm_err(O.bessel_rbf(dist, freq.double(), c), g['rbf64']) < 1e-12
    assert maxnorm_err(O.bessel_rbf(dist.float(), freq, c), g['rbf32']) < 1e-6


@pytest.mark.parametrize('name,small', [('qm9_d32_l2', False), ('pdbbind_d32_l2', False), ('qm9s_d32_l2', True),
                                        ('qm9_ragged_d32_l2', False), ('qm9s_ragged_d32_l2', True),
                                        ('qm9_flow_t2s_d32_l2', False), ('pdbbind_flow_t2s_d32_l2', False),
                                        ('pdbbind_d128_l3', False), ('qm9s_d128_l2', True),
                                        ('qm9_d128_l6', False), ('qm9_basis_5x4_p6_d32_l2', False),
                                        ('qm9_basis_8x7_p4_d128_l2', False)])
def test_random_init_forward(golden, name, small):
    g = golden(name)
    cfg = _cfg(g)
    sd32 = O.init_state_dict(cfg, seed=int(g['seed']), small=small)
    assert abs(sum(float(v.double().abs().sum()) for v in sd32.values()) - float(g['weights_checksum'])) < 1e-6
    x, batch, pos, ei, y = _inputs(g)
    fwd = O.pamnet_s_forward if small else O.pamnet_forward
    for tag, dt, tol in (('64', torch.float64, 1e-6), ('32', torch.float32, 1e-5)):
        sd = {k: v.to(dt) for k, v in sd32.items()}
        inter = {}
        xin = x.to(dt) if cfg.dataset == 'PDBbind' else x
        out = fwd(sd, cfg, xin, batch, pos, ei, dtype=dt, intermediates=inter)
        if cfg.dataset == 'PDBbind':
            # complex - pocket - ligand cancels ~1000x (|out| ~1e-2 from summands totalling ~9): the reference's own
            # fp32 run is 4e-5 (max-normalised) away from its fp64 run.  Parity is relative to the summed magnitude.
            scale = max(float(np.abs(g['node_out64'][g['in/batch'] == b]).sum()) for b in range(len(g['out64'])))
            assert float(np.max(np.abs(out.numpy() - g['out' + tag]))) / scale < tol, (tag, 'out')
        else:
            assert maxnorm_err(out, g['out' + tag]) < tol, (tag, 'out')
        assert maxnorm_err(inter['pool_in'], g['node_out' + tag]) < tol, (tag, 'node_out')   # golden = pool input
        if ('x_layers' + tag) in g.files:
            assert maxnorm_err(inter['x_layers'], g['x_layers' + tag]) < tol, (tag, 'x_layers')
        if tag == '32' and not small:
            assert inter['edge_index_l'].shape[1] == int(g['num_edges_l'])
            assert inter['idx_kj'].numel() == int(g['num_triplets'])
            assert inter['idx_jj_pair'].numel() == int(g['num_pairs'])
            if 'ref/idx_kj' in g.files:
                for k in ('idx_kj', 'idx_ji', 'idx_jj_pair', 'idx_ji_pair'):
                    assert np.array_equal(inter[k].numpy(), g['ref/' + k]), k


@pytest.mark.parametrize('name', ['qm9_d32_l2', 'pdbbind_d32_l2', 'qm9_basis_5x4_p6_d32_l2', 'qm9_flow_t2s_d32_l2',
                                  'pdbbind_flow_t2s_d32_l2'])
def test_loss_gradient_fp64(golden, name):
    """d L1-loss / d params through the oracle == through the reference (fp64)."""
    g = golden(name)
    cfg = _cfg(g)
    sd = O.as_params({k: v.double() for k, v in O.init_state_dict(cfg, seed=int(g['seed'])).items()})
    x, batch, pos, ei, y = _inputs(g)
    xin = x.double() if cfg.dataset == 'PDBbind' else x
    out = O.pamnet_forward(sd, cfg, xin, batch, pos, ei, dtype=torch.float64)
    loss = torch.nn.functional.l1_loss(out, y.double())
    loss.backward()
    assert abs(loss.item() - float(g['loss64'])) < 1e-7 * max(1.0, abs(float(g['loss64'])))
    gn = float(torch.sqrt(sum((p.grad ** 2).sum() for p in sd.values() if p.grad is not None)))
    assert abs(gn / float(g['grad_norm64']) - 1) < 1e-6
    for k in g.files:
        if k.startswith('grad64/'):
            assert maxnorm_err(sd[k[7:]].grad, g[k]) < 1e-6, k


def test_rna_checkpoint_end_to_end(golden):
    """Shipped RNA-Puzzles graphs + shipped checkpoint (the only real end-to-end golden in the reference tree)."""
    g = golden('rna_native')
    cfg = O.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    sd32 = {k: torch.from_numpy(g['ckpt/' + k]) for k in g['ckpt_keys'].tolist()}
    assert len(sd32) == 74 and sum(v.numel() for v in sd32.values()) == 11714
    # survey probe values, fp32, one graph per batch
    assert abs(float(g['g6/out32'][0]) - 2.912703) < 2e-6 and abs(float(g['g4/out32'][0]) - 2.248216) < 2e-6
    for gid in (6, 4, 17):
        x = torch.from_numpy(g['g%d/x' % gid])
        batch = torch.zeros(x.size(0), dtype=torch.long)
        for tag, dt, tol in (('64', torch.float64, 1e-7), ('32', torch.float32, 1e-5)):
            sd = {k: v.to(dt) for k, v in sd32.items()}
            inter = {}
            out = O.pamnet_forward(sd, cfg, x.to(dt), batch, dtype=dt, intermediates=inter)
            assert maxnorm_err(out, g['g%d/out%s' % (gid, tag)]) < tol, (gid, tag)
            assert inter['edge_index_l'].shape[1] == int(g['g%d/num_edges_l' % gid])
            assert inter['idx_kj'].numel() == int(g['g%d/num_triplets' % gid])
            if gid == 6:
                assert maxnorm_err(inter['x_layers'], g['g6/x_layers' + tag]) < tol
                assert maxnorm_err(inter['node_out'], g['g6/node_out' + tag]) < tol
    # two graphs in one batch == the same graphs alone (independent units)
    x = torch.cat([torch.from_numpy(g['g4/x']), torch.from_numpy(g['g6/x'])])
    batch = torch.cat([torch.zeros(g['g4/x'].shape[0], dtype=torch.long), torch.ones(g['g6/x'].shape[0], dtype=torch.long)])
    out = O.pamnet_forward(sd32, cfg, x, batch)
    assert maxnorm_err(out, g['batched_4_6_out32']) < 1e-5


def test_rna_second_loader_batch_of_the_shipped_set(golden):
    """inference_rna_puzzles.py:55-66 scores all 21 shipped structures in two DataLoader batches (16 + 5, batch_size=16,
    shuffle=False).  The oracle on the second batch (5 graphs, 9 908 nodes) against the reference's own fp32 / fp64 outputs and
    the integer sizes of its graphs (the 16-graph batch runs on the GPU side: tests/test_hip_model.py)."""
    g, ck = golden('rna_native_all'), golden('rna_native')
    cfg = O.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    sd32 = {k: torch.from_numpy(ck['ckpt/' + k]) for k in ck['ckpt_keys'].tolist()}
    ptr = g['node_ptr']
    assert len(ptr) == 22 and int(g['batch_size']) == 16
    assert np.array_equal(np.diff(ptr), ck['all_num_nodes'])
    x = torch.from_numpy(g['x_all'][ptr[16]:ptr[21]])
    batch = torch.from_numpy(np.repeat(np.arange(5), np.diff(ptr)[16:21]))
    for tag, dt, tol in (('64', torch.float64, 1e-7), ('32', torch.float32, 1e-5)):
        inter = {}
        out = O.pamnet_forward({k: v.to(dt) for k, v in sd32.items()}, cfg, x.to(dt), batch, dtype=dt, intermediates=inter)
        assert maxnorm_err(out, g['out' + tag][16:21]) < tol, tag
        assert inter['edge_index_l'].shape[1] == int(g['batch_sizes'][1, 1])
        assert inter['idx_kj'].numel() == int(g['batch_sizes'][1, 2]) and inter['idx_jj_pair'].numel() == int(g['batch_sizes'][1, 3])
    # batched == alone (the per-graph fp32 scores of rna_native.npz: graphs are independent units)
    assert maxnorm_err(g['out32'], ck['all_out32']) < 2e-6


def test_unknown_dataset_raises():
    cfg = O.Config(dataset='nope', dim=8, n_layer=1, cutoff_l=2.0, cutoff_g=5.0)
    with pytest.raises(ValueError):
        O.build_graph(cfg, torch.zeros(3), torch.zeros(3, dtype=torch.long), torch.zeros(3, 3), torch.zeros(2, 0, dtype=torch.long))


def _baseline_batch(name):
    """The inputs of a baseline_* fixture, regenerated from pamnet_amd.synth (the fixture stores outputs + a checksum)."""
    from pamnet_amd import synth
    return {'baseline_qm9_b32': lambda: synth.qm9_batch(0, 0, 32),
            'baseline_qm9_b128': lambda: synth.qm9_batch(0, 0, 128),
            'baseline_pdbbind_b8': lambda: synth.collate([synth.pdbbind_complex(1, i) for i in range(8)]),
            'baseline_pdbbind_b32': lambda: synth.collate([synth.pdbbind_complex(1, i) for i in range(32)]),
            'baseline_rna_b8': lambda: synth.rna_batch(2, 0, 8)}[name]()


@pytest.mark.parametrize('name', ['baseline_qm9_b32', 'baseline_pdbbind_b8'])
def test_oracle_at_baseline_sizes_vs_reference_runs(golden, name):
    """The oracle against the reference ITSELF at BASELINE.json batch sizes (configs[0]: QM9 d=128 L=6 B=32; one
    8-complex shard of configs[3]); the larger fixtures (B=128, PDBbind B=32, RNA B=8) are checked against the HIP path
    on the GPU box (tests/test_hip_model.py::test_baseline_sizes_vs_reference_runs)."""
    g = golden(name)
    cfg = _cfg(g)
    b = _baseline_batch(name)
    assert b.x.size(0) == int(g['num_nodes']) and abs(float(b.x.double().abs().sum()) - float(g['x_checksum'])) < 1e-6
    sd32 = O.init_state_dict(cfg, seed=int(g['seed']))
    assert abs(sum(float(v.double().abs().sum()) for v in sd32.values()) - float(g['weights_checksum'])) < 1e-6
    torch.set_num_threads(8)
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    for tag, dt, tol in (('64', torch.float64, 1e-6), ('32', torch.float32, 1e-5)):
        sd = {k: v.to(dt) for k, v in sd32.items()}
        inter = {}
        xin = b.x.to(dt) if cfg.dataset == 'PDBbind' else b.x
        with torch.no_grad():
            out = O.pamnet_forward(sd, cfg, xin, b.batch, pos, ei, dtype=dt, intermediates=inter)
        if cfg.dataset == 'PDBbind':
            scale = max(float(np.abs(g['node_out64'][b.batch.numpy() == k]).sum()) for k in range(len(g['out64'])))
            assert float(np.max(np.abs(out.numpy() - g['out' + tag]))) / scale < tol, (tag, 'out')
        else:
            assert maxnorm_err(out, g['out' + tag]) < tol, (tag, 'out')
        assert maxnorm_err(inter['pool_in'], g['node_out' + tag]) < tol, (tag, 'node_out')
        if tag == '32':
            assert inter['edge_index_l'].shape[1] == int(g['num_edges_l'])
            assert inter['idx_kj'].numel() == int(g['num_triplets'])
            assert inter['idx_jj_pair'].numel() == int(g['num_pairs'])


def _wide_batch(name):
    from pamnet_amd import synth
    return {'wide_qm9_d192_l2': lambda: synth.qm9_batch(21, 0, 16),
            'wide_qm9s_d136_l2': lambda: synth.qm9_batch(22, 0, 16),
            'wide_pdbbind_d160_l2': lambda: synth.pdbbind_batch(7, 0, 2, n_pocket=70, n_ligand=14),
            'wide_rna_d144_l1': lambda: synth.rna_batch(5, 0, 2, n_nodes=150)}[name]()


@pytest.mark.parametrize('name', ['wide_qm9_d192_l2', 'wide_qm9s_d136_l2', 'wide_pdbbind_d160_l2', 'wide_rna_d144_l1'])
def test_oracle_at_hidden_sizes_above_128_vs_reference_runs(golden, name):
    """Hidden sizes above 128 (models.py:25: any `dim`) -- the widths the HIP path runs layer by layer on csrc/dense.hip --
    pinned to runs of the REFERENCE ITSELF (gen_golden.py --wide-only): the oracle's outputs and pooled node values against
    the reference's fp64 / fp32 runs, graph sizes exact, and the oracle's fp64 loss gradient against the reference's (loss,
    global norm, every parameter's L2 norm, the stored full tensors)."""
    g = golden(name)
    cfg = _cfg(g)
    small = 'qm9s' in name
    b = _wide_batch(name)
    assert b.x.size(0) == int(g['num_nodes']) and abs(float(b.x.double().abs().sum()) - float(g['x_checksum'])) < 1e-6
    sd32 = O.init_state_dict(cfg, seed=int(g['seed']), small=small)
    assert abs(sum(float(v.double().abs().sum()) for v in sd32.values()) - float(g['weights_checksum'])) < 1e-6
    fwd = O.pamnet_s_forward if small else O.pamnet_forward
    torch.set_num_threads(8)
    pos, ei = getattr(b, 'pos', None), getattr(b, 'edge_index', None)
    for tag, dt, tol in (('64', torch.float64, 1e-6), ('32', torch.float32, 2e-5)):
        sd = {k: v.to(dt) for k, v in sd32.items()}
        inter = {}
        xin = b.x.to(dt) if cfg.dataset == 'PDBbind' else b.x
        with torch.no_grad():
            out = fwd(sd, cfg, xin, b.batch, pos, ei, dtype=dt, intermediates=inter)
        if cfg.dataset == 'PDBbind':
            scale = max(float(np.abs(g['node_out64'][b.batch.numpy() == k]).sum()) for k in range(len(g['out64'])))
            assert float(np.max(np.abs(out.numpy() - g['out' + tag]))) / scale < tol, (tag, 'out')
        else:
            assert maxnorm_err(out, g['out' + tag]) < tol, (tag, 'out')
        assert maxnorm_err(inter['pool_in'], g['node_out' + tag]) < tol, (tag, 'node_out')
        if tag == '32' and not small:              # (the small model's oracle keeps no index intermediates)
            assert inter['edge_index_l'].shape[1] == int(g['num_edges_l'])
            assert inter['idx_jj_pair'].numel() == int(g['num_pairs'])
            assert inter['idx_kj'].numel() == int(g['num_triplets'])
    p64 = O.as_params({k: v.double() for k, v in sd32.items()})
    x64 = b.x if cfg.dataset == 'QM9' else b.x.double()
    loss = torch.nn.functional.l1_loss(fwd(p64, cfg, x64, b.batch, pos, ei, dtype=torch.float64), b.y.double())
    loss.backward()
    assert abs(float(loss) - float(g['loss64'])) < 1e-9 * max(1.0, abs(float(g['loss64'])))
    for k, l2 in zip(g['grad_keys'].tolist(), g['grad_l2_64']):
        assert abs(float(p64[k].grad.norm()) - float(l2)) <= 1e-7 * max(float(l2), 1e-30) + 1e-14, k
    for k in g.files:
        if k.startswith('grad64/'):
            assert maxnorm_err(p64[k[7:]].grad.numpy(), g[k]) < 1e-6, k

"""Data formats / host helpers either side of the hot path: the TU text reader, the batch
loader, EMA (API of utils/ema.py) and the PDBbind metrics (utils/metrics.py).  CPU only."""
import os

import numpy as np
import pytest
import torch


def _write_tu(root, name, graphs, labels_one_hot=False):
    raw = os.path.join(root, name, 'raw')
    os.makedirs(raw)
    gi, attrs, labs, ys = [], [], [], []
    for g, (xyz, lab, y) in enumerate(graphs, start=1):
        gi += [g] * len(xyz)
        attrs += ['%.3f, %.3f, %.3f' % tuple(r) for r in xyz]
        labs += [str(int(v)) for v in lab]
        ys.append('%.3f' % y)
    for part, rows in (('graph_indicator', [str(v) for v in gi]), ('node_attributes', attrs), ('node_labels', labs),
                       ('graph_labels', ys)):
        with open(os.path.join(raw, '%s_%s.txt' % (name, part)), 'w') as fh:
            fh.write('\n'.join(rows) + '\n')


def test_tu_reader_roundtrip_rna_fixture(golden, tmp_path):
    """The shipped RNA graphs (coordinates + atom labels from the reference's data files, tests/golden/rna_native.npz)
    written in TU text form and read back: x = [xyz | label] fp32, graph slicing, loader batches feed the model schema."""
    from datasets import DataLoader, TUDataset
    g = golden('rna_native')
    gids = [int(k.split('/')[0][1:]) for k in g.files if k.endswith('/x') and k.startswith('g')]
    graphs = [(g['g%d/x' % i][:, :3], g['g%d/x' % i][:, 3], 0.0) for i in sorted(gids)]
    _write_tu(str(tmp_path), 'rna_native', graphs)
    ds = TUDataset(str(tmp_path), name='rna_native', use_node_attr=True)
    assert len(ds) == len(graphs) and repr(ds) == 'rna_native(%d)' % len(graphs)
    assert ds.num_node_labels == 0 and ds.num_node_attributes == 4        # labels 0/1/2 are not a one-hot block
    for k, (xyz, lab, _) in enumerate(graphs):
        d = ds[k]
        assert d.num_nodes == len(xyz) and d.x.dtype == torch.float32
        assert np.allclose(d.x[:, :3].numpy(), np.round(xyz, 3), atol=5e-4) and np.array_equal(d.x[:, 3].numpy(), lab)
    b = next(iter(DataLoader(ds, batch_size=2, shuffle=False)))
    assert b.num_graphs == 2 and b.x.shape == (len(graphs[0][0]) + len(graphs[1][0]), 4)
    assert torch.equal(b.batch, torch.cat([torch.zeros(len(graphs[0][0])), torch.ones(len(graphs[1][0]))]).long())
    assert b.y.shape == (2,)
    assert len(DataLoader(ds, batch_size=2)) == (len(graphs) + 1) // 2
    torch.manual_seed(3)
    sh = ds.shuffle()
    assert sorted(d.num_nodes for d in sh) == sorted(len(x) for x, _, _ in graphs) and len(ds[1:]) == len(graphs) - 1


def test_tu_reader_label_block_and_errors(tmp_path):
    from datasets import TUDataset, read_tu_data
    raw = os.path.join(str(tmp_path), 'toy', 'raw')
    os.makedirs(raw)
    open(os.path.join(raw, 'toy_graph_indicator.txt'), 'w').write('1\n1\n2\n2\n2\n')
    open(os.path.join(raw, 'toy_node_attributes.txt'), 'w').write('0.5, 1.5\n1.0, 2.0\n0.0, 0.0\n3.0, 1.0\n2.0, 2.0\n')
    open(os.path.join(raw, 'toy_node_labels.txt'), 'w').write('0\n1\n1\n0\n1\n')
    open(os.path.join(raw, 'toy_graph_labels.txt'), 'w').write('1\n-1\n')
    full = TUDataset(str(tmp_path), 'toy', use_node_attr=True)
    assert full.num_node_labels == 0 and full[0].x.shape == (2, 3) and float(full[1].y) == -1.0
    # x = [a0, a1, label]: the label column alone is 0/1 but does not sum to 1 per row -> not a one-hot block
    open(os.path.join(raw, 'toy_node_attributes.txt'), 'w').write('0.5, 1\n1.0, 0\n0.0, 0\n3.0, 1\n2.0, 0\n')
    onehot = TUDataset(str(tmp_path), 'toy', use_node_attr=False)          # columns (a1, label) form a one-hot pair
    assert onehot.num_node_labels == 2 and onehot.num_node_attributes == 1 and onehot[1].x.shape == (3, 2)
    d = read_tu_data(raw, 'toy')
    assert d['node_ptr'].tolist() == [0, 2, 5]
    open(os.path.join(raw, 'toy_graph_indicator.txt'), 'w').write('2\n1\n2\n2\n2\n')
    with pytest.raises(ValueError):
        read_tu_data(raw, 'toy')
    with pytest.raises(FileNotFoundError):
        read_tu_data(raw, 'absent')


def test_metrics_closed_form():
    from sklearn.linear_model import LinearRegression
    from utils import mae, pearson, rmse, sd
    rng = np.random.default_rng(0)
    y = rng.normal(6.0, 2.0, 200)
    f = 0.7 * y + rng.normal(0, 1.0, 200) + 1.0
    assert abs(rmse(y, f) - np.sqrt(np.mean((y - f) ** 2))) < 1e-12
    assert abs(mae(y, f) - np.mean(np.abs(y - f))) < 1e-12
    assert abs(pearson(y, f) - np.corrcoef(y, f)[0, 1]) < 1e-12
    lr = LinearRegression().fit(f.reshape(-1, 1), y.reshape(-1, 1))         # the reference's formulation (metrics.py:14-20)
    ref = (((y.reshape(-1, 1) - lr.predict(f.reshape(-1, 1))) ** 2).sum() / (len(y) - 1)) ** 0.5
    assert abs(sd(y, f) - ref) < 1e-10


def test_ema_api_matches_reference_law():
    from utils import EMA
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 1))
    ema = EMA(m, decay=0.999)
    shadow = {k: p.detach().clone() for k, p in m.named_parameters()}
    for step in range(3):
        with torch.no_grad():
            for p in m.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        ema(m) if step else ema(m, num_updates=5)
        d = min(0.999, (1.0 + (99999 if step else 5)) / (10.0 + (99999 if step else 5)))      # utils/ema.py:14
        for k, p in m.named_parameters():
            shadow[k] = (1 - d) * p.detach() + d * shadow[k]
    for k in shadow:
        assert torch.allclose(ema.shadow[k], shadow[k], rtol=1e-6, atol=1e-7)
    live = {k: p.detach().clone() for k, p in m.named_parameters()}
    ema.assign(m)
    assert all(torch.equal(p, ema.shadow[k]) for k, p in m.named_parameters())
    assert all(torch.equal(ema.original[k], live[k]) for k in live)
    ema.resume(m)
    assert all(torch.equal(p, live[k]) for k, p in m.named_parameters())


@pytest.mark.gpu
def test_inference_driver_path_tu_to_model(golden, tmp_path):
    """inference_rna_puzzles.py:44-66 end to end on the box: TU files -> TUDataset(use_node_attr=True) -> DataLoader ->
    PAMNet with the shipped checkpoint -> scores, against the reference's outputs for graphs 4, 6, 17."""
    import models
    from datasets import DataLoader, TUDataset
    from conftest import maxnorm_err
    g = golden('rna_native')
    order = (6, 4, 17)
    _write_tu(str(tmp_path), 'rna_native', [(g['g%d/x' % i][:, :3], g['g%d/x' % i][:, 3], 0.0) for i in order])
    ds = TUDataset(str(tmp_path), name='rna_native', use_node_attr=True)
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    model = models.PAMNet(cfg)
    model.load_state_dict({k: torch.from_numpy(g['ckpt/' + k]) for k in g['ckpt_keys'].tolist()}, strict=True)
    model = model.to('cuda:0').eval()
    scores = []
    with torch.no_grad():
        for data in DataLoader(ds, batch_size=2, shuffle=False):
            scores += model(data.to('cuda:0')).reshape(-1).tolist()
    ref64 = np.array([float(g['g%d/out64' % i][0]) for i in order])
    ref32 = np.array([float(g['g%d/out32' % i][0]) for i in order])
    assert maxnorm_err(np.array(scores), ref64) <= max(1e-5, 2 * maxnorm_err(ref32, ref64))


@pytest.mark.gpu
def test_all_21_shipped_rna_structures_as_the_inference_driver_scores_them(golden, tmp_path):
    """inference_rna_puzzles.py:46-66 on the whole shipped set: 21 graphs (841 .. 3 823 nodes), DataLoader batch_size=16 without
    shuffling -> a batch of 16 and one of 5, the shipped checkpoint.  (a) plain tensors, (b) the resident store (device-side
    collation, one-call graph), (c) the three smallest through TU text files -> datasets.TUDataset -> DataLoader.  21 / 21 scores
    within the parity bound of the reference's fp64 run (never tighter than its own fp32 noise), integer graph sizes exact."""
    import models
    from datasets import DataLoader, TUDataset
    from pamnet_amd.store import MoleculeStore
    from pamnet_amd import synth
    from conftest import maxnorm_err
    g, ck = golden('rna_native_all'), golden('rna_native')
    dev = torch.device('cuda:0')
    cfg = models.Config(dataset='rna_native', dim=16, n_layer=1, cutoff_l=2.6, cutoff_g=20.0, flow='target_to_source')
    model = models.PAMNet(cfg)
    model.load_state_dict({k: torch.from_numpy(ck['ckpt/' + k]) for k in ck['ckpt_keys'].tolist()}, strict=True)
    model = model.to(dev).eval()
    ptr, bs = g['node_ptr'], int(g['batch_size'])
    n_graphs = len(ptr) - 1
    assert n_graphs == 21 and bs == 16
    ref32, ref64 = g['out32'], g['out64']
    bound = max(1e-5, 2 * maxnorm_err(ref32, ref64))
    xs = [g['x_all'][ptr[k]:ptr[k + 1]] for k in range(n_graphs)]
    # (a) plain tensors, the loader's two batches
    scores = []
    with torch.no_grad():
        for b, b0 in enumerate(range(0, n_graphs, bs)):
            b1 = min(b0 + bs, n_graphs)
            data = synth.Batch()
            data.x = torch.from_numpy(np.concatenate(xs[b0:b1])).to(dev)
            data.batch = torch.from_numpy(np.repeat(np.arange(b1 - b0), [len(x) for x in xs[b0:b1]])).to(dev)
            data.num_graphs = b1 - b0
            scores += model(data).reshape(-1).tolist()
            gc = model._graph_cache
            assert (gc.loc.m, gc.n_trip, gc.n_pair) == tuple(int(v) for v in g['batch_sizes'][b, 1:4]), b
    assert len(scores) == 21 and maxnorm_err(np.array(scores), ref64) <= bound, (maxnorm_err(np.array(scores), ref64), bound)
    # (b) the resident store, the same two selections
    st = MoleculeStore([dict(x=x, y=np.float32(0)) for x in xs], dev).prepare_for(model)
    with torch.no_grad():
        s2 = []
        for b0 in range(0, n_graphs, bs):
            s2 += model(st.collate(list(range(b0, min(b0 + bs, n_graphs))))).reshape(-1).tolist()
    model.verify()
    assert maxnorm_err(np.array(s2), ref64) <= bound
    # (c) TU text files of the three smallest (coordinates are written with 3 decimals, as the shipped files hold them)
    small = sorted(range(n_graphs), key=lambda k: len(xs[k]))[:3]
    _write_tu(str(tmp_path), 'rna_native', [(xs[k][:, :3], xs[k][:, 3], 0.0) for k in small])
    ds = TUDataset(str(tmp_path), name='rna_native', use_node_attr=True)
    with torch.no_grad():
        s3 = []
        for data in DataLoader(ds, batch_size=16, shuffle=False):
            s3 += model(data.to(dev)).reshape(-1).tolist()
    assert maxnorm_err(np.array(s3), ref64[small]) <= bound


@pytest.mark.gpu
def test_driver_loop_example_runs(tmp_path):
    """examples/main_qm9_synth.py (the reference's main_qm9.py loop on synthetic molecules): two short epochs at a small
    configuration train (the loss falls), evaluate under EMA and save a reference-layout state_dict."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = os.path.join(str(tmp_path), 'best_model.h5')
    out = subprocess.run([sys.executable, os.path.join(repo, 'examples', 'main_qm9_synth.py'), '--epochs', '2', '--train',
                          '256', '--val', '64', '--batch_size', '32', '--n_layer', '2', '--lr', '1e-3', '--save', ck],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('Epoch')]
    assert len(lines) == 2
    maes = [float(l.split('Train MAE:')[1].split(',')[0]) for l in lines]
    assert maes[1] < maes[0]
    sd = torch.load(ck, map_location='cpu')
    import models
    m = models.PAMNet(models.Config(dataset='QM9', dim=128, n_layer=2, cutoff_l=5.0, cutoff_g=5.0))
    m.load_state_dict(sd, strict=True)

"""GPU parity tests of the individual C-ABI kernels against the CPU oracle / plain torch on seeded inputs.
Integer / index work: bit-exact.  fp32 work: tolerance stated per test.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import maxnorm_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs an MI355X'
    from pamnet_amd import lib
    lib.load()                                    # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


def _csr(rng, rows, max_len, dev):
    lens = rng.integers(0, max_len + 1, size=rows)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return torch.from_numpy(ptr).to(dev), int(ptr[-1]), torch.from_numpy(np.repeat(np.arange(rows), lens)).to(dev)


@pytest.mark.parametrize('d', [4, 16, 32, 128, 256, 48])
@pytest.mark.parametrize('rows,max_len', [(1, 0), (7, 3), (1000, 20), (333, 70)])
def test_segment_sum_plain(dev, d, rows, max_len):
    """pamnet_segment_sum_f32 == torch_scatter.scatter(src, sorted index, reduce='add') semantics."""
    from pamnet_amd import ops
    rng = np.random.default_rng(rows * 131 + d)
    ptr, m, seg = _csr(rng, rows, max_len, dev)
    src = torch.randn(m, d, device=dev)
    out = torch.full((rows, d), float('nan'), device=dev)
    ops.segment_sum_raw(out, None, src, None, None, None, None, ptr, rows, d)
    ref = torch.zeros(rows, d, dtype=torch.float64, device=dev).index_add_(0, seg, src.double())
    assert torch.isfinite(out).all()                  # empty segments are written as zeros
    assert maxnorm_err(out.cpu(), ref.cpu()) < 2e-6 or m == 0
    out2 = torch.empty_like(out)
    ops.segment_sum_raw(out2, None, src, None, None, None, None, ptr, rows, d)
    assert torch.equal(out, out2)                     # deterministic, run-to-run bitwise


@pytest.mark.parametrize('d', [16, 128])
def test_segment_sum_all_operands(dev, d):
    from pamnet_amd import ops
    rng = np.random.default_rng(5)
    rows, ra, rb = 500, 300, 200
    ptr, m, seg = _csr(rng, rows, 12, dev)
    A, B, init = torch.randn(ra, d, device=dev), torch.randn(rb, d, device=dev), torch.randn(rows, d, device=dev)
    ia = torch.from_numpy(rng.integers(0, ra, m).astype(np.int32)).to(dev)
    ib = torch.from_numpy(rng.integers(0, rb, m).astype(np.int32)).to(dev)
    perm = torch.from_numpy(rng.permutation(m).astype(np.int32)).to(dev)
    out = torch.empty(rows, d, device=dev)
    ops.segment_sum_raw(out, init, A, ia, B, ib, perm, ptr, rows, d)
    k = perm.long()
    terms = A.double()[ia.long()[k]] * B.double()[ib.long()[k]]
    ref = init.double().clone().index_add_(0, seg, terms)
    assert maxnorm_err(out.cpu(), ref.cpu()) < 2e-6
    # gather_mul
    g = torch.empty(m, d, device=dev)
    ops.gather_mul_raw(g, A, ia, B, ib, m, d)
    assert torch.equal(g, A[ia.long()] * B[ib.long()])


@pytest.mark.parametrize('rows,max_len', [(1, 0), (37, 9), (2286, 30), (20000, 6)])
def test_segment_sum_multi_and_gather_mul2(dev, rows, max_len):
    """Batched plain segment sums (with and without a transposed-CSR permutation) and the two-product gather."""
    import ctypes
    from pamnet_amd import lib
    rng = np.random.default_rng(rows)
    d = 128
    jobs = []
    for j in range(4):
        ptr, m, seg = _csr(rng, rows, max_len, dev)
        A = torch.randn(max(m, 1), d, device=dev)
        perm = torch.from_numpy(rng.permutation(m).astype(np.int32)).to(dev) if j % 2 else None
        jobs.append((ptr, m, seg, A, perm, torch.empty(rows, d, device=dev)))
    P = ctypes.c_void_p * 4
    st = lib.stream_of(jobs[0][3])
    for nj in (1, 2, 4):
        for jb in jobs:
            jb[5].fill_(float('nan'))
        lib.call('pamnet_segment_sum_multi_f32', nj, P(*[lib.ptr(jb[5]) for jb in jobs]), P(*[lib.ptr(jb[3]) for jb in jobs]),
                 P(*[None if jb[4] is None else lib.ptr(jb[4]) for jb in jobs]), P(*[lib.ptr(jb[0]) for jb in jobs]),
                 rows, d, st)
        for ptr, m, seg, A, perm, out in jobs[:nj]:
            src = A[:m].double() if perm is None else A[:m].double()[perm.long()]
            ref = torch.zeros(rows, d, device=dev, dtype=torch.float64).index_add_(0, seg, src)
            assert maxnorm_err(out.cpu(), ref.cpu()) < 2e-6 or float(ref.abs().max()) == 0.0
    m = max(jobs[0][1], 1)
    ia = torch.from_numpy(rng.integers(0, rows, m).astype(np.int32)).to(dev)
    Asrc, B1, B2 = torch.randn(rows, d, device=dev), torch.randn(m, d, device=dev), torch.randn(m, d, device=dev)
    o1, o2 = torch.empty(m, d, device=dev), torch.empty(m, d, device=dev)
    lib.call('pamnet_gather_mul2_f32', lib.ptr(o1), lib.ptr(o2), lib.ptr(Asrc), lib.ptr(ia), lib.ptr(B1), lib.ptr(B2),
             m, d, st)
    assert torch.equal(o1, Asrc[ia.long()] * B1) and torch.equal(o2, Asrc[ia.long()] * B2)


def test_segment_ops_autograd(dev):
    """Backward kernels == autograd of the equivalent torch expression."""
    from pamnet_amd import graph as G, ops
    rng = np.random.default_rng(9)
    rows, ra, d = 200, 150, 32
    ptr, m, seg = _csr(rng, rows, 9, dev)
    col = torch.from_numpy(rng.integers(0, ra, m).astype(np.int32)).to(dev)
    csr = G.CSR(ptr, seg.to(torch.int32), col)
    tr = G.Transpose(col, ra)
    A = torch.randn(ra, d, device=dev, requires_grad=True)
    B = torch.randn(m, d, device=dev, requires_grad=True)
    init = torch.randn(rows, d, device=dev, requires_grad=True)
    w = torch.randn(rows, d, device=dev)
    y = ops.aggregate(ops.gather(A, col, tr.ptr, tr.perm) * B, csr, init=init) + ops.gather_mul_aggregate(A, B, csr, tr)
    (y * w).sum().backward()
    got = [A.grad.clone(), B.grad.clone(), init.grad.clone()]
    A.grad = B.grad = init.grad = None
    t = A[col.long()] * B
    y2 = init + torch.zeros_like(init).index_add(0, seg, t) + torch.zeros_like(init).index_add(0, seg, t)
    (y2 * w).sum().backward()
    assert maxnorm_err(y.detach().cpu(), y2.detach().cpu()) < 2e-6
    for a, b in zip(got, [A.grad, B.grad, init.grad]):
        assert maxnorm_err(a.cpu(), b.cpu()) < 2e-6


@pytest.mark.parametrize('n', [0, 1, 5, 1023, 1024, 1025, 4096, 4097, 65535, 65536, 65537, 100000, 24575, 24576, 24577, 1 << 20, 4096 * 1024, 4096 * 1024 + 1, 5000001])
def test_exclusive_scan_bit_exact(dev, n):
    from pamnet_amd import graph as G
    x = torch.randint(0, 50, (n,), dtype=torch.int32, device=dev)
    out = G.exclusive_scan(x)
    ref = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), x.long().cumsum(0)])
    assert torch.equal(out.long(), ref)


@pytest.mark.parametrize('m,rows', [(0, 3), (10, 1), (5000, 700), (200000, 50), (100000, 100000),
                                    # either side of the single-workgroup form's limits (8 192 keys, 12 288 rows)
                                    (8192, 12288), (8193, 100), (8000, 12289), (33000, 2300), (1, 12288), (4316, 2286),
                                    # either side of the LDS-counter form's limits (>= 131 072 keys, <= 32 768 rows) and the
                                    # RNA batch's shape; a ragged last chunk; one row
                                    (131072, 32768), (131071, 32768), (131072, 32769), (867252, 17699), (140001, 1)])
@pytest.mark.parametrize('order', ['random', 'sorted'])
def test_csr_from_keys_stable(dev, m, rows, order):
    from pamnet_amd import graph as G
    keys = torch.randint(0, rows, (m,), dtype=torch.int32, device=dev)
    if order == 'sorted':
        keys = torch.sort(keys).values.contiguous()          # the identity-permutation path
    ptr, perm = G.csr_from_keys(keys, rows)
    ref_perm = torch.sort(keys.long(), stable=True).indices
    assert torch.equal(perm.long(), ref_perm)
    assert torch.equal(ptr.long(), torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                              torch.bincount(keys.long(), minlength=rows).cumsum(0)]))


@pytest.mark.parametrize('m,rows', [(20000, 900), (200000, 17699)])
def test_csr_from_keys_skips_keys_outside_the_rows(dev, m, rows):
    """Keys outside [0, rows) are counted nowhere and claim no slot (no out-of-bounds write), in the plain histogram and in
    the LDS-counter form alike: ptr is the histogram of the valid keys, the first ptr[rows] entries of perm are the valid
    entries in stable order."""
    from pamnet_amd import graph as G
    torch.manual_seed(m)
    keys = torch.randint(0, rows, (m,), dtype=torch.int32, device=dev)
    bad = torch.rand(m, device=dev) < 0.01
    keys = torch.where(bad, torch.where(torch.rand(m, device=dev) < 0.5, torch.full_like(keys, -1), keys + rows), keys).contiguous()
    ptr, perm = G.csr_from_keys(keys, rows)
    valid = (keys >= 0) & (keys < rows)
    vidx = valid.nonzero().view(-1)
    want = vidx[torch.sort(keys[vidx].long(), stable=True).indices]
    assert torch.equal(ptr.long(), torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                              torch.bincount(keys[vidx].long(), minlength=rows).cumsum(0)]))
    assert int(ptr[-1]) == int(valid.sum())
    assert torch.equal(perm[:int(ptr[-1])].long(), want)


def _edge_set(row, col):
    return set(zip(row.tolist(), col.tolist()))


def test_radius_graph_matches_oracle(dev):
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, synth
    b = synth.qm9_batch(3, 0, 40)
    ei, dist = O.get_edge_info(O.radius_graph(b.pos, b.batch, 5.0), b.pos)
    nodeg = b.batch.to(torch.int32).to(dev)
    gptr, _ = G.csr_from_keys(nodeg, 40)
    ptr, nbr, d = G.radius_graph(b.pos.to(dev), nodeg, gptr, 5.0)
    q = G.expand_rows(ptr, nbr.numel())
    assert _edge_set(q.cpu(), nbr.cpu()) == _edge_set(ei[0], ei[1])                    # bit-exact edge set
    # distances of the same edges (sorted by (row, col) on both sides): fp32, <= 1 ulp apart
    key = (ei[0] * 100000 + ei[1]).argsort()
    assert float(((d.cpu() - dist[key]).abs() / dist[key]).max()) < 1.3e-7


@pytest.mark.parametrize('cap', [1, 7, 40, 64, 65, 200])
def test_radius_neighbour_cap_matches_oracle(dev, cap):
    """max_num_neighbors (models.py:110,128,301): a query keeps its first `cap` hits in ascending index order, itself counted;
    both launch forms (thread per node / wavefront per node, chunk boundaries at 64 on either side of the cap) equal the
    oracle's capped search edge for edge, raise the cap bit exactly when a row was cut, and leave it alone otherwise."""
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, lib
    rng = np.random.RandomState(cap)
    counts = [150, 31, 70, 1]
    n = sum(counts)
    pos = torch.from_numpy((rng.rand(n, 3) * 6).astype(np.float32))            # dense: most pairs within r = 4
    batch = torch.from_numpy(np.repeat(np.arange(len(counts)), counts))
    nodeg, posd = batch.to(torch.int32).to(dev), pos.to(dev)
    gptr, _ = G.csr_from_keys(nodeg, len(counts))
    ref = O.radius_graph(pos, batch, 4.0, cap)
    ref = ref[:, ref[0] != ref[1]]                                             # remove_self_loops (models.py:63)
    full = O.radius_graph(pos, batch, 4.0)
    binds = ref.size(1) != int((full[0] != full[1]).sum())
    for ng in (len(counts), 0):                                                # wavefront form (n / graphs > 96 false here) ...
        for force_wave in (False, True):
            ngv = 1 if force_wave else ng                                      # ... n_graphs = 1 declares one big graph: wave form
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            cnt = torch.empty(n, dtype=torch.int32, device=dev)
            lib.call('pamnet_radius_count_i32', lib.ptr(posd), lib.ptr(nodeg), lib.ptr(gptr), n, ngv, 4.0, cap, lib.ptr(cnt),
                     lib.ptr(flag), lib.stream_of(posd))
            ptr = G.exclusive_scan(cnt)
            tot = int(ptr[-1])
            assert tot == ref.size(1) and bool(int(flag) & G.CAP_BIT) == binds and not int(flag) & ~G.CAP_BIT
            nbr, d, rows = (torch.empty(tot, dtype=torch.int32, device=dev), torch.empty(tot, device=dev),
                            torch.empty(tot, dtype=torch.int32, device=dev))
            lib.call('pamnet_radius_fill_i32', lib.ptr(posd), lib.ptr(nodeg), lib.ptr(gptr), n, ngv, 4.0, cap, lib.ptr(ptr),
                     lib.ptr(nbr), lib.ptr(d), lib.ptr(rows), tot, lib.stream_of(posd))
            assert torch.equal(rows.cpu().long(), ref[0]) and torch.equal(nbr.cpu().long(), ref[1])   # same order, too
    assert binds == (cap < 150)


def test_radius_wave_form_equals_thread_form(dev):
    """Complex-sized graphs (here 3 graphs of 130 / 257 / 64 nodes, average > 96) take the wavefront-per-node search:
    pointer, neighbours (ascending) and distances equal the thread-per-node form (selected by declaring n_graphs
    unknown) bit for bit, and the oracle's edge set."""
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, lib
    rng = np.random.RandomState(3)
    counts = [130, 257, 64]
    n = sum(counts)
    pos = torch.from_numpy((rng.rand(n, 3) * 14).astype(np.float32))
    batch = torch.from_numpy(np.repeat(np.arange(3), counts))
    nodeg = batch.to(torch.int32).to(dev)
    gptr, _ = G.csr_from_keys(nodeg, 3)
    posd = pos.to(dev)
    ptr, nbr, d = G.radius_graph(posd, nodeg, gptr, 4.0)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    lib.call('pamnet_radius_count_i32', lib.ptr(posd), lib.ptr(nodeg), lib.ptr(gptr), n, 0, 4.0, 0, lib.ptr(cnt), None, lib.stream_of(posd))
    ptr2 = G.exclusive_scan(cnt)
    assert torch.equal(ptr, ptr2)
    nbr2, d2 = torch.empty_like(nbr), torch.empty_like(d)
    rows2 = torch.empty_like(nbr)
    lib.call('pamnet_radius_fill_i32', lib.ptr(posd), lib.ptr(nodeg), lib.ptr(gptr), n, 0, 4.0, 0, lib.ptr(ptr2), lib.ptr(nbr2),
             lib.ptr(d2), lib.ptr(rows2), nbr.numel(), lib.stream_of(posd))
    assert torch.equal(nbr, nbr2) and torch.equal(d, d2)
    rows = []
    G.radius_fill(posd, nodeg, gptr, 4.0, ptr, nbr.numel(), rows_out=rows)       # the wavefront form writes the same row ids
    assert torch.equal(rows[0], rows2) and torch.equal(rows2, G.expand_rows(ptr, nbr.numel()))
    ei, _ = O.get_edge_info(O.radius_graph(pos, batch, 4.0), pos)
    q = G.expand_rows(ptr, nbr.numel())
    assert _edge_set(q.cpu(), nbr.cpu()) == _edge_set(ei[0], ei[1])


def test_knn_matches_oracle(dev):
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, synth
    b = synth.rna_batch(1, 0, 2, n_nodes=300)
    pos = b.x[:, :3].contiguous()
    ei = O.knn_graph(pos, b.batch, 50)
    ei, dist = O.get_edge_info(ei, pos)
    nodeg = b.batch.to(torch.int32).to(dev)
    gptr, _ = G.csr_from_keys(nodeg, 2)
    kp, kn, kd = G.knn_table(pos.to(dev), nodeg, gptr, 50, float('inf'))
    q = G.expand_rows(kp, kn.numel())
    keep = kn >= 0
    assert _edge_set(q[keep].cpu(), kn[keep].cpu()) == _edge_set(ei[0], ei[1])
    assert int(keep.sum()) == 49 * 600
    # small graph (fewer nodes than k) and cutoff masking
    b2 = synth.rna_batch(2, 0, 1, n_nodes=20)
    pos2 = b2.x[:, :3].contiguous()
    g1 = torch.zeros(20, dtype=torch.int32, device=dev)
    kp, kn, kd = G.knn_table(pos2.to(dev), g1, torch.tensor([0, 20], dtype=torch.int32, device=dev), 50, 3.0)
    ei2, d2 = O.get_edge_info(O.knn_graph(pos2, b2.batch, 50), pos2)
    m = d2 <= 3.0
    q = G.expand_rows(kp, kn.numel())
    keep = kn >= 0
    assert _edge_set(q[keep].cpu(), kn[keep].cpu()) == _edge_set(ei2[0][m], ei2[1][m])


def test_csr_filter_pair_equals_two_filters(dev):
    """csr_filter2 (both RNA cutoffs, one host round trip) == two csr_filter calls, bit for bit."""
    from pamnet_amd import graph as G, synth
    b = synth.rna_batch(3, 0, 2, n_nodes=200)
    pos = b.x[:, :3].contiguous().to(dev)
    nodeg = b.batch.to(torch.int32).to(dev)
    gptr, _ = G.csr_from_keys(nodeg, 2)
    kp, kn, kd = G.knn_table(pos, nodeg, gptr, 50, float('inf'))
    a, c = G.csr_filter2(kp, kn, kd, 20.0, 2.6)
    for got, cut in ((a, 20.0), (c, 2.6)):
        want = G.csr_filter(kp, kn, kd, cut)
        assert all(torch.equal(x, y) for x, y in zip(got, want))
    assert G.host_ints(kp[-1], torch.tensor(True, device=dev), torch.tensor(7, device=dev)) == [int(kp[-1]), 1, 7]


def test_csr_filter_vs_mask_ragged_rows(dev):
    """Row lengths 0 .. 200 (several 64-entry rounds per row, empty rows, dropped -1 entries): the kept entries and
    their order equal a boolean-mask compaction."""
    from pamnet_amd import graph as G
    rng = np.random.RandomState(9)
    lens = rng.randint(0, 201, size=777)
    lens[[0, 5, 776]] = 0
    lens[[1, 300]] = [64, 128]
    ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    m = int(ptr[-1])
    nbr = torch.from_numpy(rng.randint(-1, 5000, size=m).astype(np.int32)).to(dev)
    dist = torch.from_numpy(rng.rand(m).astype(np.float32) * 10).to(dev)
    row_of = G.expand_rows(ptr, m)
    for cut in (0.0, 3.3, 11.0):
        p2, n2, d2 = G.csr_filter(ptr, nbr, dist, cut)
        keep = (nbr >= 0) & (dist <= cut)
        assert torch.equal(n2, nbr[keep]) and torch.equal(d2, dist[keep])
        cnt = torch.bincount(row_of[keep].long(), minlength=777)
        assert torch.equal(p2.long(), torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), cnt.cumsum(0)]))


@pytest.mark.parametrize('n,lattice', [(300, False), (700, True), (4100, False), (5000, True)])
def test_knn_order_ties_and_large_graphs(dev, n, lattice):
    """The table rows hold the k smallest (squared distance, index) pairs in ascending order: exact integer check
    against a torch sort of the same fp32 squared distances.  Lattice points give many exact ties (broken by index);
    graphs above 4096 nodes take the streaming path, the others the register-cached selection path."""
    from pamnet_amd import graph as G
    gen = torch.Generator().manual_seed(n)
    if lattice:
        pos = torch.randint(0, 7, (n, 3), generator=gen).float()
    else:
        pos = torch.rand(n, 3, generator=gen) * 30
    K = 50
    nodeg = torch.zeros(n, dtype=torch.int32, device=dev)
    gptr = torch.tensor([0, n], dtype=torch.int32, device=dev)
    kp, kn, kd = G.knn_table(pos.to(dev), nodeg, gptr, K, float('inf'))
    p = pos.to(dev)
    dx, dy, dz = (p[:, None, 0] - p[None, :, 0]), (p[:, None, 1] - p[None, :, 1]), (p[:, None, 2] - p[None, :, 2])
    d2 = (dx * dx + dy * dy) + dz * dz                      # same association as the kernel, no fma: exact products
    order = torch.sort(d2, dim=1, stable=True).indices[:, :K]       # (d2, index) lexicographic: stable sort
    want_d = torch.gather(d2, 1, order).sqrt()
    self_idx = torch.arange(n, device=dev)[:, None]
    want_n = torch.where(order == self_idx, torch.full_like(order, -1), order).to(torch.int32)
    assert torch.equal(kn.view(n, K), want_n)
    # the kernel's sqrt is correctly rounded (__fsqrt_rn); torch's device sqrt may differ in the last bit
    assert torch.allclose(kd.view(n, K), want_d, rtol=2.5e-7, atol=0)


def test_triplets_pairs_angles_match_oracle(dev):
    """Same (k,j,i) triplets / (j,i,j') pairs and the same angles as PAMNet.indices (models.py:68-98)."""
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, synth
    b = synth.qm9_batch(7, 0, 16)
    n = b.x.numel()
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x.to(dev), b.batch.to(dev), b.pos.to(dev),
                      b.edge_index.to(dev), num_graphs=16)
    ei_l, _ = O.get_edge_info(b.edge_index, b.pos)
    (idx_i, idx_j, idx_k, idx_kj, idx_ji, i_p, j1_p, j2_p, jj_p, ji_p) = O.indices(ei_l, n)
    a2 = O.angle_between(b.pos[idx_j] - b.pos[idx_i], b.pos[idx_k] - b.pos[idx_j])
    a1 = O.angle_between(b.pos[j1_p] - b.pos[i_p], b.pos[j2_p] - b.pos[j1_p])
    src, dst = g.loc.col.cpu().long(), g.loc.row_of.cpu().long()
    e, e2 = g.tp.row_of.cpu().long(), g.tp.col.cpu().long()
    kind, ang = g.tp_kind.cpu(), g.tp_angle.cpu()
    assert g.n_trip == idx_kj.numel() and g.n_pair == jj_p.numel()
    mine_t = {(int(src[b_]), int(src[a_]), int(dst[a_])): float(x) for a_, b_, k_, x in zip(e, e2, kind, ang) if k_ == 0}
    ref_t = {(int(k), int(j), int(i)): float(x) for k, j, i, x in zip(idx_k, idx_j, idx_i, a2)}
    assert mine_t.keys() == ref_t.keys()
    assert max(abs(mine_t[k] - ref_t[k]) for k in ref_t) < 2e-6
    mine_p = {(int(src[a_]), int(dst[a_]), int(src[b_])): float(x) for a_, b_, k_, x in zip(e, e2, kind, ang) if k_ == 1}
    ref_p = {(int(i), int(j1), int(j2)): float(x) for i, j1, j2, x in zip(i_p, j1_p, j2_p, a1)}
    assert mine_p.keys() == ref_p.keys()
    assert max(abs(mine_p[k] - ref_p[k]) for k in ref_p) < 2e-3        # self-pairs: atan2(~0, -1): pi up to sqrt noise
    # rows are grouped by target edge, triplets before pairs
    assert torch.equal(e, torch.sort(e).values)


def test_star_indices_golden(dev, golden):
    from pamnet_amd import graph as G
    gd = golden('star_indices')
    ei = torch.from_numpy(gd['edge_index']).to(dev)
    pos = torch.tensor([[0., 0, 0], [1., 0, 0], [1.5, 1, 0], [1.5, -1, 0.3]], device=dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', torch.zeros(4, device=dev),
                      torch.zeros(4, dtype=torch.long, device=dev), pos, ei, num_graphs=1)
    src, dst = g.loc.col.cpu().long(), g.loc.row_of.cpu().long()
    ref_e = list(zip(gd['edge_index'][0].tolist(), gd['edge_index'][1].tolist()))
    mine = sorted((ref_e.index((int(src[b]), int(dst[b]))), ref_e.index((int(src[a]), int(dst[a]))), int(k))
                  for a, b, k in zip(g.tp.row_of.cpu(), g.tp.col.cpu(), g.tp_kind.cpu()))
    ref = sorted([(int(a), int(b), 0) for a, b in zip(gd['idx_kj'], gd['idx_ji'])] +
                 [(int(a), int(b), 1) for a, b in zip(gd['idx_jj_pair'], gd['idx_ji_pair'])])
    assert mine == ref


def test_basis_vs_golden_tables(dev, golden):
    """RBF / SBF kernels vs the reference's own layers evaluated in fp64 (basis_tables.npz)."""
    from pamnet_amd import lib, ops
    gd = golden('basis_tables')
    dist = torch.from_numpy(gd['dist']).float().to(dev)
    ang = torch.from_numpy(gd['angle']).float().to(dev)
    m = dist.numel()
    c = float(gd['cutoff'])
    freq = (torch.arange(1, 17, dtype=torch.float32) * np.pi).to(dev)
    rbf = ops.rbf(dist, freq, c)
    assert maxnorm_err(rbf.cpu(), gd['rbf64']) < 2e-6
    rad = torch.empty(m * 42, device=dev)
    lib.call('pamnet_sbf_radial_f32', lib.ptr(dist), c, m, lib.ptr(rad), lib.stream_of(dist))
    sbf = torch.empty(m, 42, device=dev)
    idx = torch.arange(m, dtype=torch.int32, device=dev)
    lib.call('pamnet_sbf_combine_f32', lib.ptr(rad), lib.ptr(idx), lib.ptr(ang), m, lib.ptr(sbf), lib.stream_of(dist))
    ref32_err = [maxnorm_err(gd['sbf32'][:, 6 * l:6 * l + 6], gd['sbf64'][:, 6 * l:6 * l + 6]) for l in range(7)]
    for l in range(7):
        blk = slice(6 * l, 6 * l + 6)
        err = maxnorm_err(sbf[:, blk].cpu(), gd['sbf64'][:, blk])
        # fp32 inputs (dist, angle rounded to fp32) bound this at ~1e-6; never worse than the reference's own fp32 run
        assert err < max(3e-6, 2 * ref32_err[l]), (l, err, ref32_err[l])


@pytest.mark.parametrize('basis', [(7, 6, 5), (5, 4, 6), (8, 7, 4), (1, 1, 1), (16, 12, 9)])
def test_table_driven_basis_vs_scipy(dev, basis):
    """The table-driven basis kernels (any num_spherical / num_radial / envelope exponent; models.SphericalBasis computes the
    zeros / normalisers on the host without scipy) on a dense grid against env_p(x) N_ln j_l(z_ln x) Y_l0(theta) with scipy's
    spherical_jn in fp64, block by block; for the default sizes also against the specialised compile-time kernels; the
    run-time-exponent Bessel rows against the oracle's."""
    import models
    from pamnet_amd import graph as G, lib, ops
    from oracle import pamnet_oracle as O
    ns, nr, p = basis
    z32, norm = models.basis_tables(ns, nr)
    k = O.basis_constants(ns, nr)
    assert np.array_equal(z32, k['zeros']) and np.max(np.abs(norm / k['norm'] - 1)) < 1e-14      # host tables == scipy's
    cutoff, m = 5.0, 4099
    dist = torch.linspace(0.03 * cutoff, 1.06 * cutoff, m, device=dev)
    ang = torch.linspace(0.0, float(np.pi), m, device=dev)
    idx = torch.randperm(m, device=dev).to(torch.int32)

    class _G(object):
        pass
    g = _G()
    g.pos, g.dist_l, g.tp_angle = dist, dist, ang
    g.loc, g.tp = _G(), _G()
    g.loc.m, g.tp.m, g.tp.col = m, m, idx
    sbf = G.spherical_basis_tab(g, cutoff, ns, nr, p, torch.from_numpy(z32).reshape(-1).to(dev),
                                torch.from_numpy(norm).reshape(-1).to(dev))
    x = (dist * torch.tensor(1.0 / cutoff, dtype=torch.float32, device=dev)).cpu().double()
    zx = x.unsqueeze(-1).numpy() * z32.astype(np.float64).reshape(1, -1)
    jl = np.concatenate([O._sph_jn_f64(l, zx[:, nr * l:nr * l + nr]) for l in range(ns)], axis=1)
    rad = O.envelope(x, p).unsqueeze(-1) * torch.from_numpy(jl * norm.reshape(1, -1))
    cbf = O.sbf_angular(ang.cpu().double(), ns)
    ref = (rad[idx.cpu().long()].view(m, ns, nr) * cbf.view(m, ns, 1)).view(m, ns * nr)
    for l in range(ns):
        blk = slice(nr * l, nr * l + nr)
        assert maxnorm_err(sbf[:, blk].cpu(), ref[:, blk]) < 3e-6, (basis, l)
    if basis == (7, 6, 5):
        g.dist_l = dist
        assert maxnorm_err(sbf.cpu(), G.spherical_basis(g, cutoff).cpu()) < 1e-6
    freq = (torch.arange(1, 17, dtype=torch.float32) * np.pi).to(dev).requires_grad_(True)
    rbf = ops.rbf(dist, freq, cutoff, exponent=p)
    r64 = O.bessel_rbf(dist.cpu().double(), freq.detach().cpu().double(), cutoff, p)
    assert maxnorm_err(rbf.detach().cpu(), r64) < 3e-6
    w = torch.randn(m, 16, device=dev)
    (rbf * w).sum().backward()
    f64 = freq.detach().cpu().double().requires_grad_(True)
    (O.bessel_rbf(dist.cpu().double(), f64, cutoff, p) * w.cpu().double()).sum().backward()
    assert maxnorm_err(freq.grad.cpu(), f64.grad) < 2e-5


@pytest.mark.parametrize('cutoff', [5.0, 2.0, 16.0])
def test_sbf_radial_dense_grid_vs_oracle(dev, cutoff):
    """The 42 radial functions on a dense grid of edge lengths (series branch, recurrence branch and beyond the cutoff)
    against env(x) N_ln j_l(z_ln x) with scipy's spherical_jn in fp64 (the oracle's closed sin / cos forms lose ~1e-6 of
    the block maximum to cancellation at small z even in fp64); block by block, relative to the block's largest value."""
    from pamnet_amd import lib
    from oracle import pamnet_oracle as O
    m = 20011
    dist = torch.linspace(0.02 * cutoff, 1.08 * cutoff, m, device=dev)
    rad = torch.empty(m * 42, device=dev)
    lib.call('pamnet_sbf_radial_f32', lib.ptr(dist), cutoff, m, lib.ptr(rad), lib.stream_of(dist))
    x32 = (dist * torch.tensor(1.0 / cutoff, dtype=torch.float32, device=dev)).cpu()    # the kernel's fp32 d / cutoff
    k = O.basis_constants()
    xd = x32.double()
    zx = xd.unsqueeze(-1).numpy() * k['zeros'].astype(np.float64).reshape(1, 42)
    jl = np.concatenate([O._sph_jn_f64(l, zx[:, 6 * l:6 * l + 6]) for l in range(7)], axis=1)
    ref = O.envelope(xd).unsqueeze(-1) * torch.from_numpy(jl * np.asarray(k['norm'], dtype=np.float64).reshape(1, 42))
    assert float((ref - O.sbf_radial(xd, 1.0)).abs().max() / ref.abs().max()) < 1e-5        # same functions as the oracle's
    got = rad.view(m, 42).cpu().double()
    assert torch.isfinite(got).all()
    assert (got[x32 >= 1.0] == 0).all()
    for l in range(7):
        blk = slice(6 * l, 6 * l + 6)
        err = float((got[:, blk] - ref[:, blk]).abs().max() / ref[:, blk].abs().max())
        assert err < 2e-7, (l, err)


def test_rbf_backward_freq(dev):
    from pamnet_amd import ops
    from oracle import pamnet_oracle as O
    dist = torch.rand(3000, device=dev) * 5.2 + 0.5
    freq = (torch.arange(1, 17, dtype=torch.float32, device=dev) * np.pi).requires_grad_()
    w = torch.randn(3000, 16, device=dev)
    (ops.rbf(dist, freq, 5.0) * w).sum().backward()
    f64 = (torch.arange(1, 17, dtype=torch.float32) * np.pi).double().requires_grad_()
    (O.bessel_rbf(dist.cpu().double(), f64, 5.0) * w.cpu().double()).sum().backward()
    assert maxnorm_err(freq.grad.cpu(), f64.grad) < 1e-5


def test_fuse_pool_fwd_bwd(dev):
    from oracle import pamnet_oracle as O
    from pamnet_amd import graph as G, ops
    L, n, B = 3, 500, 17
    batch = torch.sort(torch.randint(0, B, (n,))).values
    g = G.Graph()
    g.n_graphs, g.node_graph = B, batch.to(torch.int32).to(dev)
    g.gptr, _ = G.csr_from_keys(g.node_graph, B)
    g.sign = torch.where(torch.rand(n) > 0.5, 1.0, -1.0).to(dev)
    outs = torch.randn(2 * L, n, device=dev, requires_grad=True)
    atts = torch.randn(2 * L, n, device=dev, requires_grad=True)
    w = torch.randn(B, device=dev)
    for mean in (False, True):
        out, node_out = ops.fuse_pool(outs, atts, g, mean)
        (out * w).sum().backward()
        o64, a64 = outs.detach().cpu().double().requires_grad_(), atts.detach().cpu().double().requires_grad_()
        node = O.fuse([o64[2 * l].view(1, n, 1) for l in range(L)], [o64[2 * l + 1].view(1, n, 1) for l in range(L)],
                      [a64[2 * l].view(1, n, 1) for l in range(L)], [a64[2 * l + 1].view(1, n, 1) for l in range(L)])
        node = node * g.sign.cpu().double().unsqueeze(-1)
        ref = O.segment_add(node, batch, B).view(-1)
        if mean:
            ref = ref / torch.bincount(batch, minlength=B).clamp(min=1)
        (ref * w.cpu().double()).sum().backward()
        assert maxnorm_err(out.detach().cpu(), ref.detach()) < 5e-6
        assert maxnorm_err(outs.grad.cpu(), o64.grad) < 5e-6 and maxnorm_err(atts.grad.cpu(), a64.grad) < 5e-6
        outs.grad = atts.grad = None


def test_cabi_argument_errors(dev):
    """Every entry point validates its arguments before launching: negative sizes / unsupported widths -> PAMNET_EINVAL,
    missing required pointers -> PAMNET_ENULL (surfaced as RuntimeError by the binding); zero rows are a no-op."""
    from pamnet_amd import lib
    x = torch.randn(8, 128, device=dev)
    ptr = torch.tensor([0, 3, 8], dtype=torch.int32, device=dev)
    out = torch.empty(2, 128, device=dev)
    st = lib.stream_of(x)
    with pytest.raises(RuntimeError, match='EINVAL'):
        lib.call('pamnet_segment_sum_f32', lib.ptr(out), None, lib.ptr(x), None, None, None, None, lib.ptr(ptr), 2, 126, st)
    with pytest.raises(RuntimeError, match='ENULL'):
        lib.call('pamnet_segment_sum_f32', None, None, lib.ptr(x), None, None, None, None, lib.ptr(ptr), 2, 128, st)
    with pytest.raises(RuntimeError, match='EINVAL'):
        lib.call('pamnet_embed_fwd_f32', lib.ptr(x), 8, 17, None, lib.ptr(x), None, None, None, 1, lib.ptr(out), st)
    with pytest.raises(RuntimeError, match='ENULL'):
        lib.call('pamnet_mlp2_fwd_f32', lib.ptr(x), 8, None, None, None, None, None, None, lib.ptr(out), st)
    with pytest.raises(RuntimeError, match='EINVAL'):
        lib.call('pamnet_adam_ema_f32', lib.ptr(x), lib.ptr(x), lib.ptr(x), lib.ptr(x), lib.ptr(x), 1023, 1e-3, 0.9, 0.999,
                 1e-8, 0.0, 1, 0.999, None, 1000.0, 0, st)
    # zero rows: nothing is launched, nothing is touched
    out.fill_(7.0)
    lib.call('pamnet_segment_sum_f32', lib.ptr(out), None, lib.ptr(x), None, None, None, None, lib.ptr(ptr), 0, 128, st)
    lib.call('pamnet_mlp2_fwd_f32', lib.ptr(x), 0, None, None, None, None, None, None, lib.ptr(out), st)
    assert float(out.min()) == 7.0
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        lib.stream_of(torch.zeros(3))


# ------------------------------------------------------------------------------------------- one-launch reductions
@pytest.mark.parametrize('n', [4, 1000, 65536 + 12, 3581100 // 4 * 4])
def test_grad_norm(dev, n):
    """Partial sums of squares + the optimiser kernel's own finish: the norm it reports and clips by."""
    from pamnet_amd import lib, ops
    torch.manual_seed(n)
    g = torch.randn(n, device=dev) * 3.0
    part = ops.sumsq_partials(g)
    ref = g.double().norm()
    assert abs(float(part.sum().sqrt()) - float(ref)) <= 2e-7 * float(ref)
    assert torch.equal(part, ops.sumsq_partials(g))                  # bitwise repeatable
    p, m, v, sh = (torch.zeros(n, device=dev) for _ in range(4))
    norm = torch.zeros(1, device=dev)
    g2 = g.clone()
    lib.call('pamnet_adam_ema_norm_f32', lib.ptr(p), lib.ptr(g2), lib.ptr(m), lib.ptr(v), lib.ptr(sh), n, 1e-3, 0.9, 0.999,
             1e-8, 0.0, 1, 0.999, lib.ptr(part), lib.ptr(norm), 1.0, 0, lib.stream_of(g))
    assert abs(float(norm) - float(ref)) <= 2e-7 * float(ref)
    # same update as the device-scalar form of the kernel given the same norm
    p1, m1, v1, sh1 = (torch.zeros(n, device=dev) for _ in range(4))
    lib.call('pamnet_adam_ema_f32', lib.ptr(p1), lib.ptr(g.clone()), lib.ptr(m1), lib.ptr(v1), lib.ptr(sh1), n, 1e-3, 0.9,
             0.999, 1e-8, 0.0, 1, 0.999, lib.ptr(norm), 1.0, 0, lib.stream_of(g))
    assert torch.equal(p, p1) and torch.equal(m, m1) and torch.equal(v, v1) and torch.equal(sh, sh1)


@pytest.mark.parametrize('n', [1, 7, 128, 1024, 5000])
def test_l1_loss_with_grad(dev, n):
    from pamnet_amd import ops
    torch.manual_seed(n)
    out = torch.randn(n, device=dev)
    y = torch.randn(n, device=dev)
    if n > 4:
        y[3] = out[3]                                                # sign(0) = 0, as torch
    o = out.clone().requires_grad_()
    ref = torch.nn.functional.l1_loss(o, y)
    (ref * 0.25).backward()
    loss, d_out = ops.l1_loss_with_grad(out, y, 0.25)
    assert abs(float(loss) - float(ref)) <= 2e-7 * abs(float(ref)) + 1e-12
    assert torch.allclose(d_out, o.grad, rtol=1e-6, atol=0)


@pytest.mark.parametrize('d', [16, 64, 128])
@pytest.mark.parametrize('n,types', [(1, 3), (37, 5), (2286, 5), (17700, 3)])
def test_type_rows_gather_and_grad(dev, n, types, d):
    """embeddings[x] (models.py:107,140) and its gradient (rows of d x summed per type)."""
    from pamnet_amd import ops
    torch.manual_seed(n + d)
    table = torch.randn(types, d, device=dev, requires_grad=True)
    idx = torch.randint(0, types, (n,), device=dev, dtype=torch.int32)
    if n > 10:
        idx[idx == types - 1] = 0                                    # a type that never occurs -> zero gradient row
    out = ops.type_rows(table, idx)
    assert torch.equal(out, table.detach()[idx.long()])
    w = torch.randn(n, d, device=dev)
    (out * w).sum().backward()
    ref = torch.zeros(types, d, dtype=torch.float64, device=dev).index_add_(0, idx.long(), w.double())
    assert maxnorm_err(table.grad.cpu(), ref.cpu()) < 2e-6
    g1 = table.grad.clone()
    table.grad = None
    (ops.type_rows(table, idx) * w).sum().backward()
    assert torch.equal(g1, table.grad)                               # deterministic
    buf = torch.full((types, d), float('nan'), device=dev)           # direct-gradient mode: written in place
    t2 = table.detach().clone().requires_grad_()
    (ops.type_rows(t2, idx, buf) * w).sum().backward()
    assert t2.grad is None and torch.equal(buf, g1)


@pytest.mark.parametrize('rows', [[1, 15, 64, 65], [2286] * 4 + [32888], [700, 3, 4316, 17640]])
def test_wgrad_batched_vs_fp64(dev, rows):
    """pamnet_wgrad_batched_f32: dW = dZ^T A (A optionally SiLU'd while staging), db = column sums of dZ.  The GEMMs run
    on the bf16 matrix pipe with both operands split exactly into three bf16 pieces (csrc/gemm_core.h, "bf16x6"): held
    to fp32-GEMM accuracy -- the error against fp64 may not exceed twice that of torch's fp32 matmul on the same inputs
    (floor 2e-7) -- on operands whose magnitudes spread over 2^+-20 row by row, ragged / empty row counts, strided dW
    targets; bitwise repeatable."""
    from pamnet_amd import fused
    g = torch.Generator(device='cpu').manual_seed(sum(rows) + 7)
    jobs, refs = [], []
    for j, r in enumerate(rows):
        scale = torch.exp2(torch.randint(-20, 21, (max(r, 1), 1), generator=g).float())[:r]
        dZ = (torch.randn(r, 128, generator=g) * scale).to(dev)
        A = torch.randn(r, 128, generator=g).to(dev) * (3.0 if j % 2 else 1.0)
        mode = j % 2
        wide = torch.full((128, 384), float('nan'), device=dev)         # dW lands in a column block of a [128, 384] weight
        db = torch.full((128,), float('nan'), device=dev)
        jobs.append((dZ, 128, A, 128, mode, r, wide.data_ptr() + 4 * 128, 384, db))
        a64 = A.double()
        a64 = a64 * torch.sigmoid(a64) if mode else a64
        a32 = torch.nn.functional.silu(A) if mode else A
        refs.append((wide, db, dZ.double().t() @ a64, dZ.double().sum(0), dZ.t() @ a32))
    fused.wgrad(jobs, jobs[0][0])
    first = [(w.clone(), b.clone()) for w, b, *_ in refs]
    fused.wgrad(jobs, jobs[0][0])
    worst = 0.0
    for (wide, db, w64, b64, w32), (w1, b1) in zip(refs, first):
        assert torch.equal(wide[:, 128:256], w1[:, 128:256]) and torch.equal(db, b1)      # run-to-run bitwise
        assert torch.isnan(wide[:, :128]).all() and torch.isnan(wide[:, 256:]).all()      # neighbours untouched
        got = wide[:, 128:256]
        if w64.abs().max() == 0:
            assert (got == 0).all() and (db == 0).all()
            continue
        e, floor = maxnorm_err(got.cpu(), w64.cpu()), maxnorm_err(w32.cpu(), w64.cpu())
        worst = max(worst, e / max(floor, 1e-7))
        assert e <= max(2e-7, 2 * floor), (e, floor)
        assert maxnorm_err(db.cpu(), b64.cpu()) < 2e-6
    print('wgrad bf16x6: worst error / fp32-matmul error = %.2f' % worst)


@pytest.mark.parametrize('kind', ['QM9', 'PDBbind'])
def test_symmetric_transpose_equals_counting_sort(dev, kind):
    """The transposed CSR of a radius graph taken as its reverse-edge index (one bisection per edge) is bit for bit what
    the stable counting sort over the columns returns."""
    from pamnet_amd import graph as G, synth
    if kind == 'QM9':
        b = synth.qm9_batch(11, 0, 64).to(dev)
        g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=64)
        pairs = [(g.glob, g.glob_T)]
    else:
        b = synth.pdbbind_batch(3, 0, 4).to(dev)
        g = G.build_graph('PDBbind', 2.0, 6.0, 'source_to_target', b.x, b.batch, num_graphs=4)
        pairs = [(g.glob, g.glob_T), (g.loc, g.loc_T)]
    for csr, tr in pairs:
        assert isinstance(tr, G.SymmetricTranspose)
        ref = G.Transpose(csr.col, g.n)
        assert torch.equal(tr.ptr, ref.ptr) and torch.equal(tr.perm, ref.perm)


@pytest.mark.parametrize('kinds', [('i64', 'i64', 'i64'), ('i32', 'f32', 'i32'), ('i64', 'f32', 'i64')])
def test_ingest_indices_matches_tensor_ops(dev, kinds):
    """pamnet_ingest_indices_i32 (casts + node pointer + validation + self-loop note in one launch) against the tensor
    ops it replaces; a batch vector with empty graphs at the start, in the middle and at the end."""
    from pamnet_amd import graph as G
    dt = {'i64': torch.int64, 'i32': torch.int32, 'f32': torch.float32}
    rng = np.random.RandomState(5)
    counts = np.array([0, 0, 5, 9, 0, 1, 17, 0, 0, 3, 0])
    n_graphs, n = len(counts), int(counts.sum())
    batch = torch.from_numpy(np.repeat(np.arange(n_graphs), counts)).to(dev).to(dt[kinds[0]])
    x = torch.from_numpy(rng.randint(0, 5, size=n)).to(dev).to(dt[kinds[1]])
    ne = 83
    ei = torch.from_numpy(rng.randint(0, n, size=(2, ne))).to(dev)
    ei[1] = torch.where(ei[1] == ei[0], (ei[1] + 1) % n, ei[1])              # no self loops
    ei = ei.to(dt[kinds[2]])
    for shape in ('flat', 'column'):
        xs = x if shape == 'flat' else x.view(-1, 1)
        node_graph, gptr, types, src, dst, flag, loops, spare = G.ingest(batch, n_graphs, xs, 5, ei)
        assert int(spare.abs().sum()) == 0                  # four zeroed words for the caller
        ref_ptr, _ = G.csr_from_keys(batch.to(torch.int32).contiguous(), n_graphs)
        assert torch.equal(node_graph, batch.to(torch.int32)) and torch.equal(gptr, ref_ptr)
        assert torch.equal(types, x.to(torch.int32))
        assert torch.equal(src, ei[0].to(torch.int32)) and torch.equal(dst, ei[1].to(torch.int32))
        assert int(flag) == 0 and int(loops) == 0
    # a self loop is noted, nothing else changes
    e2 = ei.clone()
    e2[1, 40] = e2[0, 40]
    out = G.ingest(batch, n_graphs, x, 5, e2)
    assert int(out[5]) == 0 and int(out[6]) == 1
    # every kind of invalid index raises the flag and is written as an in-range value
    def bad(b=batch, xx=x, e=ei):
        o = G.ingest(b, n_graphs, xx, 5, e)
        assert int(o[5]) == 1
        assert int(o[0].min()) >= 0 and int(o[0].max()) < n_graphs and int(o[2].min()) >= 0 and int(o[2].max()) < 5
        assert int(o[1].min()) >= 0 and int(o[1].max()) <= n
        assert int(o[3].min()) >= 0 and int(o[3].max()) < n and int(o[4].min()) >= 0 and int(o[4].max()) < n
    b2 = batch.clone(); b2[7] = n_graphs
    bad(b=b2)
    b2 = batch.clone(); b2[3], b2[20] = b2[20].clone(), b2[3].clone()           # not sorted
    bad(b=b2)
    x2 = x.clone(); x2[11] = 5
    bad(xx=x2)
    x2 = x.clone(); x2[0] = -1
    bad(xx=x2)
    e2 = ei.clone(); e2[0, 3] = n
    bad(e=e2)
    e2 = ei.clone(); e2[1, 80] = -2
    bad(e=e2)


def test_qm9_graph_same_with_and_without_ingest(dev):
    """build_graph on the reference's int64 tensors (ingest launch) equals build_graph on pre-converted tensors of a
    layout the launch does not take (the tensor-op route), field by field; a bond list with self loops takes the
    remove_self_loops route on both."""
    from pamnet_amd import graph as G, synth
    b = synth.qm9_batch(4, 0, 32).to(dev)
    assert b.edge_index.dtype == torch.int64 and b.batch.dtype == torch.int64
    for loops in (False, True):
        ei = b.edge_index
        if loops:
            extra = torch.tensor([[3, 17], [3, 17]], device=dev, dtype=ei.dtype)
            ei = torch.cat([ei[:, :10], extra, ei[:, 10:]], dim=1).contiguous()
        a = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, ei, num_graphs=32, n_types=5)
        assert a.types is not None
        # int16 indices are not a kind the launch reads: same values, tensor-op route
        c = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch.to(torch.int16), b.pos, ei, num_graphs=32,
                          n_types=5)
        assert c.types is None
        for name in ('node_graph', 'gptr', 'dist_g', 'dist_l', 'tp_angle', 'tp_kind'):
            assert torch.equal(getattr(a, name), getattr(c, name)), name
        for name in ('glob', 'loc', 'tp'):
            for f in ('ptr', 'row_of', 'col'):
                assert torch.equal(getattr(getattr(a, name), f), getattr(getattr(c, name), f)), (name, f)
        assert torch.equal(a.types, b.x.reshape(-1).to(torch.int32))
    bad = b.batch.clone(); bad[5] = 40
    with pytest.raises(IndexError):
        G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, bad, b.pos, b.edge_index, num_graphs=32, n_types=5)


@pytest.mark.parametrize('flow', ['source_to_target', 'target_to_source'])
def test_rna_inverse_transpose_rows_match_counting_sort(dev, flow):
    """The backward index of the RNA kNN graphs (query-ordered pointer + inverse of the transposition) holds, row by row,
    the same entries as the stable counting sort over the stored columns (their order inside a row is the kNN order)."""
    from pamnet_amd import graph as G, synth
    b = synth.rna_batch(2, 0, 2).to(dev)
    g = G.build_graph('rna_native', 16.0, 20.0, flow, b.x, b.batch, num_graphs=2, n_types=4)
    pairs = [(g.loc, g.loc_T)] + ([(g.glob, g.glob_T)] if flow == 'source_to_target' else [])
    for csr, tr in pairs:
        assert isinstance(tr, G.InverseTranspose)
        ref = G.Transpose(csr.col, g.n)
        assert torch.equal(tr.ptr, ref.ptr)
        assert torch.equal(csr.col[tr.perm.long()], csr.col[ref.perm.long()])          # every entry sits in its row
        assert torch.equal(torch.sort(tr.perm).values, torch.arange(csr.m, device=dev, dtype=torch.int32))   # a permutation
        key = csr.col[tr.perm.long()].long() * csr.m
        assert torch.equal(torch.sort(key + tr.perm.long()).values, key + ref.perm.long())   # same set per row
    if flow == 'target_to_source':
        assert isinstance(g.glob_T, G.Transpose)


# ---- round-3 graph-construction entry points, each against the entry points it stands in for -----------------------------
@pytest.mark.parametrize('n_nodes,k', [(300, 50), (40, 50), (700, 17)])
def test_knn_cut_and_fill_equal_table_filters_and_expands(dev, n_nodes, k):
    """pamnet_knn_cut_i32 + pamnet_exclusive_scan_pair_i32 + pamnet_knn_cut_fill_i32 == pamnet_knn_i32 (no cutoff) followed by
    two pamnet_csr_filter_count/fill_i32, two expands and the pointer clamps -- same table, counts, lists, query ids."""
    from pamnet_amd import graph as G, lib, synth
    b = synth.rna_batch(4, 0, 3, n_nodes=n_nodes)
    pos = b.x[:, :3].contiguous().to(dev)
    nodeg = b.batch.to(torch.int32).to(dev)
    n = int(nodeg.numel())
    gptr, _ = G.csr_from_keys(nodeg, 3)
    cut_a, cut_b = 20.0, 2.6
    kp, kn, kd = G.knn_table(pos, nodeg, gptr, k, float('inf'))
    want = [G.csr_filter(kp, kn, kd, c) for c in (cut_a, cut_b)]
    st = lib.stream_of(pos)
    P = lib.ptr
    i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
    kn2, kd2 = i32(n * k), torch.empty(n * k, device=dev)
    ca, cb, ra, rb, tmp = i32(n), i32(n), i32(n + 1), i32(n + 1), i32(n // 4096 + 2)
    lib.call('pamnet_knn_cut_i32', P(pos), P(nodeg), P(gptr), n, k, cut_a, cut_b, P(kn2), P(kd2), P(ca), P(cb), st)
    assert torch.equal(kn2, kn) and torch.equal(kd2, kd)
    lib.call('pamnet_exclusive_scan_pair_i32', P(ca), P(ra), P(cb), P(rb), n, P(tmp), st)
    assert torch.equal(ra, want[0][0]) and torch.equal(rb, want[1][0])
    ea, eb = int(ra[-1]), int(rb[-1])
    for cap_a, cap_b in ((ea, eb), (max(ea - 5, 1), eb + 7)):       # exact sizes; one too small, one too large
        outs = [(i32(cap_a).zero_(), torch.zeros(cap_a, device=dev), i32(cap_a).zero_(), i32(n + 1)),
                (i32(cap_b).zero_(), torch.zeros(cap_b, device=dev), i32(cap_b).zero_(), i32(n + 1))]
        lib.call('pamnet_knn_cut_fill_i32', P(kn), P(kd), n, k, cut_a, P(ra), cap_a, P(outs[0][0]), P(outs[0][1]), P(outs[0][2]),
                 P(outs[0][3]), cut_b, P(rb), cap_b, P(outs[1][0]), P(outs[1][1]), P(outs[1][2]), P(outs[1][3]), st)
        for (nbr, dist, row, ptr), (wp, wn, wd), cap in zip(outs, want, (cap_a, cap_b)):
            m = min(cap, int(wp[-1]))
            assert torch.equal(ptr, torch.clamp(wp, max=cap))
            assert torch.equal(nbr[:m], wn[:m]) and torch.equal(dist[:m], wd[:m])
            assert torch.equal(row[:m], G.expand_rows(wp, int(wp[-1]))[:m])
            assert int(nbr[m:].abs().sum()) == 0                   # nothing written beyond what the list holds


@pytest.mark.parametrize('with_triplets', [True, False])
@pytest.mark.parametrize('n_nodes,k,cut', [(300, 50, 2.6), (40, 50, 2.6), (700, 17, 6.0), (900, 50, 20.0)])
def test_knn_triplet_total_before_the_graph_exists(dev, n_nodes, k, cut, with_triplets):
    """Round 6: pamnet_knn_tp_total_i64 -- the triplet + pair row total of the local graph (models.py:68-98) from the kNN table
    alone -- equals the scanned row counts of pamnet_triplet_count_i32 on the finished, transposed local graph (what the second
    host read-back of the RNA path used to fetch), for sparse and dense cuts, k below and at the reference's 50."""
    from pamnet_amd import graph as G, lib, synth
    b = synth.rna_batch(6, 0, 3, n_nodes=n_nodes)
    pos = b.x[:, :3].contiguous().to(dev)
    nodeg = b.batch.to(torch.int32).to(dev)
    n = int(nodeg.numel())
    gptr, _ = G.csr_from_keys(nodeg, 3)
    res = G.knn_cuts(pos, nodeg, gptr, k, 20.0, cut, None, tp_of_b=with_triplets)
    (qp, qn, qd, qq), total = res[1], res[2]
    lp, l_src, l_dist, _ = G._transpose_edges(qp, qn, qd, n, q=qq)
    l_dst = G.expand_rows(lp, int(l_src.numel()))
    tp_ptr, _ = G._triplet_ptr(lp, l_src, l_dst, with_triplets)
    assert total == int(tp_ptr[-1]) and (total > 0 or int(l_src.numel()) == 0)


@pytest.mark.parametrize('n', [1, 100, 24576, 24577, 70000])
def test_exclusive_scan_pair(dev, n):
    from pamnet_amd import graph as G, lib
    a = torch.randint(0, 60, (n,), device=dev, dtype=torch.int32)
    b = torch.randint(0, 9, (n,), device=dev, dtype=torch.int32)
    oa, ob = torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int32, device=dev)
    tmp = torch.empty(n // 4096 + 2, dtype=torch.int32, device=dev)
    lib.call('pamnet_exclusive_scan_pair_i32', lib.ptr(a), lib.ptr(oa), lib.ptr(b), lib.ptr(ob), n, lib.ptr(tmp), lib.stream_of(a))
    assert torch.equal(oa, G.exclusive_scan(a)) and torch.equal(ob, G.exclusive_scan(b))


@pytest.mark.parametrize('with_triplets', [True, False])
@pytest.mark.parametrize('case', ['qm9', 'dense', 'asymmetric'])
def test_triplet_transpose_equals_counting_sort(dev, case, with_triplets):
    """pamnet_triplet_transpose_count/fill_i32 against pamnet_csr_from_keys_i32 of the row list's source-bond column: usual
    degrees (both bond lists of an atom in registers), degrees over the register path's limit (a clique), a bond list with
    one direction missing (in-degree != out-degree), transposed bond lists in any order inside a row."""
    from pamnet_amd import graph as G, lib, synth
    if case == 'qm9':
        b = synth.qm9_batch(3, 0, 20)
        ei, n = b.edge_index, int(b.x.numel())
        pos = b.pos
    else:
        n = 24 if case == 'dense' else 40
        g = torch.Generator().manual_seed(1)
        pos = torch.rand(n, 3, generator=g) * 3
        src, dst = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
        keep = src != dst
        if case == 'asymmetric':
            keep &= torch.rand(n, n, generator=g) < 0.15
        ei = torch.stack([src[keep], dst[keep]])
    ei = ei.to(dev)
    lp, perm = G.csr_from_keys(ei[1].to(torch.int32).contiguous(), n)
    l_src, l_dst = ei[0].to(torch.int32)[perm.long()].contiguous(), ei[1].to(torch.int32)[perm.long()].contiguous()
    loc = G.CSR(lp, l_dst, l_src)
    tp_ptr, tcount = G._triplet_ptr(lp, l_src, l_dst, with_triplets)
    tot, e_l = int(tp_ptr[-1]), int(l_src.numel())
    i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
    tp_idx, tp_edge, tp_kind, tp_angle = i32(tot), i32(tot), i32(tot), torch.empty(tot, device=dev)
    P = lib.ptr
    lib.call('pamnet_triplet_fill_f32', P(pos.to(dev).contiguous()), P(lp), P(l_src), P(l_dst), e_l, 1 if with_triplets else 0,
             P(tp_ptr), P(tp_idx), P(tp_edge), P(tp_angle), P(tp_kind), tot, lib.stream_of(lp))
    want = G.Transpose(tp_idx, e_l)
    locT = G.Transpose(l_src, n)
    for shuffle in (False, True):
        if shuffle:                                  # the kNN path hands rows over in its own order
            rows = torch.repeat_interleave(torch.arange(n, device=dev), (locT.ptr[1:] - locT.ptr[:-1]).long())
            key = rows.double() + torch.rand(e_l, device=dev, dtype=torch.float64) * 0.5
            locT.perm = locT.perm[torch.argsort(key)].contiguous()
        got = G.TripletTranspose(loc, locT, tp_ptr, tcount, tot, with_triplets)
        assert torch.equal(got.ptr, want.ptr) and torch.equal(got.perm, want.perm), (case, shuffle)


def test_sbf_radial_both_forms_give_the_same_floats(dev):
    """pamnet_sbf_radial_f32 runs a thread per value below 40 k edges and a thread per (edge, n) above: same arithmetic per
    value, so a long call equals its short pieces bit for bit."""
    from pamnet_amd import lib
    m = 50000
    dist = (torch.rand(m, device=dev) * 4.9 + 0.05).contiguous()
    whole = torch.empty(m, 42, device=dev)
    lib.call('pamnet_sbf_radial_f32', lib.ptr(dist), 5.0, m, lib.ptr(whole), lib.stream_of(dist))
    parts = torch.empty(m, 42, device=dev)
    for a in range(0, m, 12500):
        d = dist[a:a + 12500].contiguous()
        out = torch.empty(12500, 42, device=dev)
        lib.call('pamnet_sbf_radial_f32', lib.ptr(d), 5.0, 12500, lib.ptr(out), lib.stream_of(d))
        parts[a:a + 12500] = out
    assert torch.equal(whole, parts)

"""Edge MLP -> node segment-sum as one kernel (csrc/edge_agg.hip) vs an fp64 torch evaluation of
layers/global_message_passing.py:38,52-56 (message + add-aggregation) and local_message_passing.py:49-54 on the same
GPU, and vs the unfused kernels (pamnet_global_edge_fwd_f32 + pamnet_segment_sum_f32).  Degree patterns the reference's
graphs produce plus the adversarial ones for a node-aligned work split: nodes without edges (leading, trailing, runs in
the middle), one node with more rows than a chunk holds, no edges at all, every launch geometry (3/5/9/8-tile chunks)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import maxnorm_err

pytestmark = pytest.mark.gpu
D = 128


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _csr(deg, n_src, seed, dev):
    """Random CSR-by-target graph with the given in-degrees; sources uniform in [0, n_src)."""
    rng = np.random.default_rng(seed)
    deg = np.asarray(deg, dtype=np.int64)
    ptr = np.concatenate([[0], np.cumsum(deg)])
    m = int(ptr[-1])
    row_of = np.repeat(np.arange(len(deg)), deg)
    col = rng.integers(0, n_src, size=m)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.int32).to(dev)
    return t(ptr), t(row_of), t(col), m


DEGREE_CASES = {
    'qm9_like': lambda rng: rng.integers(8, 22, size=2286),
    'tiny': lambda rng: rng.integers(0, 5, size=40),
    'mid': lambda rng: rng.integers(5, 30, size=700),          # 5-tile geometry
    'small3': lambda rng: rng.integers(3, 12, size=900),        # 3-tile geometry
    'pdbbind_like': lambda rng: rng.integers(20, 60, size=3000),   # several chunks per workgroup (8-tile geometry)
    'holes': lambda rng: np.concatenate([np.zeros(7, int), rng.integers(0, 3, size=500) * rng.integers(0, 20, size=500),
                                         np.zeros(40, int), rng.integers(10, 20, size=300), np.zeros(5, int)]),
    'giant': lambda rng: np.concatenate([rng.integers(5, 20, size=100), [700], rng.integers(5, 20, size=100), [150, 0, 0, 145],
                                         rng.integers(5, 20, size=50)]),
    'one_giant': lambda rng: np.array([1000]),
    'no_edges': lambda rng: np.zeros(33, int),
    'single_edge': lambda rng: np.array([0, 0, 1, 0]),
}


def _weights(dev, seed):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda *s: (torch.randn(*s, generator=gen) / 8.0).to(dev)
    return mk(D, 3 * D), mk(D), mk(D, D)          # mlp_m.weight [128,384], bias, W_edge_attr


def _ref_fwd(e, Wm, bm, Wea, Pi, Pj, row_of, col, init, n):
    dd = lambda t: t.double()
    z = dd(e) @ dd(Wm[:, 2 * D:]).t() + dd(bm) + dd(Pi)[row_of.long()] + dd(Pj)[col.long()]
    ea = dd(e) @ dd(Wea).t()
    msg = torch.nn.functional.silu(z) * ea
    out = dd(init).clone()
    out.index_add_(0, row_of.long(), msg)
    return z, ea, msg, out


@pytest.mark.parametrize('case', sorted(DEGREE_CASES))
def test_global_edge_agg_fwd_bwd(dev, case):
    from pamnet_amd import lib
    from pamnet_amd.ops import segment_sum_raw
    rng = np.random.default_rng(3)
    deg = DEGREE_CASES[case](rng)
    n = len(deg)
    ptr, row_of, col, m = _csr(deg, n, 5, dev)
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)
    e, Pi, Pj, init = mk(max(m, 1), D)[:m], mk(n, D), mk(n, D), mk(n, D)
    Wm, bm, Wea = _weights(dev, 2)
    st = lib.stream_of(Pi)
    sub = lambda w, c0: w.data_ptr() + 4 * c0
    z, ea = (torch.full((max(m, 1), D), float('nan'), device=dev)[:m] for _ in range(2))
    out = torch.full((n, D), float('nan'), device=dev)
    # the node-aligned work split, precomputed once per graph (first call) / derived by every workgroup itself (second call)
    cuts_t = torch.full((257,), -1, dtype=torch.int32, device=dev)
    lib.call('pamnet_seg_cuts_i32', lib.ptr(ptr), lib.ptr(row_of), n, m, lib.ptr(cuts_t), None, st)
    cuts = lib.ptr(cuts_t)
    lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), m, n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
             lib.ptr(Pi), lib.ptr(Pj), lib.ptr(ptr), lib.ptr(row_of), lib.ptr(col), cuts, lib.ptr(init), lib.ptr(z),
             lib.ptr(ea), lib.ptr(out), st)
    z64, ea64, msg64, out64 = _ref_fwd(e, Wm, bm, Wea, Pi, Pj, row_of, col, init, n)
    assert torch.isfinite(out).all()
    assert maxnorm_err(out.cpu(), out64.cpu()) < 2e-6
    if m:
        assert maxnorm_err(z.cpu(), z64.cpu()) < 2e-6 and maxnorm_err(ea.cpu(), ea64.cpu()) < 2e-6
    # inference mode: no saves, same result bit for bit; run-to-run bitwise identical
    out2 = torch.empty_like(out)
    cuts = None
    lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), m, n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
             lib.ptr(Pi), lib.ptr(Pj), lib.ptr(ptr), lib.ptr(row_of), lib.ptr(col), cuts, lib.ptr(init), None, None,
             lib.ptr(out2), st)
    assert torch.equal(out, out2)
    # against the unfused pair (edge kernel + scatter-add kernel): same maths, different summation order
    if m:
        z3, ea3, msg3 = (torch.empty(m, D, device=dev) for _ in range(3))
        lib.call('pamnet_global_edge_fwd_f32', lib.ptr(e), m, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
                 lib.ptr(Pi), lib.ptr(Pj), lib.ptr(row_of), lib.ptr(col), lib.ptr(z3), lib.ptr(ea3), lib.ptr(msg3), st)
        out3 = torch.empty_like(out)
        segment_sum_raw(out3, init, msg3, None, None, None, None, ptr, n, D)
        # (the fused kernel runs its GEMMs as bf16x6 on the bf16 matrix pipe, the unfused one as fp32 MFMAs: same values to
        # fp32 rounding, not the same bits)
        assert maxnorm_err(z.cpu(), z3.cpu()) < 1e-6 and maxnorm_err(ea.cpu(), ea3.cpu()) < 1e-6
        assert maxnorm_err(out.cpu(), out3.cpu()) < 2e-6

    # ---- backward (first call with the precomputed work split, the accumulate call without)
    d_agg = mk(n, D)
    bcuts = lib.ptr(cuts_t)
    dz, dea, d_e = (torch.full((max(m, 1), D), float('nan'), device=dev)[:m] for _ in range(3))
    dPi = torch.full((n, D), float('nan'), device=dev)
    lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), bcuts, lib.ptr(z),
             lib.ptr(ea), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), 0,
             lib.ptr(dPi), st)
    dm = d_agg.double()[row_of.long()]
    sg = torch.sigmoid(z64)
    dz64 = dm * ea64 * (sg * (1 + z64 * (1 - sg)))
    dea64 = dm * torch.nn.functional.silu(z64)
    de64 = dz64 @ Wm[:, 2 * D:].double() + dea64 @ Wea.double()
    dPi64 = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, row_of.long(), dz64)
    assert torch.isfinite(dPi).all()
    assert maxnorm_err(dPi.cpu(), dPi64.cpu()) < 3e-6 or float(dPi64.abs().max()) == 0.0
    if m:
        for a, b in ((dz, dz64), (dea, dea64), (d_e, de64)):
            assert maxnorm_err(a.cpu(), b.cpu()) < 3e-6
        # accumulate flag
        d_e2 = d_e.clone()
        bcuts = None
        lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), bcuts, lib.ptr(z),
                 lib.ptr(ea), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e2), 1,
                 lib.ptr(dPi), st)
        assert maxnorm_err(d_e2.cpu(), (2 * de64).cpu()) < 3e-6


@pytest.mark.parametrize('case', sorted(DEGREE_CASES) + ['pdbbind_big'])
@pytest.mark.parametrize('accumulate', [0, 1])
def test_global_edge_agg_bwd_with_weight_gradients(dev, case, accumulate):
    """Round 5: dW_e = dz^T e, dW_ea = dea^T e and db_m = colsum(dz) formed INSIDE the fused backward edge kernel
    (layers/global_message_passing.py:52-56: the gradients of mlp_m's e-block and of W_edge_attr), partial tiles per workgroup,
    summed in fixed order by the deferred weight-gradient reduction.  dz / d_e / dPi must be bitwise the plain backward
    kernel's; the weight gradients are checked against fp64 at the tolerance of the bf16x6 split-K kernels."""
    from pamnet_amd import lib
    rng = np.random.default_rng(3)
    deg = rng.integers(25, 50, size=20000) if case == 'pdbbind_big' else DEGREE_CASES[case](rng)
    n = len(deg)
    ptr, row_of, col, m = _csr(deg, n, 5, dev)
    gen = torch.Generator().manual_seed(13)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)
    e, z, ea, d_agg = mk(max(m, 1), D)[:m], mk(max(m, 1), D)[:m], mk(max(m, 1), D)[:m], mk(n, D)
    Wm, bm, Wea = _weights(dev, 2)
    st = lib.stream_of(d_agg)
    sub = lambda w, c0: w.data_ptr() + 4 * c0
    nanlike = lambda r: torch.full((max(r, 1), D), float('nan'), device=dev)[:r]
    # the plain kernel
    dz0, dea0, de0, dPi0 = nanlike(m), nanlike(m), nanlike(m), nanlike(n)
    base = mk(max(m, 1), D)[:m]
    if accumulate:
        de0.copy_(base)
    lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), None, lib.ptr(z),
             lib.ptr(ea), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz0), lib.ptr(dea0), lib.ptr(de0), accumulate,
             lib.ptr(dPi0), st)
    # the fused one
    need, slots = ctypes.c_int64(0), ctypes.c_int64(0)
    lib.call('pamnet_global_edge_agg_wg_floats', m, ctypes.addressof(need), ctypes.addressof(slots))
    partial = torch.full((int(need.value),), float('nan'), device=dev)
    dz1, de1, dPi1 = nanlike(m), nanlike(m), nanlike(n)
    if accumulate:
        de1.copy_(base)
    dWm = torch.full((D, 3 * D), float('nan'), device=dev)       # the [d, 3d] gradient of mlp_m: only its e-block is written
    dWea, db = torch.full((D, D), float('nan'), device=dev), torch.full((D,), float('nan'), device=dev)
    nbytes = ctypes.c_int64(0)
    lib.call('pamnet_wgrad_ctx_bytes', ctypes.addressof(nbytes))
    ctx = (ctypes.c_char * int(nbytes.value))()
    scratch = base.clone() if accumulate else nanlike(m)
    for rep in range(2):                                         # twice: run-to-run bitwise identical
        lib.call('pamnet_global_edge_agg_bwd_wg_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), None, lib.ptr(z),
                 lib.ptr(ea), lib.ptr(e), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz1), lib.ptr(de1 if rep == 0 else scratch),
                 accumulate, lib.ptr(dPi1), lib.ptr(partial), st)
        lib.call('pamnet_wgrad_edge_enqueue_f32', ctypes.addressof(ctx), int(slots.value), sub(dWm, 2 * D), 3 * D, lib.ptr(db),
                 lib.ptr(dWea), D, lib.ptr(partial))
        lib.call('pamnet_wgrad_flush_f32', ctypes.addressof(ctx), st)
        if rep == 0:
            first = (dWm[:, 2 * D:].clone(), dWea.clone(), db.clone(), dPi1.clone())
    assert torch.equal(first[0], dWm[:, 2 * D:]) and torch.equal(first[1], dWea) and torch.equal(first[2], db)
    assert torch.equal(first[3], dPi1)
    assert torch.isnan(dWm[:, :2 * D]).all()                     # the node blocks of the [d, 3d] gradient are not touched
    # d z, d e: the plain kernel's arithmetic in the plain kernel's order; d P_i: the node sums run on the matrix pipe here
    # (exact 0/1 x piece products, fp32 accumulation) -- same values to fp32 rounding, fixed order, run-to-run identical
    assert maxnorm_err(dPi1.cpu(), dPi0.double().cpu()) < 1e-6 or float(dPi0.abs().max()) == 0.0
    assert torch.equal(dPi0 == 0, dPi1 == 0)                    # nodes without edges: exact zeros
    if m:
        assert torch.equal(dz0, dz1) and torch.equal(de0, de1)
    # fp64 reference of the three gradients
    dd = lambda t: t.double()
    dm = dd(d_agg)[row_of.long()]
    sg = torch.sigmoid(dd(z))
    dz64 = dm * dd(ea) * (sg * (1 + dd(z) * (1 - sg)))
    dea64 = dm * torch.nn.functional.silu(dd(z))
    for got, want in ((dWm[:, 2 * D:], dz64.t() @ dd(e)), (dWea, dea64.t() @ dd(e)), (db, dz64.sum(0))):
        assert torch.isfinite(got).all()
        if m:
            assert maxnorm_err(got.cpu(), want.cpu()) < 3e-6, case
        else:
            assert float(got.abs().max()) == 0.0
    # and inside a deferred sequence: the partial tiles are reduced by the launch of the next batch (both fused forms)
    from pamnet_amd.fused import DeferredWgrad
    rows = 700
    A, dZ = mk(rows, D), mk(rows, D)
    for njobs in (1, 20):                                        # compact descriptors / wide ones
        dw = DeferredWgrad(d_agg)
        outs = [torch.empty(D, D, device=dev) for _ in range(njobs)]
        jobs = [(lib.ptr(dZ), D, lib.ptr(A), D, 0, rows, lib.ptr(o), D, None) for o in outs]
        dw.launch(jobs)                                          # an earlier batch waits for its reduction
        lib.call('pamnet_global_edge_agg_bwd_wg_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), None, lib.ptr(z),
                 lib.ptr(ea), lib.ptr(e), sub(Wm, 2 * D), 3 * D, lib.ptr(Wea), D, lib.ptr(dz1), lib.ptr(de1.clone()), accumulate,
                 lib.ptr(dPi1), lib.ptr(partial), st)
        g2, gea2, db2 = torch.full((D, 3 * D), float('nan'), device=dev), torch.empty(D, D, device=dev), torch.empty(D, device=dev)
        lib.call('pamnet_wgrad_edge_enqueue_f32', ctypes.addressof(dw.ctx), int(slots.value), sub(g2, 2 * D), 3 * D, lib.ptr(db2),
                 lib.ptr(gea2), D, lib.ptr(partial))
        dw.launch(jobs)
        dw.flush()
        assert torch.equal(g2[:, 2 * D:], first[0]) and torch.equal(gea2, first[1]) and torch.equal(db2, first[2])
        assert maxnorm_err(outs[0].cpu(), (dd(dZ).t() @ dd(A)).cpu()) < 3e-6


def test_fused_result_independent_of_batching(dev):
    """A node's sum has one owner and CSR order whatever the launch geometry: a graph evaluated alone and as part of
    a larger batch (different workgroup cuts, different chunk instantiation) gives bitwise identical rows."""
    from pamnet_amd import lib
    rng = np.random.default_rng(9)
    deg_a = rng.integers(4, 25, size=300)
    deg_b = rng.integers(4, 25, size=2500)
    Wm, bm, Wea = _weights(dev, 4)
    sub = lambda w, c0: w.data_ptr() + 4 * c0
    gen = torch.Generator().manual_seed(1)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)

    def run(deg, e, Pi, Pj, init, col_np):
        n = len(deg)
        ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)])).to(torch.int32).to(dev)
        row_of = torch.from_numpy(np.repeat(np.arange(n), deg)).to(torch.int32).to(dev)
        col = torch.from_numpy(col_np).to(torch.int32).to(dev)
        out = torch.empty(n, D, device=dev)
        cuts = None
        lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), e.size(0), n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm),
                 lib.ptr(Wea), D, lib.ptr(Pi), lib.ptr(Pj), lib.ptr(ptr), lib.ptr(row_of), lib.ptr(col), cuts, lib.ptr(init),
                 None, None, lib.ptr(out), lib.stream_of(e))
        return out

    ma, mb = int(deg_a.sum()), int(deg_b.sum())
    na, nb = len(deg_a), len(deg_b)
    e, Pi, Pj, init = mk(ma + mb, D), mk(na + nb, D), mk(na + nb, D), mk(na + nb, D)
    col_a = rng.integers(0, na, size=ma)
    col_b = rng.integers(0, nb, size=mb) + na
    alone = run(deg_a, e[:ma].contiguous(), Pi, Pj, init, col_a)
    both = run(np.concatenate([deg_a, deg_b]), e, Pi, Pj, init, np.concatenate([col_a, col_b]))
    assert torch.equal(alone, both[:na])


@pytest.mark.parametrize('case', ['qm9_like', 'holes', 'tiny', 'no_edges', 'giant'])
def test_local_agg_fwd(dev, case):
    """m_t = m_ji + sum_r m_nb[idx[r]] * s[r];  x2 = x1 + sum_{e -> i} q3[e] * m_t[e]  (local_message_passing.py:49-54)."""
    from pamnet_amd import lib
    from pamnet_amd.ops import segment_sum_raw
    rng = np.random.default_rng(21)
    deg = np.minimum(DEGREE_CASES[case](rng), 40) // 4                # local in-degrees 0..10
    n = len(deg)
    l_ptr, l_row, l_col, m = _csr(deg, n, 2, dev)
    tdeg = rng.integers(0, 9, size=m) if m else np.zeros(0, int)
    if case == 'giant' and m:
        tdeg[m // 2] = 300
    t_ptr, t_row, t_col, t = _csr(tdeg, max(m, 1), 3, dev) if m else (torch.zeros(1, dtype=torch.int32, device=dev), None, None, 0)
    gen = torch.Generator().manual_seed(5)
    mk = lambda r: (0.5 * torch.randn(max(r, 1), D, generator=gen)).to(dev)[:r]
    m_ji, m_nb, q3, s, x1 = mk(m), mk(m), mk(m), mk(t), mk(n)
    m_t = torch.full((max(m, 1), D), float('nan'), device=dev)[:m]
    out = torch.full((n, D), float('nan'), device=dev)
    lib.call('pamnet_local_agg_fwd_f32', lib.ptr(m_ji), lib.ptr(m_nb), lib.ptr(s), lib.ptr(q3), lib.ptr(t_ptr),
             lib.ptr(t_col), lib.ptr(l_ptr), lib.ptr(x1), n, lib.ptr(m_t), lib.ptr(out), lib.stream_of(x1))
    mt64 = m_ji.double().clone()
    if t:
        mt64.index_add_(0, t_row.long(), m_nb.double()[t_col.long()] * s.double())
    out64 = x1.double().clone()
    if m:
        out64.index_add_(0, l_row.long(), q3.double() * mt64)
    assert torch.isfinite(out).all()
    assert maxnorm_err(out.cpu(), out64.cpu()) < 2e-6
    if m:
        assert maxnorm_err(m_t.cpu(), mt64.cpu()) < 2e-6
    out2 = torch.empty_like(out)
    lib.call('pamnet_local_agg_fwd_f32', lib.ptr(m_ji), lib.ptr(m_nb), lib.ptr(s), lib.ptr(q3), lib.ptr(t_ptr),
             lib.ptr(t_col), lib.ptr(l_ptr), lib.ptr(x1), n, None, lib.ptr(out2), lib.stream_of(x1))
    assert torch.equal(out, out2)


@pytest.mark.parametrize('case', ['qm9_like', 'holes', 'tiny', 'giant'])
def test_local_agg_bwd(dev, case):
    """d m_t, d q3, d s, d m_nb of the local aggregations vs fp64 autograd of the same expression."""
    from pamnet_amd import graph as G, lib
    rng = np.random.default_rng(22)
    deg = np.minimum(DEGREE_CASES[case](rng), 40) // 4
    n = len(deg)
    l_ptr, l_row, l_col, m = _csr(deg, n, 2, dev)
    tdeg = rng.integers(0, 9, size=m)
    if case == 'giant':
        tdeg[m // 2] = 300
    t_ptr, t_row, t_col, t = _csr(tdeg, m, 3, dev)
    tT = G.Transpose(t_col, m)
    gen = torch.Generator().manual_seed(6)
    mk = lambda r: (0.5 * torch.randn(r, D, generator=gen)).to(dev)
    m_ji, m_nb, q3, s, d_x2 = mk(m), mk(m), mk(m), mk(t), mk(n)
    v = [x.double().requires_grad_() for x in (m_ji, m_nb, q3, s)]
    mt64 = v[0] + torch.zeros(m, D, dtype=torch.float64, device=dev).index_add(0, t_row.long(), v[1][t_col.long()] * v[3])
    x2 = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add(0, l_row.long(), v[2] * mt64)
    (x2 * d_x2.double()).sum().backward()
    m_t = mt64.detach().float().contiguous()
    d_mt, d_q3, d_mnb = (torch.full((m, D), float('nan'), device=dev) for _ in range(3))
    d_s = torch.full((t, D), float('nan'), device=dev)
    lib.call('pamnet_local_agg_bwd_f32', lib.ptr(d_x2), lib.ptr(l_row), lib.ptr(q3), lib.ptr(m_t), lib.ptr(m_nb),
             lib.ptr(s), lib.ptr(t_ptr), lib.ptr(t_col), lib.ptr(t_row), lib.ptr(tT.ptr), lib.ptr(tT.perm), None, None, m,
             lib.ptr(d_mt), lib.ptr(d_q3), lib.ptr(d_s), lib.ptr(d_mnb), lib.stream_of(d_x2))
    for got, ref in ((d_mt, v[0].grad), (d_mnb, v[1].grad), (d_q3, v[2].grad), (d_s, v[3].grad)):
        assert torch.isfinite(got).all()
        assert maxnorm_err(got.cpu(), ref.cpu()) < 2e-6
    # with the gather's two index hops made ahead of time (pamnet_triplet_transpose_aux_i32: what graph construction hands the
    # engine): the same terms in the same order -- bitwise the same gradients
    if t:
        te, tn = (torch.full((t,), -1, dtype=torch.int32, device=dev) for _ in range(2))
        lib.call('pamnet_triplet_transpose_aux_i32', lib.ptr(tT.perm), lib.ptr(t_row), lib.ptr(l_row), t, lib.ptr(te), lib.ptr(tn),
                 lib.stream_of(d_x2))
        assert torch.equal(te.long(), t_row.long()[tT.perm.long()]) and torch.equal(tn.long(), l_row.long()[te.long()])
        outs2 = [torch.full_like(x, float('nan')) for x in (d_mt, d_q3, d_s, d_mnb)]
        lib.call('pamnet_local_agg_bwd_f32', lib.ptr(d_x2), lib.ptr(l_row), lib.ptr(q3), lib.ptr(m_t), lib.ptr(m_nb),
                 lib.ptr(s), lib.ptr(t_ptr), lib.ptr(t_col), lib.ptr(t_row), lib.ptr(tT.ptr), lib.ptr(tT.perm), lib.ptr(te),
                 lib.ptr(tn), m, lib.ptr(outs2[0]), lib.ptr(outs2[1]), lib.ptr(outs2[2]), lib.ptr(outs2[3]), lib.stream_of(d_x2))
        for a_, b_ in zip(outs2, (d_mt, d_q3, d_s, d_mnb)):
            assert torch.equal(a_, b_)


PP_CASES = dict(DEGREE_CASES)
PP_CASES.update({
    # several hundred rows per workgroup: the ping-pong pipeline in steady state (odd / even group counts, ragged last groups)
    'pdbbind_size': lambda rng: rng.integers(20, 60, size=6000),
    'long_stream': lambda rng: rng.integers(1, 120, size=5000),
    'short_nodes_stream': lambda rng: rng.integers(0, 6, size=60000),       # several node changes per 8-row batch of the walker
    'holes_stream': lambda rng: rng.integers(0, 2, size=40000) * rng.integers(0, 40, size=40000),
})


@pytest.mark.parametrize('save', [True, False])
@pytest.mark.parametrize('with_init', [True, False])
@pytest.mark.parametrize('case', sorted(PP_CASES))
def test_ping_pong_forward_equals_the_chunked_kernel_bit_for_bit(dev, case, save, with_init, monkeypatch):
    """Round 6: pamnet_global_edge_agg_fwd_pp_f32 (the two halves of a workgroup in opposite phases, node sums by a walking wave)
    against the chunked kernel behind pamnet_global_edge_agg_fwd_f32: out, z, ea bit for bit -- same GEMM piece order per
    accumulator, same epilogue expression, node sums in CSR order -- for every degree pattern incl. empty nodes at the ends of a
    workgroup's range, nodes longer than many groups, no edges at all; with and without the precomputed work split."""
    from pamnet_amd import lib
    rng = np.random.default_rng(7)
    deg = PP_CASES[case](rng)
    n = len(deg)
    ptr, row_of, col, m = _csr(deg, n, 9, dev)
    gen = torch.Generator().manual_seed(13)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)
    e, Pi, Pj, init = mk(max(m, 1), D)[:m], mk(n, D), mk(n, D), mk(n, D)
    Wm, bm, Wea = _weights(dev, 4)
    st = lib.stream_of(Pi)
    sub = lambda w, c0: w.data_ptr() + 4 * c0
    cuts_t = torch.full((257,), -1, dtype=torch.int32, device=dev)
    lib.call('pamnet_seg_cuts_i32', lib.ptr(ptr), lib.ptr(row_of), n, m, lib.ptr(cuts_t), None, st)

    def run(entry, cuts):
        z, ea = (torch.full((max(m, 1), D), float('nan'), device=dev)[:m] for _ in range(2))
        out = torch.full((n, D), float('nan'), device=dev)
        lib.call(entry, lib.ptr(e), m, n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D, lib.ptr(Pi), lib.ptr(Pj),
                 lib.ptr(ptr), lib.ptr(row_of), lib.ptr(col), cuts, lib.ptr(init) if with_init else None,
                 lib.ptr(z) if save else None, lib.ptr(ea) if save else None, lib.ptr(out), st)
        torch.cuda.synchronize()
        return out, z, ea

    monkeypatch.setenv('PAMNET_AGG_PP', '0')                          # the plain entry point takes the chunked kernel
    ref = run('pamnet_global_edge_agg_fwd_f32', lib.ptr(cuts_t))
    for cuts in (lib.ptr(cuts_t), None):
        got = run('pamnet_global_edge_agg_fwd_pp_f32', cuts)
        assert torch.isfinite(got[0]).all()
        assert torch.equal(got[0], ref[0]), 'out'
        if save and m:
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), 'saves'
    monkeypatch.setenv('PAMNET_AGG_PP', '1')                          # ... and the ping-pong kernel when told to
    assert torch.equal(run('pamnet_global_edge_agg_fwd_f32', lib.ptr(cuts_t))[0], ref[0])


def test_ping_pong_forward_is_run_to_run_identical_under_load(dev):
    """The ping-pong kernel's phases hand LDS slots between the halves of a workgroup through barriers only: a missed hazard would
    show as run-to-run differences once timing moves.  200 launches at a PDBbind-like size while another stream keeps the memory
    system and some CUs busy: every output (out, z, ea) bit for bit the first launch's."""
    from pamnet_amd import lib
    rng = np.random.default_rng(21)
    deg = rng.integers(15, 70, size=9000)
    n = len(deg)
    ptr, row_of, col, m = _csr(deg, n, 3, dev)
    gen = torch.Generator().manual_seed(17)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)
    e, Pi, Pj, init = mk(m, D), mk(n, D), mk(n, D), mk(n, D)
    Wm, bm, Wea = _weights(dev, 6)
    st = lib.stream_of(Pi)
    sub = lambda w, c0: w.data_ptr() + 4 * c0
    cuts_t = torch.full((257,), -1, dtype=torch.int32, device=dev)
    lib.call('pamnet_seg_cuts_i32', lib.ptr(ptr), lib.ptr(row_of), n, m, lib.ptr(cuts_t), None, st)
    z, ea, out = torch.empty(m, D, device=dev), torch.empty(m, D, device=dev), torch.empty(n, D, device=dev)

    def run():
        lib.call('pamnet_global_edge_agg_fwd_pp_f32', lib.ptr(e), m, n, sub(Wm, 2 * D), 3 * D, lib.ptr(bm), lib.ptr(Wea), D,
                 lib.ptr(Pi), lib.ptr(Pj), lib.ptr(ptr), lib.ptr(row_of), lib.ptr(col), lib.ptr(cuts_t), lib.ptr(init), lib.ptr(z),
                 lib.ptr(ea), lib.ptr(out), st)

    run()
    torch.cuda.synchronize()
    ref = (out.clone(), z.clone(), ea.clone())
    side = torch.cuda.Stream()
    noise_a, noise_b = torch.randn(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
    small = torch.randn(512, 512, device=dev)
    for it in range(200):
        with torch.cuda.stream(side):                        # traffic + a few busy CUs beside every other launch
            if it % 2 == 0:
                noise_b.copy_(noise_a)
            else:
                for _ in range(4):
                    small = torch.tanh(small @ small * 1e-3)
        out.fill_(float('nan'))
        run()
        if it % 20 == 19:
            torch.cuda.synchronize()
            assert torch.equal(out, ref[0]) and torch.equal(z, ref[1]) and torch.equal(ea, ref[2]), it
    torch.cuda.synchronize()
    assert torch.equal(out, ref[0]) and torch.equal(z, ref[1]) and torch.equal(ea, ref[2])


# ---- round 6: the edge-level kernels take ready-made bf16x3 fragment images of their weight slices (row stride 0) ----------
def _edge_images(mats, transposed, dev):
    """pamnet_pack_weights_mixed_f32 kind 1 images of [(tensor, column offset, row stride), ...] -> list of pointers + keepalive."""
    from pamnet_amd import lib
    n = len(mats)
    IMG = 3 * D * D // 2
    images = torch.full((n * IMG,), float('nan'), device=dev)
    W = (ctypes.c_void_p * n)(*[t.data_ptr() + 4 * c0 for t, c0, _ in mats])
    ld = (ctypes.c_int64 * n)(*[s for _, _, s in mats])
    kind = (ctypes.c_int32 * n)(*([1] * n))
    off = (ctypes.c_int64 * n)(*[i * IMG for i in range(n)])
    lib.call('pamnet_pack_weights_mixed_f32', n, W, ld, kind, off, transposed, lib.ptr(images), lib.stream_of(images))
    assert torch.isfinite(images.view(torch.int32).float()).all()
    return [images.data_ptr() + 4 * i * IMG for i in range(n)], images


@pytest.mark.parametrize('case', ['qm9_like', 'tiny', 'mid', 'small3', 'pdbbind_like', 'holes', 'giant'])
def test_fragment_images_give_the_bits_of_the_fp32_matrices(dev, case, monkeypatch):
    """Forward (chunked and ping-pong forms, training and inference) and plain backward of the fused global-edge step with
    W_e / W_ea as pamnet_pack_weights_mixed_f32 images (ld = 0) against the same calls on the fp32 matrices: every output
    bit for bit (the image holds exactly the pieces a wave splits out of its slice)."""
    from pamnet_amd import lib
    rng = np.random.default_rng(5)
    deg = DEGREE_CASES[case](rng)
    n = len(deg)
    ptr, row_of, col, m = _csr(deg, n, 9, dev)
    gen = torch.Generator().manual_seed(17)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)
    e, Pi, Pj, init, d_agg = mk(max(m, 1), D)[:m], mk(n, D), mk(n, D), mk(n, D), mk(n, D)
    Wm, bm, Wea = _weights(dev, 4)
    st = lib.stream_of(Pi)
    (we_f, wea_f), keep_f = _edge_images([(Wm, 2 * D, 3 * D), (Wea, 0, D)], 0, dev)
    (we_b, wea_b), keep_b = _edge_images([(Wm, 2 * D, 3 * D), (Wea, 0, D)], 1, dev)
    plain = (Wm.data_ptr() + 8 * D, 3 * D, Wea.data_ptr(), D)

    def fwd(entry, w, save):
        z, ea = (torch.full((max(m, 1), D), float('nan'), device=dev)[:m] for _ in range(2))
        out = torch.full((n, D), float('nan'), device=dev)
        lib.call(entry, lib.ptr(e), m, n, w[0], w[1], lib.ptr(bm), w[2], w[3], lib.ptr(Pi), lib.ptr(Pj), lib.ptr(ptr),
                 lib.ptr(row_of), lib.ptr(col), None, lib.ptr(init), lib.ptr(z) if save else None, lib.ptr(ea) if save else None,
                 lib.ptr(out), st)
        return z, ea, out

    for entry in ('pamnet_global_edge_agg_fwd_f32', 'pamnet_global_edge_agg_fwd_pp_f32'):
        for save in (True, False):
            a, b = fwd(entry, plain, save), fwd(entry, (we_f, 0, wea_f, 0), save)
            assert torch.equal(a[2], b[2]) and torch.isfinite(b[2]).all()
            if save and m:
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    z, ea, _ = fwd('pamnet_global_edge_agg_fwd_f32', plain, True)

    def bwd(w, acc):
        dz, dea = (torch.full((max(m, 1), D), float('nan'), device=dev)[:m] for _ in range(2))
        d_e = torch.ones(max(m, 1), D, device=dev)[:m]
        dPi = torch.full((n, D), float('nan'), device=dev)
        lib.call('pamnet_global_edge_agg_bwd_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), None, lib.ptr(z),
                 lib.ptr(ea), w[0], w[1], w[2], w[3], lib.ptr(dz), lib.ptr(dea), lib.ptr(d_e), acc, lib.ptr(dPi), st)
        return dz, dea, d_e, dPi

    for acc in (0, 1):
        for x, y in zip(bwd(plain, acc), bwd((we_b, 0, wea_b, 0), acc)):
            assert torch.equal(x, y)
    # one stride zero, the other not: refused; the weight-gradient-forming backward takes matrices only
    with pytest.raises(RuntimeError, match='EINVAL'):
        lib.call('pamnet_global_edge_agg_fwd_f32', lib.ptr(e), m, n, we_f, 0, lib.ptr(bm), Wea.data_ptr(), D, lib.ptr(Pi),
                 lib.ptr(Pj), lib.ptr(ptr), lib.ptr(row_of), lib.ptr(col), None, lib.ptr(init), None, None, lib.ptr(Pi), st)
    if m:
        need = ctypes.c_int64(0)
        lib.call('pamnet_global_edge_agg_wg_floats', m, ctypes.addressof(need), None)
        part = torch.empty(int(need.value), device=dev)
        with pytest.raises(RuntimeError, match='EINVAL'):
            lib.call('pamnet_global_edge_agg_bwd_wg_f32', lib.ptr(d_agg), m, n, lib.ptr(ptr), lib.ptr(row_of), None, lib.ptr(z),
                     lib.ptr(ea), lib.ptr(e), we_b, 0, wea_b, 0, lib.ptr(z), lib.ptr(ea), 0, lib.ptr(Pi), lib.ptr(part), st)


@pytest.mark.parametrize('m', [4316, 300, 36656])
def test_local_edge_forward_on_fragment_images(dev, m):
    """pamnet_local_edge_fwd_f32 with its four slices as images (all strides 0): the outputs of the fp32 matrices bit for bit."""
    from pamnet_amd import lib
    gen = torch.Generator().manual_seed(23)
    mk = lambda *s: (0.5 * torch.randn(*s, generator=gen)).to(dev)
    n = max(m // 2, 8)
    rbf, P = mk(m, D), [mk(n, D) for _ in range(4)]
    Wji, Wkj, Wlr, Wlo, bji, bkj = mk(D, 3 * D) / 8, mk(D, 3 * D) / 8, mk(D, D) / 8, mk(D, D) / 8, mk(D), mk(D)
    row_of = torch.sort(torch.randint(0, n, (m,), generator=gen))[0].to(torch.int32).to(dev)
    col = torch.randint(0, n, (m,), generator=gen).to(torch.int32).to(dev)
    st = lib.stream_of(rbf)
    imgs, keep = _edge_images([(Wji, 2 * D, 3 * D), (Wkj, 2 * D, 3 * D), (Wlr, 0, D), (Wlo, 0, D)], 0, dev)
    parr = lambda ps: (ctypes.c_void_p * len(ps))(*ps)
    iarr = lambda vs: (ctypes.c_int64 * len(vs))(*vs)

    def run(wq, ldq):
        outs = [torch.full((m, D), float('nan'), device=dev) for _ in range(6)]
        lib.call('pamnet_local_edge_fwd_f32', lib.ptr(rbf), m, parr(wq), iarr(ldq), lib.ptr(bji), lib.ptr(bkj),
                 parr([p.data_ptr() for p in P]), lib.ptr(row_of), lib.ptr(col), *[lib.ptr(o) for o in outs], st)
        return outs

    a = run([Wji.data_ptr() + 8 * D, Wkj.data_ptr() + 8 * D, Wlr.data_ptr(), Wlo.data_ptr()], [3 * D, 3 * D, D, D])
    b = run(imgs, [0, 0, 0, 0])
    for x, y in zip(a, b):
        assert torch.isfinite(y).all() and torch.equal(x, y)

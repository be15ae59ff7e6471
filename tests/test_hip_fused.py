"""Fused fp32-MFMA chain kernels (dim=128) vs the plain-PyTorch fp32 formulation of the same ops on the same GPU and
vs an fp64 evaluation.  Tolerance: err(fused, fp64) <= max(2e-6, 2 * err(torch_fp32, fp64)) per tensor."""
import numpy as np
import pytest
import torch

from conftest import maxnorm_err
import torch_formulation as T

pytestmark = pytest.mark.gpu
D = 128


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _ok(a, b32, b64, floor=2e-6):
    e, f = maxnorm_err(a.detach().cpu(), b64.detach().cpu()), maxnorm_err(b32.detach().cpu(), b64.detach().cpu())
    return e <= max(floor, 2 * f), (e, f)


def _run(layer_fn, inputs, params):
    """Outputs and gradients of `layer_fn` (the HIP layer, or its plain-PyTorch statement in tests/torch_formulation.py)."""
    for t in list(inputs) + list(params):
        t.grad = None
    outs = layer_fn()
    gen = torch.Generator().manual_seed(7)
    w = [torch.randn(o.shape, generator=gen, dtype=torch.float64).to(o.dtype).to(o.device) for o in outs]
    sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
    return ([o.detach().clone() for o in outs],
            [None if t.grad is None else t.grad.detach().clone() for t in list(inputs) + list(params)])


@pytest.mark.parametrize('n', [1, 16, 37, 2286])
def test_node_tail_and_pre(dev, n):
    from pamnet_amd import modules
    torch.manual_seed(n)
    g32 = modules.GlobalMP(D).to(dev)
    l32 = modules.LocalMP(D).to(dev)
    x = torch.randn(n, D, device=dev, requires_grad=True)
    r = torch.randn(n, D, device=dev, requires_grad=True)
    for layer in (g32, l32):
        params = [p for p in layer.parameters()]
        of, gf = _run(lambda: modules.update_and_heads(layer, x, r), [x, r], params)
        ot, gt = _run(lambda: T.update_and_heads(layer, x, r), [x, r], params)
        l64 = type(layer)(D).to(dev).double()
        l64.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
        x64, r64 = x.detach().double().requires_grad_(), r.detach().double().requires_grad_()
        p64 = [p for p in l64.parameters()]
        o64, g64 = _run(lambda: T.update_and_heads(l64, x64, r64), [x64, r64], p64)
        names = ['x', 'res_x'] + [k for k, _ in layer.named_parameters()]
        for a, b, c in zip(of, ot, o64):
            ok, info = _ok(a, b, c)
            assert ok, ('out', info)
        used = {id(p) for p in modules.fused.tail_params(layer)} if hasattr(modules, 'fused') else set()
        for nm, a, b, c in zip(names, gf, gt, g64):
            if a is None or c is None:
                continue
            ok, info = _ok(a, b, c, floor=5e-6)
            assert ok, (nm, info)


def _layer_case(dev, n_mol, seed):
    from pamnet_amd import graph as G, synth
    b = synth.qm9_batch(seed, 0, n_mol).to(dev)
    g = G.build_graph('QM9', 5.0, 5.0, 'source_to_target', b.x, b.batch, b.pos, b.edge_index, num_graphs=n_mol)
    gen = torch.Generator().manual_seed(seed)
    mk = lambda r: (0.5 * torch.randn(r, D, generator=gen)).to(dev).requires_grad_()
    return g, mk(g.n), mk(g.glob.m), mk(g.loc.m), mk(g.tp.m)


def _oracle_layer(kind, layer, g, x, e, rbf, sbf, w):
    """fp64 CPU evaluation of the same layer through the oracle (reference formulation with cat / Linear on 3*dim)."""
    from oracle import pamnet_oracle as O
    sd = {('L.' + k): v.detach().cpu().double().requires_grad_() for k, v in layer.state_dict().items()}
    ins = [t.detach().cpu().double().requires_grad_() for t in (x, e, rbf, sbf)]
    if kind == 'global':
        ei = torch.stack([g.glob.col.cpu().long(), g.glob.row_of.cpu().long()])
        outs = O.global_mp(sd, 'L', ins[0], ins[1], ei, 'source_to_target')
    else:
        ei = torch.stack([g.loc.col.cpu().long(), g.loc.row_of.cpu().long()])
        empty_i = torch.zeros(0, dtype=torch.long)
        outs = O.local_mp(sd, 'L', ins[0], ins[2], ins[3], ins[3][:0], g.tp.col.cpu().long(), g.tp.row_of.cpu().long(),
                          empty_i, empty_i, ei)
    outs = [outs[0], outs[1].view(-1), outs[2].view(-1)]
    sum((o * ww.cpu().double()).sum() for o, ww in zip(outs, w)).backward()
    return outs, ins, sd


@pytest.mark.parametrize('n_mol', [3, 128])
@pytest.mark.parametrize('kind', ['global', 'local'])
def test_full_layer_fused_vs_torch_vs_fp64(dev, kind, n_mol):
    """Whole message-passing layer through the fused kernels: outputs, input gradients and every parameter gradient."""
    from pamnet_amd import modules
    torch.manual_seed(11)
    layer = (modules.GlobalMP(D) if kind == 'global' else modules.LocalMP(D)).to(dev)
    g, x, e, rbf, sbf = _layer_case(dev, n_mol, 5)
    ins = [x, e, rbf, sbf]
    params = list(layer.parameters())
    fn = (lambda: layer(x, e, g)) if kind == 'global' else (lambda: layer(x, rbf, sbf, g))
    fn_t = (lambda: T.global_forward(layer, x, e, g)) if kind == 'global' else (lambda: T.local_forward(layer, x, rbf, sbf, g))
    of, gf = _run(fn, ins, params)
    ot, gt = _run(fn_t, ins, params)
    gen = torch.Generator().manual_seed(7)
    w = [torch.randn(o.shape, generator=gen, dtype=torch.float64) for o in of]
    o64, i64, sd64 = _oracle_layer(kind, layer, g, x, e, rbf, sbf, w)
    for a, b, c in zip(of, ot, o64):
        ok, info = _ok(a, b, c, floor=3e-6)
        assert ok, ('out', info)
    g64 = [t.grad for t in i64] + [sd64['L.' + k].grad for k, _ in layer.named_parameters()]
    names = ['x', 'e', 'rbf', 'sbf'] + [k for k, _ in layer.named_parameters()]
    for nm, a, b, c in zip(names, gf, gt, g64):
        if a is None and c is None:
            continue
        assert a is not None and c is not None, nm
        ok, info = _ok(a, b, c, floor=1e-5)
        assert ok, (nm, info)
    # deterministic
    of2, gf2 = _run(fn, ins, params)
    assert all(torch.equal(a, b) for a, b in zip(of, of2))
    assert all(torch.equal(a, b) for a, b in zip(gf, gf2) if a is not None)


@pytest.mark.parametrize('rows', [0, 1, 63, 64, 65, 4316, 40001])
@pytest.mark.parametrize('case', ['rbf16', 'init18', 'sbf42_kind', 'sbf42'])
def test_embed_layers(dev, rows, case):
    """csrc/embed.hip vs torch Linear+SiLU (fp32 and fp64): forward, weight/bias gradients, dx for the rbf layer."""
    import torch.nn as nn
    import torch.nn.functional as F
    from pamnet_amd import fused
    torch.manual_seed(rows + len(case))
    K = {'rbf16': 16, 'init18': 18, 'sbf42_kind': 42, 'sbf42': 42}[case]
    act, bias = case != 'init18', case != 'init18'
    two = case == 'sbf42_kind'
    lin0 = nn.Linear(K, D, bias=bias).to(dev)
    lin1 = nn.Linear(K, D, bias=bias).to(dev) if two else None
    x = torch.randn(rows, K, device=dev, requires_grad=(case == 'rbf16'))
    kind = (torch.rand(rows, device=dev) < 0.4).to(torch.int32) if two else None
    w = torch.randn(rows, D, device=dev, dtype=torch.float64)

    def ref(dtype):
        l0 = nn.Linear(K, D, bias=bias).to(dev).to(dtype)
        l0.load_state_dict({k: v.to(dtype) for k, v in lin0.state_dict().items()})
        xx = x.detach().to(dtype).requires_grad_(x.requires_grad)
        z = F.linear(xx, l0.weight, l0.bias)
        ps = list(l0.parameters())
        if two:
            l1 = nn.Linear(K, D, bias=bias).to(dev).to(dtype)
            l1.load_state_dict({k: v.to(dtype) for k, v in lin1.state_dict().items()})
            z = torch.where(kind.bool().unsqueeze(1), F.linear(xx, l1.weight, l1.bias), z)
            ps += list(l1.parameters())
        y = F.silu(z) if act else z
        (y * w.to(dtype)).sum().backward()
        return y.detach(), [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps], xx.grad

    y = fused.embed(x, lin0, lin1, kind=kind, act=act)
    (y * w.float()).sum().backward()
    ps = list(lin0.parameters()) + (list(lin1.parameters()) if two else [])
    y32, g32, dx32 = ref(torch.float32)
    y64, g64, dx64 = ref(torch.float64)
    assert y.shape == (rows, D)
    if rows == 0:
        for p in ps:
            assert float(p.grad.abs().max()) == 0.0
        return
    ok, info = _ok(y, y32, y64)
    assert ok, ('y', info)
    for p, a, b in zip(ps, g32, g64):
        ok, info = _ok(p.grad, a, b, floor=5e-6)
        assert ok, (tuple(p.shape), info)
    if case == 'rbf16':
        ok, info = _ok(x.grad, dx32, dx64)
        assert ok, ('dx', info)
    else:
        assert x.grad is None


def _bessel_rows(dist, freq, cutoff):
    """layers/basic.py:36-51,74-76 in torch (p = 5 envelope)."""
    x = (dist / cutoff).unsqueeze(-1)
    u = torch.where(x < 1.0, 1.0 / x - 21.0 * x ** 5 + 35.0 * x ** 6 - 15.0 * x ** 7, torch.zeros_like(x))
    return u * torch.sin(freq * x)


@pytest.mark.parametrize('sizes', [(0, 0, 0, 0), (1, 1, 1, 1), (63, 700, 65, 37), (4316, 32888, 17640, 2286),
                                    (300, 70001, 40000, 513)])
@pytest.mark.parametrize('variant', ['types_two', 'init18_two', 'types_one'])
def test_input_stage(dev, sizes, variant):
    """pamnet_embed_multi_*: Bessel rows + mlp_rbf_l/g, the sbf embeddings (two weight sets by row kind, or one) and the
    type-table rows / init_linear in one forward launch and two backward launches, against torch (fp32 and fp64):
    outputs and every gradient, including the Bessel frequencies whose [rows, 16] gradient is never materialised."""
    import math
    import torch.nn as nn
    import torch.nn.functional as F
    from pamnet_amd import fused
    e_l, e_g, tp, n = sizes
    torch.manual_seed(sum(sizes) + len(variant))
    two, init18 = variant != 'types_one', variant == 'init18_two'
    cl, cg = 2.0, 5.0
    dist_l = (torch.rand(e_l, device=dev) * 1.9 + 0.05)
    dist_g = (torch.rand(e_g, device=dev) * 5.5 + 0.05)                   # some beyond the cutoff: envelope 0
    sbf = torch.randn(tp, 42, device=dev)
    kind = (torch.rand(tp, device=dev) < 0.4).to(torch.int32) if two else None
    freq_l = nn.Parameter((torch.arange(1, 17, device=dev) * math.pi + 0.01 * torch.randn(16, device=dev)).float())
    freq_g = nn.Parameter((torch.arange(1, 17, device=dev) * math.pi + 0.01 * torch.randn(16, device=dev)).float())
    lin_l, lin_g = nn.Linear(16, D).to(dev), nn.Linear(16, D).to(dev)
    lin_a, lin_b = nn.Linear(42, D).to(dev), (nn.Linear(42, D).to(dev) if two else None)
    layers = [(None, dist_l, cl, None, True, True), (None, dist_g, cg, None, True, True), (sbf, None, None, kind, True, True)]
    params = [freq_l, lin_l.weight, lin_l.bias, freq_g, lin_g.weight, lin_g.bias, lin_a.weight, lin_a.bias]
    if two:
        params += [lin_b.weight, lin_b.bias]
    types = feats = None
    if init18:
        feats = torch.randn(n, 18, device=dev)
        init = nn.Linear(18, D, bias=False).to(dev)
        layers.append((feats, None, None, None, False, False))
        params.append(init.weight)
    else:
        table = nn.Parameter(torch.randn(5, D, device=dev))
        types = torch.randint(0, 5, (n,), device=dev, dtype=torch.int32)
        params.append(table)
    ws = [torch.randn(r, D, device=dev, dtype=torch.float64) for r in (e_l, e_g, tp, n)]

    def ref(dtype):
        ps = [p.detach().to(dtype).requires_grad_(True) for p in params]
        it = iter(ps)
        outs = []
        for dist, c in ((dist_l, cl), (dist_g, cg)):
            f, W, b = next(it), next(it), next(it)
            outs.append(F.silu(F.linear(_bessel_rows(dist.to(dtype), f, c), W, b)))
        W0, b0 = next(it), next(it)
        z = F.linear(sbf.to(dtype), W0, b0)
        if two:
            W1, b1 = next(it), next(it)
            z = torch.where(kind.bool().unsqueeze(1), F.linear(sbf.to(dtype), W1, b1), z)
        outs.append(F.silu(z))
        last = next(it)
        outs.append(F.linear(feats.to(dtype), last) if init18 else last[types.long()])
        sum((o * w.to(dtype)).sum() for o, w in zip(outs, ws)).backward()
        return [o.detach() for o in outs], [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]

    outs = fused.input_stage(fused.InputSpec(layers, types), params)
    sum((o * w.float()).sum() for o, w in zip(outs, ws)).backward()
    o32, g32 = ref(torch.float32)
    o64, g64 = ref(torch.float64)
    for o, r in zip(outs, (e_l, e_g, tp, n)):
        assert o.shape == (r, D)
    for k, (a, b, c) in enumerate(zip(outs, o32, o64)):
        if a.numel():
            ok, info = _ok(a, b, c)
            assert ok, ('out', k, info)
    for k, (p, b, c) in enumerate(zip(params, g32, g64)):
        assert p.grad is not None and p.grad.shape == p.shape
        if float(c.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        ok, info = _ok(p.grad, b, c, floor=1e-5)
        assert ok, ('grad', k, tuple(p.shape), info)
    # deterministic, and identical to the one-layer-per-launch entry points on the plain layers
    g1 = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    outs2 = fused.input_stage(fused.InputSpec(layers, types), params)
    sum((o * w.float()).sum() for o, w in zip(outs2, ws)).backward()
    assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, params))
    y = fused.embed(sbf, lin_a, lin_b, kind=kind)
    assert torch.equal(y, outs[2])


@pytest.mark.parametrize('width', [128, 16])
def test_bessel_gradients_at_the_band_edges(dev, width):
    """the backward of the Bessel-row embeddings recomputes sin / cos on the hardware units (gemm_core.h sin_turns)
    while the forward keeps sinf.  Bound the weight and frequency gradients against fp64 where that matters most: the highest
    frequency (16 pi and a trained offset), edge lengths crowded at x -> 0 (the envelope's 1/x pole) and at x -> 1 (the envelope
    and its derivative vanish: pure cancellation), at the wide (dim 128, embed.hip) and the narrow (dim 16, narrow_core.h) form."""
    import math
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(3 + width)
    c = 5.0
    m = 30000
    x = torch.cat([torch.rand(m // 3) * 0.02 + 1e-3, 1.0 - torch.rand(m // 3) * 0.02, torch.rand(m - 2 * (m // 3))])
    dist = (x * c).to(dev)
    freq = nn.Parameter((torch.arange(1, 17, device=dev) * math.pi + 0.05 * torch.randn(16, device=dev)).float())
    lin = nn.Linear(16, width).to(dev)
    w = torch.randn(m, width, device=dev, dtype=torch.float64)
    params = [freq, lin.weight, lin.bias]

    def ref(dtype):
        f, W, b = (p.detach().to(dtype).requires_grad_(True) for p in params)
        (F.silu(F.linear(_bessel_rows(dist.to(dtype), f, c), W, b)) * w.to(dtype)).sum().backward()
        return [f.grad, W.grad, b.grad]

    if width == 128:
        from pamnet_amd import fused
        outs = fused.input_stage(fused.InputSpec([(None, dist, c, None, True, True)], None), params)
        y = outs[0]
    else:
        from pamnet_amd import narrow
        y = narrow.embed_rbf(dist, freq, c, lin)
    (y * w.float()).sum().backward()
    g32, g64 = ref(torch.float32), ref(torch.float64)
    worst = 0.0
    for name, p, a, b in zip(('freq', 'weight', 'bias'), params, g32, g64):
        e, floor = maxnorm_err(p.grad.cpu(), b.cpu()), maxnorm_err(a.cpu(), b.cpu())
        worst = max(worst, e)
        assert e <= max(1e-5, 2 * floor), (name, e, floor)
    # the column of the highest frequency on its own scale (not hidden behind the larger low-frequency columns)
    e16 = maxnorm_err(lin.weight.grad[:, 15].cpu(), g64[1][:, 15].cpu())
    f16 = abs(float(freq.grad[15]) - float(g64[0][15])) / max(abs(float(g64[0][15])), 1e-300)
    print('Bessel-row gradients at the band edges (dim %d): worst tensor %.1e, dW[:, n=16] %.1e, dfreq[16] %.1e vs fp64'
          % (width, worst, e16, f16))
    assert e16 <= 1e-5 and f16 <= max(1e-5, 2 * abs(float(g32[0][15]) - float(g64[0][15])) / abs(float(g64[0][15])))


@pytest.mark.parametrize('waves', ['4', '8'])
def test_edge_kernel_geometries(waves):
    """Both workgroup geometries of the edge kernels (8 waves, one per CU / paired 4-wave workgroups) give the same
    parity: the layer-level and model-level checks re-run in a subprocess with the geometry forced."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PAMNET_EDGE_WAVES=waves)
    out = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.join(here, 'test_hip_fused.py'),
                          os.path.join(here, 'test_hip_model.py'), '-k',
                          'full_layer or fused_engine or forward_vs_reference or gradients_vs_reference'],
                         capture_output=True, text=True, env=env, timeout=900, cwd=os.path.dirname(here))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]


@pytest.mark.parametrize('n', [1, 37, 2286])
def test_node_tail_fwd_bf16x6(dev, n):
    """The forward node chain on the bf16 matrix pipe (packed = 2: bf16x3 weight images, activations split exactly into
    three bf16 pieces as they are produced, six piece products per product -- csrc/node_tail.hip) against the fp32-MFMA
    form (packed = 1) on the same inputs, with the next layer's head fused behind it: every output within 2e-6 of the
    fp32-MFMA result (both are fp32-accurate; the summation order differs), ragged last tile included."""
    import ctypes
    from pamnet_amd import lib
    torch.manual_seed(n)
    P, PA10, PA4 = ctypes.c_void_p, ctypes.c_void_p * 10, ctypes.c_void_p * 4
    x2, rx = torch.randn(n, D, device=dev), torch.randn(n, D, device=dev)
    W = [torch.randn(D, D, device=dev) * 0.08 for _ in range(10)]
    b = [torch.randn(D, device=dev) * 0.1 for _ in range(10)]
    Wx1, bx1 = torch.randn(D, D, device=dev) * 0.08, torch.randn(D, device=dev) * 0.1
    Wm = torch.randn(D, 3 * D, device=dev) * 0.08                     # the two node-side blocks of a [d, 3d] message weight
    w_out, b_out, w_att = torch.randn(D, device=dev), torch.zeros(1, device=dev), torch.randn(D, device=dev)
    st = lib.stream_of(x2)
    mats = W + [Wx1]
    srcs = [m.data_ptr() for m in mats] + [Wm.data_ptr(), Wm.data_ptr() + 4 * D]
    lds = [D] * 11 + [3 * D, 3 * D]
    res = {}
    for packed, fn, stride in ((1, 'pamnet_pack_weights_f32', 16384), (2, 'pamnet_pack_weights_bf16x3', 24576)):
        img = torch.empty(13 * stride, device=dev)
        lib.call(fn, 13, (P * 13)(*srcs), (ctypes.c_int64 * 13)(*lds), 0, lib.ptr(img), st)
        ip = [img.data_ptr() + 4 * stride * k for k in range(13)]
        Z, R = torch.zeros(10, n, D, device=dev), torch.zeros(2, n, D, device=dev)
        xo, Zx1, x1 = (torch.zeros(n, D, device=dev) for _ in range(3))
        Pn = torch.zeros(2, n, D, device=dev)
        lib.call('pamnet_node_tail_fwd_f32', lib.ptr(x2), lib.ptr(rx), n, PA10(*ip[:10]), PA10(*[t.data_ptr() for t in b]),
                 lib.ptr(w_out), lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(xo), None, None, ip[10],
                 lib.ptr(bx1), PA4(ip[11], ip[12], None, None), 3 * D, 2, lib.ptr(Zx1), lib.ptr(x1), lib.ptr(Pn), packed, st)
        res[packed] = dict(x_out=xo, Z=Z[:7], R=R, Zx1=Zx1, x1=x1, P=Pn)
    torch.cuda.synchronize()
    # fp64 statement of the chain for the output that matters most
    h = torch.nn.functional.silu
    xd, rd = x2.double(), rx.double()
    lin = lambda k, t: t @ W[k].double().t() + b[k].double()
    h0 = h(lin(0, xd))
    r1 = h(lin(2, h(lin(1, h0)))) + h0 + rd
    r2 = h(lin(4, h(lin(3, r1)))) + r1
    r3 = h(lin(6, h(lin(5, r2)))) + r2
    e1 = maxnorm_err(res[1]['x_out'].cpu(), r3.cpu())
    e2 = maxnorm_err(res[2]['x_out'].cpu(), r3.cpu())
    assert e2 <= max(2e-6, 2 * e1), (e2, e1)
    for k in res[1]:
        assert maxnorm_err(res[2][k].cpu(), res[1][k].cpu()) < 3e-6, k


@pytest.mark.parametrize('rows,edges', [(17640, 4316), (127922, 36890), (37, 5), (1, 1), (700, 9000), (40, 0), (0, 33)])
@pytest.mark.parametrize('acc', [0, 1])
def test_local_backward_pair_equals_the_two_launches(dev, rows, edges, acc):
    """pamnet_local_bwd_pair_f32 (the triplet / pair MLP's backward and the local edge stage's as one launch, CUs split by
    work) against pamnet_mlp2_bwd_f32 + pamnet_local_edge_bwd_f32: every output bit for bit, at the QM9 and PDBbind
    batch shapes, tiny and lopsided ones, empty lists, with and without accumulation into d_sbf / d_rbf."""
    import ctypes
    from pamnet_amd import lib
    D = 128
    gen = torch.Generator(device='cpu').manual_seed(rows * 7 + edges)
    rnd = lambda *s: (torch.randn(*s, generator=gen) * 0.5).to(dev)
    dy, z1, z2 = rnd(rows, D), rnd(rows, D), rnd(rows, D)
    W1, W2 = rnd(D, D) / 8, rnd(D, D) / 8
    d_mji, d_mnb, d_q3, z_ji, z_kj, q2 = (rnd(edges, D) for _ in range(6))
    Wq = [rnd(D, 3 * D) / 8, rnd(D, 3 * D) / 8, rnd(D, D) / 8, rnd(D, D) / 8]
    wq = (ctypes.c_void_p * 4)(Wq[0].data_ptr() + 8 * D, Wq[1].data_ptr() + 8 * D, Wq[2].data_ptr(), Wq[3].data_ptr())
    ldq = (ctypes.c_int64 * 4)(3 * D, 3 * D, D, D)
    dx0, drbf0 = rnd(rows, D), rnd(edges, D)
    st = lib.stream_of(W1)

    def outs():
        return ([torch.full((rows, D), float('nan'), device=dev) for _ in range(2)] + [dx0.clone()],
                [torch.full((edges, D), float('nan'), device=dev) for _ in range(3)] + [drbf0.clone()])
    (dz1, dz2, dx), (dzji, dzkj, dq2, drbf) = outs()
    lib.call('pamnet_mlp2_bwd_f32', lib.ptr(dy), rows, lib.ptr(z1), lib.ptr(z2), lib.ptr(W1), lib.ptr(W2), lib.ptr(dz1),
             lib.ptr(dz2), lib.ptr(dx), acc, st)
    lib.call('pamnet_local_edge_bwd_f32', lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji), lib.ptr(z_kj),
             lib.ptr(q2), wq, ldq, lib.ptr(dzji), lib.ptr(dzkj), lib.ptr(dq2), lib.ptr(drbf), acc, st)
    (pz1, pz2, px), (pji, pkj, pq2, prbf) = outs()
    lib.call('pamnet_local_bwd_pair_f32', lib.ptr(dy), rows, lib.ptr(z1), lib.ptr(z2), lib.ptr(W1), lib.ptr(W2), lib.ptr(pz1),
             lib.ptr(pz2), lib.ptr(px), acc, lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji),
             lib.ptr(z_kj), lib.ptr(q2), wq, ldq, lib.ptr(pji), lib.ptr(pkj), lib.ptr(pq2), lib.ptr(prbf), acc, st)
    for a, b in ((dz1, pz1), (dz2, pz2), (dx, px), (dzji, pji), (dzkj, pkj), (dq2, pq2), (drbf, prbf)):
        assert torch.equal(a, b) and not bool(torch.isnan(a).any())
    if rows == 0 or edges == 0:
        return
    # round 6: the same launch on weight images (pamnet_pack_weights_mixed_f32, transposed, kind 1: bf16x3 fragments): W1 / W2
    # (flag PAMNET_WEIGHT_IMAGES on accumulate_dx) and the four local slices (strides 0)
    IMG1 = 3 * D * D // 2
    images = torch.full((6 * IMG1,), float('nan'), device=dev)
    srcs = (ctypes.c_void_p * 6)(W1.data_ptr(), W2.data_ptr(), *[wq[i] for i in range(4)])
    lds = (ctypes.c_int64 * 6)(D, D, 3 * D, 3 * D, D, D)
    kinds = (ctypes.c_int32 * 6)(1, 1, 1, 1, 1, 1)
    offs = (ctypes.c_int64 * 6)(*[i * IMG1 for i in range(6)])
    lib.call('pamnet_pack_weights_mixed_f32', 6, srcs, lds, kinds, offs, 1, lib.ptr(images), st)
    ip = lambda i: images.data_ptr() + 4 * offs[i]
    (iz1, iz2, ix), (iji, ikj, iq2, irbf) = outs()
    lib.call('pamnet_local_bwd_pair_f32', lib.ptr(dy), rows, lib.ptr(z1), lib.ptr(z2), ip(0), ip(1), lib.ptr(iz1),
             lib.ptr(iz2), lib.ptr(ix), acc | 2, lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji),
             lib.ptr(z_kj), lib.ptr(q2), (ctypes.c_void_p * 4)(ip(2), ip(3), ip(4), ip(5)), (ctypes.c_int64 * 4)(0, 0, 0, 0),
             lib.ptr(iji), lib.ptr(ikj), lib.ptr(iq2), lib.ptr(irbf), acc, st)
    for a, b in ((dz1, iz1), (dz2, iz2), (dx, ix), (dzji, iji), (dzkj, ikj), (dq2, iq2), (drbf, irbf)):
        assert torch.equal(a, b)
    # the stand-alone local edge backward takes the same images; mixed strides are refused
    (_, _, _), (sji, skj, sq2, srbf) = outs()
    lib.call('pamnet_local_edge_bwd_f32', lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji), lib.ptr(z_kj),
             lib.ptr(q2), (ctypes.c_void_p * 4)(ip(2), ip(3), ip(4), ip(5)), (ctypes.c_int64 * 4)(0, 0, 0, 0), lib.ptr(sji),
             lib.ptr(skj), lib.ptr(sq2), lib.ptr(srbf), acc, st)
    for a, b in ((dzji, sji), (dzkj, skj), (dq2, sq2), (drbf, srbf)):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match='EINVAL'):
        lib.call('pamnet_local_edge_bwd_f32', lib.ptr(d_mji), lib.ptr(d_mnb), lib.ptr(d_q3), edges, lib.ptr(z_ji),
                 lib.ptr(z_kj), lib.ptr(q2), wq, (ctypes.c_int64 * 4)(0, 3 * D, D, D), lib.ptr(sji), lib.ptr(skj), lib.ptr(sq2),
                 lib.ptr(srbf), acc, st)


@pytest.mark.gpu
@pytest.mark.parametrize('nblk', [0, 2, 4])
@pytest.mark.parametrize('save', [True, False])
def test_lean_forward_chain_equals_the_parked_one(dev, nblk, save):
    """Batches of more than 512 row tiles run the forward chain with three workgroups per CU and the saves written straight
    from the accumulators (node_tail_fwd_lean_kernel); same arithmetic per row, so a launch over 8 200 rows (lean) must
    reproduce, bit for bit, the first 8 000 rows as a launch of their own (500 tiles: the parked form) computes them."""
    from pamnet_amd import lib
    from pamnet_amd.fused import _parr, _iarr
    torch.manual_seed(5 + nblk)
    n_big, n_small = 8200 + 7, 8000
    NW = 15
    W = [(torch.randn(D, D, device=dev) * 0.08) for _ in range(NW)]
    b = [torch.randn(D, device=dev) * 0.1 for _ in range(11)]
    images = torch.empty(NW, D * D, device=dev)
    lib.call('pamnet_pack_weights_f32', NW, _parr(W), _iarr([D] * NW), 0, lib.ptr(images), lib.stream_of(images))
    img = [images[i] for i in range(NW)]
    x2, rx = torch.randn(n_big, D, device=dev), torch.randn(n_big, D, device=dev)
    w_out, b_out, w_att = torch.randn(D, device=dev), torch.zeros(1, device=dev), torch.randn(D, device=dev)

    def run(n):
        Z = torch.full((10, n, D), float('nan'), device=dev) if save else None
        R = torch.full((2, n, D), float('nan'), device=dev) if save else None
        xo = torch.full((n, D), float('nan'), device=dev)
        zx1 = torch.full((n, D), float('nan'), device=dev) if (save and nblk) else None
        x1 = torch.full((n, D), float('nan'), device=dev) if nblk else None
        P = torch.full((max(nblk, 1), n, D), float('nan'), device=dev) if nblk else None
        lib.call('pamnet_node_tail_fwd_f32', lib.ptr(x2[:n].contiguous()), lib.ptr(rx[:n].contiguous()), n, _parr(img[:10]),
                 _parr(b[:10]), lib.ptr(w_out), lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(xo), None, None,
                 lib.ptr(img[10]) if nblk else None, lib.ptr(b[10]) if nblk else None, _parr(img[11:11 + nblk]) if nblk else None,
                 D, nblk, lib.ptr(zx1), lib.ptr(x1), lib.ptr(P), 1, lib.stream_of(x2))
        return Z, R, xo, zx1, x1, P

    big, small = run(n_big), run(n_small)
    torch.cuda.synchronize()
    for name, a, c in zip(['Z', 'R', 'x_out', 'Zx1', 'x1', 'P'], big, small):
        if a is None:
            continue
        a = a[..., :n_small, :] if a.dim() == 3 else a[:n_small]
        if name == 'Z':
            a, c = a[:7], c[:7]                                  # (deferred heads: z7..z9 are node_heads_fwd's)
        assert torch.equal(a, c), name
        assert not torch.isnan(a).any(), name
    tail = big[2][n_small:]
    assert not torch.isnan(tail).any() and tail.abs().max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize('nblk', [0, 2, 4])
def test_lean_backward_chain_equals_the_parked_one(dev, nblk):
    """node_tail_bwd_lean_kernel (more than 256 row tiles: two co-resident workgroups per CU, the per-element state in
    registers) against node_tail_bwd_kernel on the same rows: 4 200 rows (lean) reproduce, bit for bit, what a launch over the
    first 4 000 of them (250 tiles: the parked form) computes.  nblk = 0: the chain alone; 2 / 4: with the next head's backward
    in front (in-place d_x2 / d_resx as the engine calls it)."""
    from pamnet_amd import lib
    from pamnet_amd.fused import _parr, _iarr
    torch.manual_seed(11 + nblk)
    n_big, n_small = 4200 + 5, 4000
    NW = 12
    W = [(torch.randn(D, D, device=dev) * 0.08) for _ in range(NW)]
    images = torch.empty(NW, D * D, device=dev)
    lib.call('pamnet_pack_weights_f32', NW, _parr(W), _iarr([D] * NW), 1, lib.ptr(images), lib.stream_of(images))
    img = [images[i] for i in range(NW)]
    Z = torch.randn(10, n_big, D, device=dev)
    g_head, dP = torch.randn(n_big, D, device=dev), torch.randn(4, n_big, D, device=dev)
    dx1, dadd, zx1 = torch.randn(n_big, D, device=dev), torch.randn(n_big, D, device=dev), torch.randn(n_big, D, device=dev)
    d_xout = torch.randn(n_big, D, device=dev)

    def run(n):
        Zn = Z[:, :n].contiguous()
        dZ = torch.full((10, n, D), float('nan'), device=dev)
        if nblk == 0:
            dx2, drx = torch.full((n, D), float('nan'), device=dev), torch.full((n, D), float('nan'), device=dev)
            lib.call('pamnet_node_tail_main_bwd_f32', lib.ptr(d_xout[:n].contiguous()), lib.ptr(g_head[:n].contiguous()), n,
                     _parr(img[:7]), lib.ptr(Zn), lib.ptr(dZ), lib.ptr(dx2), lib.ptr(drx), 1, lib.stream_of(Z))
            return dZ[:7], dx2, drx, None
        dx2, drx = dx1[:n].clone(), dadd[:n].clone()              # in place, as pamnet_stack_bwd_f32 calls it
        dzx1 = torch.full((n, D), float('nan'), device=dev)
        lib.call('pamnet_node_pre_tail_bwd_f32', lib.ptr(dP[:nblk, :n].contiguous()), lib.ptr(dx2), lib.ptr(drx), n,
                 lib.ptr(img[7]), _parr(img[8:8 + nblk]), nblk, lib.ptr(zx1[:n].contiguous()), lib.ptr(dzx1),
                 lib.ptr(g_head[:n].contiguous()), _parr(img[:7]), lib.ptr(Zn), lib.ptr(dZ), lib.ptr(dx2), lib.ptr(drx), None,
                 lib.stream_of(Z))
        return dZ[:7], dx2, drx, dzx1

    big, small = run(n_big), run(n_small)
    torch.cuda.synchronize()
    for name, a, c in zip(['dZ', 'd_x2', 'd_resx', 'dZx1'], big, small):
        if a is None:
            continue
        a = a[:, :n_small] if a.dim() == 3 else a[:n_small]
        assert not torch.isnan(a).any(), name
        assert torch.equal(a, c), name
    assert not torch.isnan(big[1][n_small:]).any() and big[1][n_small:].abs().max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize('nblk', [0, 2, 4])
@pytest.mark.parametrize('n', [1, 37, 2286])
def test_backward_chain_on_the_bf16_pipe(dev, nblk, n):
    """Round 6: node_tail_bwd_bf16_kernel (packed == 2 / nblk | PAMNET_CHAIN_PIECES: bf16x3 images in the transposed orientation,
    every GEMM six bf16 piece products) against the fp32-MFMA chain on the same inputs: every output within 2e-6 of the
    outputs' scale (seven chained GEMMs and, with nblk > 0, the head's in front; the error of one bf16x6 GEMM against an fp32
    one is ~3e-7), padding rows never read, in-place d_x2 / d_resx as the engine calls it."""
    from pamnet_amd import lib
    from pamnet_amd.fused import _parr, _iarr
    PIECES = 16                                                   # PAMNET_CHAIN_PIECES
    torch.manual_seed(23 + nblk + n)
    NW = 12
    W = [(torch.randn(D, D, device=dev) * 0.08) for _ in range(NW)]
    img32 = torch.empty(NW, D * D, device=dev)
    img16 = torch.empty(NW, 3 * D * D // 2, device=dev)
    lib.call('pamnet_pack_weights_f32', NW, _parr(W), _iarr([D] * NW), 1, lib.ptr(img32), lib.stream_of(img32))
    lib.call('pamnet_pack_weights_bf16x3', NW, _parr(W), _iarr([D] * NW), 1, lib.ptr(img16), lib.stream_of(img16))
    Z = torch.randn(10, n, D, device=dev)
    g_head, dP = torch.randn(n, D, device=dev), torch.randn(4, n, D, device=dev)
    dx1, dadd, zx1 = torch.randn(n, D, device=dev), torch.randn(n, D, device=dev), torch.randn(n, D, device=dev)
    d_xout = torch.randn(n, D, device=dev)

    def run(images, pieces):
        img = [images[i] for i in range(NW)]
        dZ = torch.full((10, n, D), float('nan'), device=dev)
        if nblk == 0:
            dx2, drx = torch.full((n, D), float('nan'), device=dev), torch.full((n, D), float('nan'), device=dev)
            lib.call('pamnet_node_tail_main_bwd_f32', lib.ptr(d_xout), lib.ptr(g_head), n, _parr(img[:7]), lib.ptr(Z), lib.ptr(dZ),
                     lib.ptr(dx2), lib.ptr(drx), 2 if pieces else 1, lib.stream_of(Z))
            return dZ[:7], dx2, drx, None
        dx2, drx = dx1.clone(), dadd.clone()
        dzx1 = torch.full((n, D), float('nan'), device=dev)
        lib.call('pamnet_node_pre_tail_bwd_f32', lib.ptr(dP[:nblk].contiguous()), lib.ptr(dx2), lib.ptr(drx), n, lib.ptr(img[7]),
                 _parr(img[8:8 + nblk]), nblk | (PIECES if pieces else 0), lib.ptr(zx1), lib.ptr(dzx1), lib.ptr(g_head),
                 _parr(img[:7]), lib.ptr(Z), lib.ptr(dZ), lib.ptr(dx2), lib.ptr(drx), None, lib.stream_of(Z))
        return dZ[:7], dx2, drx, dzx1

    ref, got = run(img32, False), run(img16, True)
    torch.cuda.synchronize()
    for name, a, c in zip(['dZ', 'd_x2', 'd_resx', 'dZx1'], got, ref):
        if a is None:
            continue
        assert not torch.isnan(a).any(), name
        err = (a - c).abs().max().item() / c.abs().max().item()
        assert err < 2e-6, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize('tile0,ntiles,wgs', [(0, 1103, 113), (551, 552, 113), (3, 7, 5), (0, 1, 1)])
def test_riders_of_the_bf16_forward_chain(dev, tile0, ntiles, wgs):
    """Round 6: the rider workgroups of the bf16x6 forward chain launch (packed = 2) run the triplet / pair MLP's own 8-wave
    geometry.  Row tiles [tile0, tile0 + ntiles) of the MLP computed by `wgs` riders must equal, bit for bit, what
    pamnet_mlp2_fwd_f32 computes for those rows (same per-row arithmetic), rows outside stay untouched, and the chain's own
    outputs must be the bits of the launch without riders."""
    import ctypes
    from pamnet_amd import lib
    torch.manual_seed(tile0 + ntiles)
    n, tp = 2286, 17640
    P, PA10, PA4 = ctypes.c_void_p, ctypes.c_void_p * 10, ctypes.c_void_p * 4
    x2, rx = torch.randn(n, D, device=dev), torch.randn(n, D, device=dev)
    W = [torch.randn(D, D, device=dev) * 0.08 for _ in range(15)]
    b = [torch.randn(D, device=dev) * 0.1 for _ in range(11)]
    w_out, b_out, w_att = torch.randn(D, device=dev), torch.zeros(1, device=dev), torch.randn(D, device=dev)
    st = lib.stream_of(x2)
    img = torch.empty(15 * 24576, device=dev)
    lib.call('pamnet_pack_weights_bf16x3', 15, (P * 15)(*[m.data_ptr() for m in W]), (ctypes.c_int64 * 15)(*([D] * 15)), 0,
             lib.ptr(img), st)
    ip = [img.data_ptr() + 4 * 24576 * k for k in range(15)]
    sbf = torch.randn(tp, D, device=dev)
    M = [torch.randn(D, D, device=dev) * 0.08, torch.randn(D, device=dev) * 0.1, torch.randn(D, D, device=dev) * 0.08,
         torch.randn(D, device=dev) * 0.1]

    def chain(riding):
        Z, R = torch.zeros(10, n, D, device=dev), torch.zeros(2, n, D, device=dev)
        xo, Zx1, x1 = (torch.zeros(n, D, device=dev) for _ in range(3))
        Pn = torch.zeros(4, n, D, device=dev)
        mo = [torch.full((tp, D), 7.0, device=dev) for _ in range(3)]
        lib.call('pamnet_node_tail_fwd_rider_f32', lib.ptr(x2), lib.ptr(rx), n, PA10(*ip[:10]), PA10(*[t.data_ptr() for t in b[:10]]),
                 lib.ptr(w_out), lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(xo), ip[10], lib.ptr(b[10]),
                 PA4(*ip[11:15]), D, 4, lib.ptr(Zx1), lib.ptr(x1), lib.ptr(Pn), lib.ptr(sbf), tp, tile0, ntiles if riding else 0,
                 PA4(*[t.data_ptr() for t in M]), (P * 3)(*[t.data_ptr() for t in mo]), wgs, 2, st)
        return dict(Z=Z[:7], R=R, x_out=xo, Zx1=Zx1, x1=x1, P=Pn), mo

    plain, _ = chain(False)
    ridden, mo = chain(True)
    ref = [torch.empty(tp, D, device=dev) for _ in range(3)]
    lib.call('pamnet_mlp2_fwd_f32', lib.ptr(sbf), tp, lib.ptr(M[0]), lib.ptr(M[1]), lib.ptr(M[2]), lib.ptr(M[3]), lib.ptr(ref[0]),
             lib.ptr(ref[1]), lib.ptr(ref[2]), st)
    torch.cuda.synchronize()
    for k in plain:
        assert torch.equal(plain[k], ridden[k]), k
    r0, r1 = tile0 * 16, min(tp, (tile0 + ntiles) * 16)
    for got, want in zip(mo, ref):
        assert torch.equal(got[r0:r1], want[r0:r1])
        assert bool((got[:r0] == 7.0).all()) and bool((got[r1:] == 7.0).all())

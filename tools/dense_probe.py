"""csrc/dense.hip at the shapes a dim = 256 model would run it (QM9 batch of 128 molecules: n = 2 286 nodes, E_g = 32 888,
T + P = 17 640) against torch's fp32 GEMM (rocBLAS / hipBLASLt: the library call it replaces): time per launch, TFLOP/s of the
algorithmic 2 n k m, and the error of both against fp64."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
from pamnet_amd import ops
dev = torch.device('cuda:0')


def t_us(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def err(a, r):
    return float((a.double() - r).abs().max() / r.abs().max())


print('%-34s %10s %10s %10s %10s %10s' % ('shape [n, k] x [m, k]^T', 'own us', 'TFLOP/s', 'torch us', 'own err', 'torch err'))
for n, k, m in ((2286, 256, 256), (2286, 256, 512), (32888, 256, 512), (17640, 42, 256), (32888, 16, 256), (709656, 128, 256)):
    torch.manual_seed(0)
    x, w, b = torch.randn(n, k, device=dev), torch.randn(m, k, device=dev) / k ** 0.5, torch.randn(m, device=dev)
    g = torch.randn(n, m, device=dev)
    y = ops._dense_fwd(x, w, b, True, True)
    ref = torch.nn.functional.silu(torch.nn.functional.linear(x.double(), w.double(), b.double()))
    yt = torch.nn.functional.silu(torch.nn.functional.linear(x, w, b))
    fl = 2.0 * n * k * m
    u = t_us(lambda: ops._dense_fwd(x, w, b, True, True))
    ut = t_us(lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, w, b)))
    print('fwd  [%6d, %3d] x [%3d, %3d]^T     %10.1f %10.1f %10.1f %10.1e %10.1e' % (n, k, m, k, u, fl / u * 1e-6, ut, err(y[0], ref), err(yt, ref)))
    z = y[1]
    u = t_us(lambda: ops._dense_bwd(g, z, x, w, True, True, False, False))
    print('dx   (dZ W, SiLU\' while staging)     %10.1f %10.1f' % (u, fl / u * 1e-6))
    u = t_us(lambda: ops._dense_bwd(g, z, x, w, True, False, True, True))
    print('dW   (dZ^T X + db, split + reduce)   %10.1f %10.1f' % (u, fl / u * 1e-6))

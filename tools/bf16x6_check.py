#!/usr/bin/env python
"""numpy restatement of the bf16x6 arithmetic of csrc/gemm_core.h (no GPU needed): a float splits exactly into three bf16
pieces; the six piece products with i + j <= 2 reproduce an fp32 product to fp32 accuracy, three do not."""
import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + ((u >> 16) & 1) + 0x7fff) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def split(x):
    p0 = bf16_rne(x)
    r = x - p0
    p1 = bf16_rne(r)
    q = r - p1
    p2 = bf16_rne(q)
    return p0, p1, p2, q - p2


rng = np.random.default_rng(0)
K = 2304
a = (rng.standard_normal((K, 128)) * rng.lognormal(0, 2, (K, 1))).astype(np.float32)
b = rng.standard_normal((K, 128)).astype(np.float32)
a0, a1, a2, ar = split(a)
b0, b1, b2, br = split(b)
print('residual after three pieces: max |x - p0 - p1 - p2| =', float(np.abs(ar).max()), float(np.abs(br).max()))
mm = lambda x, y: x.astype(np.float64).T @ y.astype(np.float64)
ref = mm(a, b)
m = np.abs(ref).max()
x6 = mm(a2, b0) + mm(a1, b1) + mm(a0, b2) + mm(a1, b0) + mm(a0, b1) + mm(a0, b0)
x3 = mm(a1, b0) + mm(a0, b1) + mm(a0, b0)
acc = np.zeros((128, 128), np.float32)                       # fp32 accumulation per 32-row MFMA, small products first
for k in range(0, K, 32):
    s = slice(k, k + 32)
    for x, y in ((a2, b0), (a1, b1), (a0, b2), (a1, b0), (a0, b1), (a0, b0)):
        acc = (acc + mm(x[s], y[s]).astype(np.float32)).astype(np.float32)
print('max|err| / max|ref|:  fp32 matmul %.2e | six products, exact sum %.2e | six products, fp32 accumulation %.2e | '
      'three products %.2e' % (np.abs(a.T @ b - ref).max() / m, np.abs(x6 - ref).max() / m, np.abs(acc - ref).max() / m,
                               np.abs(x3 - ref).max() / m))

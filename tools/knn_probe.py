#!/usr/bin/env python
"""Where the kNN search (csrc/graph.hip knn_kernel, RNA B=8: 17.7 k queries x ~2 200 candidates, K = 50) spends its time:
private copies of graph.hip cut after the distance pass / the first bisection / the survivor compaction / complete."""
import ctypes, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd'))
import torch
from pamnet_amd import graph as G, lib, synth
CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')
lib.load()
dev = torch.device('cuda:0')
b = synth.rna_batch(2, 0, 8).to(dev)
pos = b.x[:, :3].contiguous()
n = pos.size(0)
nodeg = b.batch.to(torch.int32)
gptr, _ = G.csr_from_keys(nodeg, 8)
K = 50
nbr, dist = torch.empty(n * K, dtype=torch.int32, device=dev), torch.empty(n * K, device=dev)
P, I = ctypes.c_void_p, ctypes.c_int64
for cut, what in ((1, 'distance pass (2 212 candidates per query into registers, min / max)'), (2, '+ first bisection (down to <= 256 survivors)'),
                  (3, '+ compaction of the survivors'), (4, '+ second bisection (<= 64 survivors)'), (5, '+ second compaction'),
                  (0, 'complete (+ 64-lane sort, square roots, stores)')):
    so = '/tmp/libpamnet_knn%d.so' % cut
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] +
                          (['-DKNN_PHASE_CUT=%d' % cut] if cut else []) +
                          ['-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-ffp-contract=on', os.path.join(CSRC, 'graph.hip'), '-o', so])
    fn = ctypes.CDLL(so).pamnet_knn_i32
    fn.argtypes = [P, P, P, I, ctypes.c_int32, ctypes.c_float, P, P, P]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: fn(pos.data_ptr(), nodeg.data_ptr(), gptr.data_ptr(), n, K, 1e30, nbr.data_ptr(), dist.data_ptr(), st)
    for _ in range(20):
        assert call() == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        call()
    e.record(); e.synchronize()
    print('%6.1f us  %s' % (s.elapsed_time(e) / 50 * 1e3, what))

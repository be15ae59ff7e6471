"""exclusive scan launch time at the sizes graph construction uses (GPU box)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')): sys.path.insert(0, p)
import torch
from pamnet_amd import lib
dev = torch.device('cuda:0')
for n in (2286, 4900, 17700, 65536, 75760, 131072):
    x = torch.randint(0, 50, (n,), device=dev, dtype=torch.int32)
    out = torch.empty(n + 1, device=dev, dtype=torch.int32); tmp = torch.empty(n // 4096 + 8, device=dev, dtype=torch.int32)
    st = lib.stream_of(x)
    res = []
    for _once in (0,):                            # (the form is fixed per process: PAMNET_SMALL_FORMS=0 for the three-launch one)
        for _ in range(5):
            lib.call('pamnet_exclusive_scan_i32', lib.ptr(x), lib.ptr(out), n, lib.ptr(tmp), st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            lib.call('pamnet_exclusive_scan_i32', lib.ptr(x), lib.ptr(out), n, lib.ptr(tmp), st)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1e3 / 200)
    print('n = %6d: %.1f us  (%s)' % (n, res[0], 'three launches' if os.environ.get('PAMNET_SMALL_FORMS') == '0' else 'one workgroup up to 24576'))

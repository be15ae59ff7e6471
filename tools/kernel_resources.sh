#!/bin/bash
# Register / scratch / LDS use of the kernels of one csrc file (hipcc cross-compiles for gfx950: no GPU needed).
# usage: tools/kernel_resources.sh edge_agg.hip [name filter]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/physics-aware-multiplex-gnn_amd/csrc -ffp-contract=on \
  -c $R/physics-aware-multiplex-gnn_amd/csrc/$1 -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    if flt and flt not in name: continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print('%-90s vgpr %3s agpr %3s scratch %4s lds %6s occ %s' % (name[:90], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'LDS Size \[bytes/block\]'), g(r'Occupancy \[waves/SIMD\]')))
" "$2"
rm -f /tmp/kres_$$.o

# HBM traffic of the fused edge MLP -> node segment-sum kernel at the PDBbind B=32 shape (E_g ~ 700 k: 360 MB per
# [E_g, 128] tensor, beyond the 256 MB Infinity Cache -- at the QM9 batch everything is cache resident and the counters
# read almost nothing).  Separate --pmc passes with --kernel-trace only; FETCH_SIZE x2 / WRITE_SIZE as in pmc_scatter.sh
# (MI355X_MICROARCH.md, HBM section).  Writes gpurun_out/${PMC_OUT:-r06_edge_agg_pmc.json} (copy into profiles/ and commit).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmca_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmca_$c -- python $R/tools/agg_bench.py pdbbind > /tmp/pmca_$c.log 2>&1
done
python - <<'PY' > $R/gpurun_out/${PMC_OUT:-r06_edge_agg_pmc.json}
import csv, glob, json, re
D = 128
def counter(name, kernel):
    f = glob.glob('/tmp/pmca_%s/**/*counter_collection.csv' % name, recursive=True)[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if kernel in r['Kernel_Name'] and r['Counter_Name'] == name]
    return (sum(vals) / len(vals), len(vals)) if vals else (float('nan'), 0)
log = open('/tmp/pmca_FETCH_SIZE.log').read()
n, eg = (int(v) for v in re.search(r'N=(\d+) E_g=(\d+)', log).groups())
alg = 4.0 * D * eg + 8.0 * eg + 4.0 * D * n * 4        # e + indices + x1, P_i, P_j in, x2 out
out = {'shape': {'N': n, 'E_g': eg}, 'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- '
       'python tools/agg_bench.py pdbbind; traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction), KB -> bytes',
       'kernels': {}}
# template arguments: <MT, SAVE, ...>; the training instantiation also writes z and ea (2 x 4 d E_g bytes)
f = glob.glob('/tmp/pmca_FETCH_SIZE/**/*counter_collection.csv', recursive=True)[0]
names = sorted({r['Kernel_Name'] for r in csv.DictReader(open(f)) if 'global_edge_agg' in r['Kernel_Name']})
for k in names:
    fetch, n1 = counter('FETCH_SIZE', k)
    write, n2 = counter('WRITE_SIZE', k)
    traffic = (2.0 * fetch + write) * 1024.0
    bwd = 'bwd' in k
    targs = k.split('<')[1].split('>')[0].split(',')
    save = ('fwd_pp' in k and 'true' in targs[0]) or ('fwd' in k and len(targs) > 2 and 'true' in targs[2])   # pp: <SAVE>; chunked: <MTX, PRE, SAVE>
    a = alg + (8.0 * D * eg if save else 0.0)
    if bwd:       # reads d x2 (n), z, ea, writes dz, dea, d_e (+ read for accumulate), dP_i: 6 edge tensors + 2 node planes
        a = 4.0 * D * eg * 6 + 8.0 * eg + 4.0 * D * n * 2
    if 'bwd_wg' in k:   # round 5: z, ea, e in; dz out; d_e read-modify-write; d_agg in, dP_i zero-filled + written; 2 x 256 partial tiles
        a = 4.0 * D * eg * 6 + 8.0 * eg + 4.0 * D * n * 3 + 4.0 * (2 * 256 * (D * D + 2 * D) + 2048)
    short = re.search(r'global_edge_agg_\w+<[^>]*>', k).group(0)
    out['kernels'][short] = {'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write, 'launches': [n1, n2],
                              'traffic_bytes_per_launch': traffic, 'algorithmic_bytes_per_launch': a,
                              'traffic_over_algorithmic': traffic / a}
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/${PMC_OUT:-r06_edge_agg_pmc.json}

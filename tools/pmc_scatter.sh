# HBM traffic of the scatter-add roofline kernel (bench.py `roofline.traffic`): separate --pmc passes with --kernel-trace
# only (MI355X_MICROARCH.md HBM section: TCC_EA0_RDREQ-style sizes are unreliable on gfx950; FETCH_SIZE is in 32-byte
# units per 64-byte request -> x2 correction, WRITE_SIZE in 64-byte units as documented there).  Writes
# profiles/${PMC_OUT:-r05_scatter_add_pmc.json} (copy it from gpurun_out/ into profiles/ and commit).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/${PROBE:-tools/scatter_probe.py} > /tmp/pmc_$c.log 2>&1
done
python - <<'PY' > $R/gpurun_out/${PMC_OUT:-r05_scatter_add_pmc.json}
import csv, glob, json
def counter(name):
    f = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % name, recursive=True)[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if 'segment_sum' in r['Kernel_Name'] and r['Counter_Name'] == name]
    global kname
    kname = sorted({r['Kernel_Name'].split('::')[-1].split('<')[0].split('(')[0] for r in csv.DictReader(open(f)) if 'segment_sum' in r['Kernel_Name']})
    return sum(vals) / len(vals), len(vals)
fetch, n1 = counter('FETCH_SIZE')
write, n2 = counter('WRITE_SIZE')
# FETCH_SIZE / WRITE_SIZE are reported in kilobytes; on gfx950 FETCH_SIZE tallies wide coalesced 16 B/lane streaming reads at
# half their size (MI355X_MICROARCH.md, HBM section): traffic = 2 * FETCH + WRITE
import re
alg = float(re.search(r'algorithmic bytes (\d+)', open('/tmp/pmc_FETCH_SIZE.log').read()).group(1))
traffic = (2.0 * fetch + write) * 1024.0
print(json.dumps({'kernel': ' / '.join(kname), 'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write, 'launches': [n1, n2],
                  'traffic_bytes_per_launch': traffic, 'algorithmic_bytes_per_launch': alg,
                  'traffic_over_algorithmic': traffic / alg,
                  'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python ' + __import__('os').environ.get('PROBE', 'tools/scatter_probe.py') + '; traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)'}))
PY
cat $R/gpurun_out/${PMC_OUT:-r05_scatter_add_pmc.json}
# kernel-trace statistics of the same probe (average launch duration of the roofline kernel)
rm -rf /tmp/stat_scatter
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stat_scatter -- python $R/${PROBE:-tools/scatter_probe.py} > /tmp/stat_scatter.log 2>&1
python - <<'PY' > $R/gpurun_out/${STATS_OUT:-r05_scatter_add_kernel_stats.txt}
import csv, glob
f = glob.glob('/tmp/stat_scatter/**/*kernel_stats.csv', recursive=True)[0]
import os, re
# the shape label comes from the probe's own output (its segment lengths are drawn at random: the row count is not a constant)
m = re.search(r'rows_in (\d+) rows_out (\d+) algorithmic bytes (\d+)', open('/tmp/stat_scatter.log').read())
shape = ('[%s,128] -> [%s,128], %.3f GB algorithmic' % (m.group(1), m.group(2), int(m.group(3)) / 1e9)) if m else 'shape not reported by the probe'
print('rocprofv3 --kernel-trace --stats -- python %s   (%s%s)' % (os.environ.get('PROBE', 'tools/scatter_probe.py'), os.environ.get('PROBE_NOTE', ''), shape))
for r in csv.DictReader(open(f)):
    if 'segment_sum' in r['Name']:
        print('%s\n  calls %s  average %.1f us  min %.1f us  max %.1f us' % (r['Name'][:120], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
cat $R/gpurun_out/${STATS_OUT:-r05_scatter_add_kernel_stats.txt}

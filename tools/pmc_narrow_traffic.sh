# HBM traffic of the narrow row kernels from PMC counters (separate passes, --pmc with --kernel-trace only;
# FETCH_SIZE x2 correction for 16 B/lane coalesced reads on gfx950 as in profiles/r01_scatter_add_pmc.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_nt
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_nt -- python $R/tools/narrow_bench.py > /dev/null 2>&1
  f=$(find /tmp/pmc_nt -name '*counter_collection.csv' | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'\(.*', '', k)
    if k.startswith('void n') and 'reduce' not in k:
        acc[k].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    # launches per (d, m) config: 23 each, three m per d; report the largest-m group = last 23 of the d's 69
    print(sys.argv[2], k, 'max per launch (KB): %.0f' % max(v), 'n=%d' % len(v))
PY
done

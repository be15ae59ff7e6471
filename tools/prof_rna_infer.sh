cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/rna_infer.py 16 1 10
python $R/tools/rna_infer.py 64 2 10
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rna_infer -- python $R/tools/rna_infer.py 16 1 10 > $R/gpurun_out/rna_infer.log 2>&1
f=$(find $R/gpurun_out/rna_infer -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)/13e6
print('GPU ms/forward',tot)
for r in rows[:25]:
    print('  %-80s %4s %8.3f ms/fwd'%(r['Name'].replace('(anonymous namespace)::','')[:80],r['Calls'],int(r['TotalDurationNs'])/13e6))
PY

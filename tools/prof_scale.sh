cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in ${CONFIGS:-rna rna_d64}; do
  rm -rf $R/gpurun_out/scale_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scale_$c -- python $R/tools/scale_smoke.py $c > $R/gpurun_out/scale_$c.log 2>&1
  f=$(find $R/gpurun_out/scale_$c -name '*kernel_stats.csv' | head -1)
  echo "== $c"; grep ms/step $R/gpurun_out/scale_$c.log
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)/25e6
print('GPU ms/step',round(tot,3))
for r in rows[:34]:
    print('  %-90s %4s %8.3f ms/step'%(r['Name'].replace('(anonymous namespace)::','')[:90],r['Calls'],int(r['TotalDurationNs'])/25e6))
PY
done

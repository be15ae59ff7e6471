cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in pdbbind rna rna_d64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scale_$c -- python $R/tools/scale_smoke.py $c > $R/gpurun_out/scale_$c.log 2>&1
  f=$(find $R/gpurun_out/scale_$c -name '*kernel_stats.csv' | head -1)
  echo "== $c"; tail -1 $R/gpurun_out/scale_$c.log; head -22 $f | cut -c1-150
done

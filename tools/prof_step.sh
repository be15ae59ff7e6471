cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-rooflines --no-other-configs > $R/gpurun_out/prof_step.log 2>&1
f=$(find $R/gpurun_out/prof_step -name '*kernel_trace.csv' | head -1)
python $R/tools/step_profile.py $f 30 > $R/gpurun_out/step_budget.txt
head -34 $R/gpurun_out/step_budget.txt | cut -c1-110

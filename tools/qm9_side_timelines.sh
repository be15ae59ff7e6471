cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf /tmp/p_q /tmp/p_b
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_q -- python $R/tools/store_steps.py qm9 60 > /tmp/p_q.log 2>&1
f=$(find /tmp/p_q -name '*kernel_trace.csv' | head -1)
python $R/tools/step_timeline.py $f 30 | head -40 > $O/qm9_store_timeline.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_b -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-rooflines --no-other-configs > /tmp/p_b.log 2>&1
f=$(find /tmp/p_b -name '*kernel_trace.csv' | head -1)
python $R/tools/step_timeline.py $f 20 | head -40 > $O/qm9_plain_timeline.txt

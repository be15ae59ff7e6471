cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf /tmp/p_s
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_s -- python $R/tools/store_steps.py rna 60 serial > /tmp/p_s.log 2>&1
f=$(find /tmp/p_s -name '*kernel_trace.csv' | head -1)
(grep ms/step /tmp/p_s.log; python $R/tools/step_profile.py $f 80) > $O/rna_serial_budget.txt
python $R/tools/step_timeline.py $f 30 > $O/rna_serial_timeline.txt

# Kernel budget of a training step through the store with the graph built in line on the main stream (no cross-stream overlap:
# every duration is the kernel's own).  usage: serial_budget.sh [qm9|rna|pdbbind]  -> gpurun_out/<kind>_serial_budget.txt
cd /tmp && export TMPDIR=/tmp
K=${1:-rna}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf /tmp/p_s
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_s -- python $R/tools/store_steps.py $K 60 serial > /tmp/p_s.log 2>&1
f=$(find /tmp/p_s -name '*kernel_trace.csv' | head -1)
(grep ms/step /tmp/p_s.log; python $R/tools/step_profile.py $f 90) > $O/${K}_serial_budget.txt
python $R/tools/step_timeline.py $f 30 > $O/${K}_serial_timeline.txt

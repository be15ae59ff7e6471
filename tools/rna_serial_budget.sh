#!/bin/bash
# Kernel budget of the RNA step with graph construction IN LINE on the main stream (no overlap: kernel durations are their own):
# rocprofv3 kernel trace of `tools/store_steps.py rna 60 serial`, summarised per step -> gpurun_out/rna_serial_budget.txt
# (committed as profiles/r05_rna_serial_kernel_budget.txt).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_rna
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_rna -- python $R/tools/store_steps.py rna 60 serial > /tmp/p_rna.log 2>&1
f=$(find /tmp/p_rna -name '*kernel_trace.csv' | head -1)
mkdir -p $R/gpurun_out
(grep ms/step /tmp/p_rna.log; python $R/tools/step_profile.py $f 60) > $R/gpurun_out/rna_serial_budget.txt

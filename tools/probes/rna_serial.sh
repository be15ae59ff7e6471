# serial RNA kernel budget under rocprofv3 -> gpurun_out/rna_serial_budget.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_rna; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_rna -- python $GRAFT_REPO_ROOT/tools/store_steps.py rna 60 serial > /tmp/p_rna.log 2>&1
f=$(find /tmp/p_rna -name '*kernel_trace.csv' | head -1)
(grep ms/step /tmp/p_rna.log; python $GRAFT_REPO_ROOT/tools/step_profile.py $f 60) > $GRAFT_REPO_ROOT/gpurun_out/rna_serial_budget.txt
cd $GRAFT_REPO_ROOT

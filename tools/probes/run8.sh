cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_budget.txt $O/r05_step_budget.txt
python $R/tools/step_timeline.py $(ls $O/prof_step/*/*_kernel_trace.csv | head -1) > $O/r05_step_timeline.txt
rm -rf /tmp/p_pdb
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_pdb -- python $R/tools/store_steps.py pdbbind 60 > /tmp/p_pdb.log 2>&1
f=$(find /tmp/p_pdb -name '*kernel_trace.csv' | head -1)
(grep ms/step /tmp/p_pdb.log; python $R/tools/step_profile.py $f 60) > $O/r05_pdbbind_step_budget.txt
rm -rf $O/prof_step
head -40 $O/r05_step_budget.txt | cut -c1-118; head -24 $O/r05_pdbbind_step_budget.txt | cut -c1-118

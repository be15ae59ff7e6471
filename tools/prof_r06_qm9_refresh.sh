cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_budget.txt $O/r06_step_budget.txt
python $R/tools/speed_of_light.py $O/r06_step_budget.txt > $O/r06_speed_of_light.txt
cp $(ls $O/prof_step/*/*_kernel_stats.csv | head -1) $O/r06_kernel_stats.csv
python $R/tools/step_timeline.py $(ls $O/prof_step/*/*_kernel_trace.csv | head -1) > $O/r06_step_timeline.txt
cd $R && python bench.py 2>/dev/null | tail -1 > $O/r06_bench_line.json
cd $R && python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r06_gpu_suite.txt
tail -3 $O/r06_speed_of_light.txt; cat $O/r06_gpu_suite.txt; head -c 600 $O/r06_bench_line.json

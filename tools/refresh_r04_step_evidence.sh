cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_budget.txt $O/r04_step_budget.txt
python $R/tools/speed_of_light.py $O/r04_step_budget.txt > $O/r04_speed_of_light.txt
cp $(ls $O/prof_step/*/*_kernel_stats.csv | head -1) $O/r04_kernel_stats.csv
python $R/tools/step_timeline.py $(ls $O/prof_step/*/*_kernel_trace.csv | head -1) > $O/r04_step_timeline.txt
for k in pdbbind; do
  rm -rf /tmp/p_$k
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$k -- python $R/tools/store_steps.py $k 60 > /tmp/p_$k.log 2>&1
  f=$(find /tmp/p_$k -name '*kernel_trace.csv' | head -1)
  (grep ms/step /tmp/p_$k.log; python $R/tools/step_profile.py $f 60) > $O/r04_${k}_step_budget.txt
  python $R/tools/step_timeline.py $f 30 > $O/r04_${k}_step_timeline.txt
done
cd $R && python bench.py 2>/dev/null | tail -1 > $O/r04_bench_line.json
rm -rf $O/prof_step
head -12 $O/r04_step_budget.txt; head -8 $O/r04_pdbbind_step_budget.txt
python -c "
import json; d=json.load(open('$O/r04_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:(v.get('train_ms_per_step'), v.get('store_train_ms_per_step')) for k,v in d['other_configs'].items()}); print([ (k['kernel'][:40], k.get('us_per_launch')) for k in d['step_kernels']])"

# Round-5 evidence bundle (run on the GPU box; copies land in gpurun_out/r05_*, to be committed under profiles/):
#   kernel stats + per-step budget + main-queue timeline of the bench command, MFMA-pipe utilisation (PMC pass), scatter-add
#   HBM traffic (PMC passes) + its kernel statistics, fused edge-kernel traffic at the PDBbind shape (PMC passes) + their
#   micro-benchmark, host-phase profiles, kernel budgets of the RNA / PDBbind steps through the store, forward splits, the
#   parity figures the tests print, the bench line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_budget.txt $O/r05_step_budget.txt
python $R/tools/speed_of_light.py $O/r05_step_budget.txt > $O/r05_speed_of_light.txt
cp $(ls $O/prof_step/*/*_kernel_stats.csv | head -1) $O/r05_kernel_stats.csv
python $R/tools/step_timeline.py $(ls $O/prof_step/*/*_kernel_trace.csv | head -1) > $O/r05_step_timeline.txt
bash $R/tools/pmc_mfma.sh > /dev/null 2>&1
cp $O/mfma_util.txt $O/r05_mfma_util_pmc.txt
PMC_OUT=r05_scatter_add_pmc.json STATS_OUT=r05_scatter_add_kernel_stats.txt bash $R/tools/pmc_scatter.sh > /dev/null 2>&1
PROBE=tools/perm_probe.py PROBE_NOTE='transposed-CSR gather form at the PDBbind B=32 shape' PMC_OUT=r05_perm_segment_sum_pmc.json STATS_OUT=r05_perm_segment_sum_kernel_stats.txt bash $R/tools/pmc_scatter.sh > /dev/null 2>&1
PMC_OUT=r05_edge_agg_pmc.json bash $R/tools/pmc_edge_agg.sh > /dev/null 2>&1
(python $R/tools/agg_bench.py qm9 2>/dev/null; python $R/tools/agg_bench.py pdbbind 2>/dev/null) | grep -v amdgpu.ids > $O/r05_edge_agg_microbench.txt
for k in rna qm9 pdbbind; do python $R/tools/host_phases.py $k 2>/dev/null | grep -v amdgpu.ids > $O/r05_host_phases_$k.txt; done
for k in rna pdbbind; do
  rm -rf /tmp/p_$k
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$k -- python $R/tools/store_steps.py $k 60 > /tmp/p_$k.log 2>&1
  f=$(find /tmp/p_$k -name '*kernel_trace.csv' | head -1)
  (grep ms/step /tmp/p_$k.log; python $R/tools/step_profile.py $f 60) > $O/r05_${k}_step_budget.txt
  python $R/tools/step_timeline.py $f 30 > $O/r05_${k}_step_timeline.txt
done
(python $R/tools/fwd_store_pipe.py qm9 2>/dev/null; python $R/tools/fwd_store_pipe.py rna 2>/dev/null; python $R/tools/store_steps.py qm9 300 2>/dev/null; python $R/tools/store_steps.py rna 200 2>/dev/null; python $R/tools/store_steps.py pdbbind 60 2>/dev/null) | grep -v amdgpu.ids > $O/r05_forward_and_store_steps.txt
cd $R && python -m pytest tests/test_hip_model.py tests/test_store.py -m gpu -q -s -k "baseline or trainer_step_path or large_batch or configs1" 2>/dev/null | grep -E "vs the reference|vs oracle|12 targets|Trainer.forward_backward|through the store|passed|failed" > $O/r05_parity_figures.txt
cd $R && python bench.py 2>/dev/null | tail -1 > $O/r05_bench_line.json
ls -la $O | grep r05_
# round 5: the fused global-edge backward + weight gradients against what it replaces, its phase timestamps, same-box A/B of the
# PDBbind step with the old route (PAMNET_EDGE_WGRAD=0), the upper bound of "edge-embedding pieces once per step" (split probe)
(python $R/tools/edge_wgrad_probe.py pdbbind 32 2>/dev/null; python $R/tools/edge_wgrad_probe.py qm9 128 2>/dev/null) | grep -v amdgpu.ids > $O/r05_edge_wgrad_probe.txt
python $R/tools/edge_wgrad_phase_probe.py pdbbind 2>/dev/null | grep -v amdgpu.ids > $O/r05_edge_wgrad_phase_probe.txt
(for i in 1 2 3; do for v in 0 1; do PAMNET_EDGE_WGRAD=$v python $R/tools/pdbbind_steps.py 60 2>/dev/null | tail -1; done; done; for v in 0 1; do PAMNET_EDGE_WGRAD=$v python $R/tools/store_steps.py qm9 300 2>/dev/null | tail -1; done) > $O/r05_edge_wgrad_step_ab.txt
(echo "global_edge_agg_fwd / bwd kernels, exact three-piece split of the staged rows (production):"; python $R/tools/agg_probe.py pdbbind 2>/dev/null | grep "kernel .* us"; python $R/tools/agg_probe.py 2>/dev/null | grep "kernel .* us"; echo "the same with ONE conversion per element instead of the split (-DPAMNET_SPLIT_PROBE: wrong numbers, the instruction count of ready-made pieces):"; PAMNET_PROBE_FLAGS=-DPAMNET_SPLIT_PROBE python $R/tools/agg_probe.py pdbbind 2>/dev/null | grep "kernel .* us"; PAMNET_PROBE_FLAGS=-DPAMNET_SPLIT_PROBE python $R/tools/agg_probe.py 2>/dev/null | grep "kernel .* us") > $O/r05_split_probe.txt
# issue-slot counters per kernel (two SQ passes each) and the forward chain's phase timestamps (production form)
KIND=pdbbind bash $R/tools/pmc_issue.sh > /dev/null 2>&1; cp $O/issue_pdbbind.txt $O/r05_issue_slots_pdbbind_pmc.txt
KIND=qm9 STEPS=30 bash $R/tools/pmc_issue.sh > /dev/null 2>&1; cp $O/issue_qm9.txt $O/r05_issue_slots_qm9_pmc.txt
# the GPU suite as the driver runs it
cd $R && python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r05_gpu_suite.txt
ls -la $O | grep r05_

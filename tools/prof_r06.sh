# Round-6 evidence bundle (run on the GPU box; copies land in gpurun_out/r06_*, to be committed under profiles/):
#   kernel stats + per-step budget + main-queue timeline of the bench command, MFMA-pipe utilisation (PMC pass), scatter-add HBM
#   traffic (PMC passes) + its kernel statistics, fused edge-kernel traffic at the PDBbind shape (PMC passes; the ping-pong forward
#   included), issue-slot counters, kernel budgets of the RNA / PDBbind steps, the ping-pong forward's probes (chunked against
#   ping-pong, phase stamps with and without the partner's GEMMs), the bench line, the GPU suite.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_budget.txt $O/r06_step_budget.txt
python $R/tools/speed_of_light.py $O/r06_step_budget.txt > $O/r06_speed_of_light.txt
cp $(ls $O/prof_step/*/*_kernel_stats.csv | head -1) $O/r06_kernel_stats.csv
python $R/tools/step_timeline.py $(ls $O/prof_step/*/*_kernel_trace.csv | head -1) > $O/r06_step_timeline.txt
bash $R/tools/pmc_mfma.sh > /dev/null 2>&1
cp $O/mfma_util.txt $O/r06_mfma_util_pmc.txt
PMC_OUT=r06_scatter_add_pmc.json STATS_OUT=r06_scatter_add_kernel_stats.txt bash $R/tools/pmc_scatter.sh > /dev/null 2>&1
PROBE=tools/perm_probe.py PROBE_NOTE='transposed-CSR gather form at the PDBbind B=32 shape' PMC_OUT=r06_perm_segment_sum_pmc.json STATS_OUT=r06_perm_segment_sum_kernel_stats.txt bash $R/tools/pmc_scatter.sh > /dev/null 2>&1
PMC_OUT=r06_edge_agg_pmc.json bash $R/tools/pmc_edge_agg.sh > /dev/null 2>&1
(python $R/tools/agg_bench.py qm9 2>/dev/null; python $R/tools/agg_bench.py pdbbind 2>/dev/null) | grep -v amdgpu.ids > $O/r06_edge_agg_microbench.txt
# the ping-pong forward: against the chunked form (bitwise check + timing, alternated), then where its waves spend an iteration
(python $R/tools/pp_probe.py pdbbind 2>/dev/null; python $R/tools/pp_probe.py qm9 2>/dev/null) | grep -v amdgpu.ids > $O/r06_pp_probe.txt
(echo "== production form"; python $R/tools/pp_phase_probe.py pdbbind 2>/dev/null
 echo "== -DPP_NO_GEMM: the vector work / the walker without the partner's GEMMs on their SIMD"; PAMNET_PROBE_FLAGS=-DPP_NO_GEMM python $R/tools/pp_phase_probe.py pdbbind 2>/dev/null
 echo "== -DPP_NO_VEC: the GEMM phases without the partner's vector work"; PAMNET_PROBE_FLAGS=-DPP_NO_VEC python $R/tools/pp_phase_probe.py pdbbind 2>/dev/null
 echo "== -DPP_NO_SILU: the epilogue without its transcendentals"; PAMNET_PROBE_FLAGS=-DPP_NO_SILU python $R/tools/pp_phase_probe.py pdbbind 2>/dev/null
 echo "== -DPP_VEC_PRIO=3: s_setprio 3 around the vector work") | grep -v amdgpu.ids > $O/r06_pp_phase_probe.txt
PAMNET_PROBE_FLAGS=-DPP_VEC_PRIO=3 python $R/tools/pp_phase_probe.py pdbbind 2>/dev/null | grep -v amdgpu.ids >> $O/r06_pp_phase_probe.txt
for k in rna qm9 pdbbind; do python $R/tools/host_phases.py $k 2>/dev/null | grep -v amdgpu.ids > $O/r06_host_phases_$k.txt; done
for k in rna pdbbind; do
  rm -rf /tmp/p_$k
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$k -- python $R/tools/store_steps.py $k 60 > /tmp/p_$k.log 2>&1
  f=$(find /tmp/p_$k -name '*kernel_trace.csv' | head -1)
  (grep ms/step /tmp/p_$k.log; python $R/tools/step_profile.py $f 60) > $O/r06_${k}_step_budget.txt
  python $R/tools/step_timeline.py $f 30 > $O/r06_${k}_step_timeline.txt
done
(python $R/tools/fwd_store_pipe.py qm9 2>/dev/null; python $R/tools/fwd_store_pipe.py rna 2>/dev/null; python $R/tools/store_steps.py qm9 300 2>/dev/null; python $R/tools/store_steps.py rna 200 2>/dev/null; python $R/tools/store_steps.py pdbbind 60 2>/dev/null) | grep -v amdgpu.ids > $O/r06_forward_and_store_steps.txt
cd $R && python -m pytest tests/test_hip_model.py tests/test_store.py -m gpu -q -s -k "baseline or trainer_step_path or large_batch or configs1" 2>/dev/null | grep -E "vs the reference|vs oracle|12 targets|Trainer.forward_backward|through the store|passed|failed" > $O/r06_parity_figures.txt
cd $R && python bench.py 2>/dev/null | tail -1 > $O/r06_bench_line.json
KIND=pdbbind bash $R/tools/pmc_issue.sh > /dev/null 2>&1; cp $O/issue_pdbbind.txt $O/r06_issue_slots_pdbbind_pmc.txt
KIND=qm9 STEPS=30 bash $R/tools/pmc_issue.sh > /dev/null 2>&1; cp $O/issue_qm9.txt $O/r06_issue_slots_qm9_pmc.txt
cd $R && python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r06_gpu_suite.txt
ls -la $O | grep r06_

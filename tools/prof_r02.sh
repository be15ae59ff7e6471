# Round-2 evidence bundle (run on the GPU box; copies land in gpurun_out/r02_*, to be committed under profiles/):
#   kernel stats + per-step budget + main-queue timeline of the bench command, MFMA-pipe utilisation (PMC pass),
#   scatter-add HBM traffic (PMC passes), forward-only host split, fused-kernel phase probes, PDBbind-shape micro-benchmark.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_step.sh > /dev/null 2>&1
cp $O/step_budget.txt $O/r02_step_budget.txt
cp $(ls $O/prof_step/*/*_kernel_stats.csv | head -1) $O/r02_kernel_stats.csv
python $R/tools/step_timeline.py $(ls $O/prof_step/*/*_kernel_trace.csv | head -1) > $O/r02_step_timeline.txt
bash $R/tools/pmc_mfma.sh > /dev/null 2>&1
cp $O/mfma_util.txt $O/r02_mfma_util_pmc.txt
bash $R/tools/pmc_scatter.sh > /dev/null 2>&1
bash $R/tools/pmc_edge_agg.sh > /dev/null 2>&1
python $R/tools/qm9_host_split.py 2>/dev/null | tail -1 > $O/r02_qm9_host_split.txt
python $R/tools/agg_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/r02_edge_agg_phase_probe.txt
(python $R/tools/agg_bench.py qm9 2>/dev/null; python $R/tools/agg_bench.py pdbbind 2>/dev/null) | grep -v amdgpu.ids > $O/r02_edge_agg_microbench.txt
(python $R/tools/scale_smoke.py 2>/dev/null; python $R/tools/scale_pipelined.py 2>/dev/null; python $R/tools/rna_infer.py 16 1 100 2>/dev/null; python $R/tools/rna_infer.py 64 2 60 2>/dev/null) | grep -v amdgpu.ids > $O/r02_other_configs.txt
# forward-only kernel trace: no segment_sum launch in the forward of either layer kind, 5 launches per layer pair
rm -rf $O/prof_fwd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -- python $R/tools/fwd_only.py > /dev/null 2>&1
cp $(ls $O/prof_fwd/*/*_kernel_stats.csv | head -1) $O/r02_forward_only_kernel_stats.csv
python $R/tools/scale_store.py 2>/dev/null | grep -v amdgpu.ids > $O/r02_store_other_configs.txt
(python $R/tools/noprefetch_bound.py 2>/dev/null; python $R/tools/side_launch_probe.py 2>/dev/null) | grep -v amdgpu.ids > $O/r02_side_stream_interference.txt
CONFIGS="rna rna_d64" bash $R/tools/prof_scale.sh > $O/r02_rna_kernel_table.txt 2>&1
cd $R && python bench.py 2>/dev/null | tail -1 > $O/r02_bench_line.json
ls -la $O | grep r02_

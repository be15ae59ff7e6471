python -m pytest tests/test_hip_fused.py -x -q -m gpu -k "input_stage or embed" 2>&1 | tail -2
python -m pytest tests/test_hip_model.py tests/test_store.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for v in 0 65536; do echo "== PAMNET_EMBED_WAVE_ROWS=$v"; PAMNET_EMBED_WAVE_ROWS=$v python tools/store_steps.py pdbbind 60 2>&1 | tail -1; done; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_pdb; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_pdb -- python $GRAFT_REPO_ROOT/tools/store_steps.py pdbbind 40 > /tmp/p_pdb.log 2>&1
f=$(find /tmp/p_pdb -name '*kernel_trace.csv' | head -1)
(grep ms/step /tmp/p_pdb.log; python $GRAFT_REPO_ROOT/tools/step_profile.py $f 40) > $GRAFT_REPO_ROOT/gpurun_out/pdb_budget.txt

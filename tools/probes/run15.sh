python -m pytest tests/test_hip_kernels.py tests/test_graph_engine.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do for v in 0 1; do echo "== PAMNET_HIST_LDS=$v"; PAMNET_HIST_LDS=$v python tools/store_steps.py rna 200 2>&1 | tail -1; done; done

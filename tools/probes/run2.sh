python tools/edge_wgrad_phase_probe.py pdbbind 2>&1 | tail -32
for i in 1 2; do for v in 0 1; do echo "== PAMNET_EDGE_WGRAD=$v"; PAMNET_EDGE_WGRAD=$v python tools/store_steps.py pdbbind 100 2>&1 | tail -1; PAMNET_EDGE_WGRAD=$v python tools/store_steps.py qm9 300 2>&1 | tail -1; done; done
PAMNET_EDGE_WGRAD=1 python -m pytest tests/test_hip_model.py tests/test_hip_fused.py tests/test_store.py -x -q 2>&1 | tail -5

python tools/edge_wgrad_phase_probe.py pdbbind 2>&1 | tail -28
python -m pytest tests/test_hip_edge_agg.py -x -q -k "weight_gradients" 2>&1 | tail -3
python tools/edge_wgrad_probe.py pdbbind 32 2>&1 | tail -6

set -x
python -m pytest tests/test_hip_edge_agg.py -x -q -k "weight_gradients" 2>&1 | tail -15
python tools/edge_wgrad_probe.py pdbbind 32 2>&1 | tail -8
python tools/edge_wgrad_probe.py qm9 128 2>&1 | tail -8

PAMNET_EDGE_WGRAD=1 python -m pytest tests/test_hip_model.py tests/test_hip_fused.py tests/test_store.py tests/test_hip_edge_agg.py tests/test_train_golden.py -x -q 2>&1 | tail -3
python -m pytest tests/test_hip_model.py tests/test_train_golden.py -x -q -k "pdbbind" 2>&1 | tail -3

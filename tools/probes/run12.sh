python -m pytest tests/test_hip_edge_agg.py tests/test_graph_engine.py tests/test_store.py tests/test_hip_model.py -x -q 2>&1 | tail -2
for i in 1 2 3; do for v in 0 1; do echo -n "PAMNET_TT_AUX=$v "; PAMNET_TT_AUX=$v python tools/store_steps.py qm9 400 2>&1 | tail -1; done; done
for i in 1 2; do for v in 0 1; do echo -n "PAMNET_TT_AUX=$v "; PAMNET_TT_AUX=$v python tools/pdbbind_steps.py 60 2>&1 | tail -1; done; done
python bench.py --steps 300 --no-other-configs --no-cpu-baseline --no-rooflines 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench plain path', d['value'], d['ms_per_step'])"

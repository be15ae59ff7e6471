python -m pytest tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for v in 1 ""; do echo "== recheck=${v:-0}"; PAMNET_TMP_RECHECK=$v python tools/store_steps.py rna 300 2>&1 | tail -1; done; done
for v in 1 ""; do echo "== recheck=${v:-0}"; PAMNET_TMP_RECHECK=$v python tools/store_steps.py qm9 300 2>&1 | tail -1; done

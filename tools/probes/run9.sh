python -m pytest tests/test_hip_kernels.py -x -q -k "embed" 2>&1 | tail -2
python tools/pdbbind_steps.py 60 2>&1 | tail -1
python tools/store_steps.py pdbbind 60 2>&1 | tail -1
python tools/store_steps.py qm9 300 2>&1 | tail -1
python tools/store_steps.py rna 200 2>&1 | tail -1

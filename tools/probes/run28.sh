python -m pytest tests/test_hip_narrow.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for l in libK4 ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/tools/probes/$l.so} python tools/store_steps.py rna 200 2>&1 | tail -1; done; done
bash tools/probes/rna_serial.sh

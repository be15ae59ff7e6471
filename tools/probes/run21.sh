python -m pytest tests/test_hip_fused.py tests/test_hip_kernels.py tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for l in libV1 ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/tools/probes/$l.so} python tools/store_steps.py qm9 300 2>&1 | tail -1; done; done
for l in libV1 ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/tools/probes/$l.so} python tools/store_steps.py pdbbind 60 2>&1 | tail -1; done

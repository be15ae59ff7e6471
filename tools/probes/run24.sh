python -m pytest tests/test_hip_kernels.py tests/test_graph_engine.py -x -q -m gpu -k "knn or graph or rna" 2>&1 | tail -2
python tools/knn_probe.py 2>&1 | grep -v amdgpu.ids
for i in 1 2 3; do for l in libK1 ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/tools/probes/$l.so} python tools/store_steps.py rna 200 2>&1 | tail -1; done; done

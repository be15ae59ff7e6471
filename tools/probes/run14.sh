python -m pytest tests/test_hip_narrow.py tests/test_hip_kernels.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/store_steps.py rna 200 2>&1 | tail -1; done; done
for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/narrow_embed_bench.py 2>&1 | tail -3; done

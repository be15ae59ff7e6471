python -m pytest tests/test_hip_edge_agg.py tests/test_hip_fused.py -x -q -m gpu 2>&1 | tail -2
PAMNET_PROBE_FLAGS=-DPAMNET_PROBE_CHUNK=10 python tools/agg_probe.py pdbbind 2>&1 | grep -v amdgpu.ids | head -12
for i in 1 2; do for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/agg_bench.py pdbbind 2>&1 | grep -i "fwd\|forward"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/agg_bench.py qm9 2>&1 | grep -i "fwd\|forward"; done; done
for i in 1 2; do for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/store_steps.py pdbbind 60 2>&1 | tail -1; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/store_steps.py qm9 300 2>&1 | tail -1; done; done

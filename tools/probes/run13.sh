python -m pytest tests/test_hip_edge_agg.py -x -q 2>&1 | tail -1
for i in 1 2 3; do for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/agg_bench.py qm9 2>&1 | grep "global.*fused"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/store_steps.py qm9 400 2>&1 | tail -1; done; done
for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/agg_bench.py pdbbind 2>&1 | grep "global.*fused"; done

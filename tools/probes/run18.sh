python -m pytest tests/test_hip_edge_agg.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for l in tools/probes/libpamnet_before.so ""; do echo "== lib: ${l:-current}"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/agg_bench.py qm9 2>&1 | grep -i "bwd\|backward"; PAMNET_HIP_LIB=${l:+$PWD/$l} python tools/store_steps.py qm9 300 2>&1 | tail -1; done; done

for i in 1 2 3; do for l in libpamnet_before libHEAD libV1 libV2; do echo "== lib: $l"; PAMNET_HIP_LIB=$PWD/tools/probes/$l.so python tools/store_steps.py qm9 300 2>&1 | tail -1; done; done
for l in libHEAD libV1 libV2; do echo "== lib: $l"; PAMNET_HIP_LIB=$PWD/tools/probes/$l.so python tools/agg_bench.py qm9 2>&1 | grep -i "fused"; done

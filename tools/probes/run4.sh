python tools/embed_bwd_probe.py 2>&1 | tail -8
python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -x -q -k "embed or golden or baseline_qm9" 2>&1 | tail -3
for i in 1 2; do python tools/store_steps.py qm9 400 2>&1 | tail -1; done
python bench.py --steps 200 > gpurun_out/r05_bench_mid.json 2> gpurun_out/r05_bench_mid.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_mid.json'))
print('QM9', d['value'], d['ms_per_step'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict): print(k, v.get('train_ms_per_step'), v.get('store_train_ms_per_step'))
PY
echo "=== split probe (fwd kernel, pdbbind / qm9): exact split, then one conversion"
for f in "" "-DPAMNET_SPLIT_PROBE"; do PAMNET_PROBE_FLAGS="$f" python tools/agg_probe.py pdbbind 2>&1 | grep "kernel .* us"; PAMNET_PROBE_FLAGS="$f" python tools/agg_probe.py 2>&1 | grep "kernel .* us"; done

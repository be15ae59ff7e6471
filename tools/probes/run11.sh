for i in 1 2; do for v in 0 1; do echo -n "PAMNET_CHAIN_BF16=$v "; PAMNET_CHAIN_BF16=$v python tools/pdbbind_steps.py 60 2>&1 | tail -1; done; done
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_gpu_suite.txt

for i in 1 2; do for v in 0 1; do PAMNET_EDGE_WGRAD=$v python tools/pdbbind_steps.py 60 2>&1 | tail -1; done; python tools/pdbbind_steps.py 60 2>&1 | tail -1; done
for v in 0 1; do PAMNET_EDGE_WGRAD=$v python tools/store_steps.py pdbbind 60 2>&1 | tail -1; done
python tools/store_steps.py pdbbind 60 2>&1 | tail -1

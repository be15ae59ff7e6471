for i in 1 2 3; do for v in 0 1; do echo -n "PAMNET_SEG_WIDE=$v "; PAMNET_SEG_WIDE=$v python tools/perm_probe.py 2>&1 | grep "us per launch" | cut -c1-40; done; done

for i in 1 2 3; do for v in 0 8 16; do PAMNET_SEG_FLAT=$v python tools/seg_flat_ab.py 2>&1 | tail -1; done; done
PAMNET_SEG_FLAT=8 python -m pytest tests/test_hip_kernels.py -x -q -k "segment" 2>&1 | tail -2

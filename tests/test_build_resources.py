"""Build-time resource check of the weight-gradient kernels (no GPU needed: hipcc cross-compiles for gfx950).

Their register allocation sits at the 256-register edge that lets a second workgroup share the CU (csrc/wgrad.hip); a
small change to the body tips the compiler into spilling, which doubled the launch time once without failing any parity
test.  This test compiles csrc/wgrad.hip with -Rpass-analysis=kernel-resource-usage and requires zero scratch."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd', 'csrc')


def test_wgrad_kernels_do_not_spill(tmp_path):
    hipcc = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else shutil.which('hipcc')
    if not hipcc:
        pytest.skip('hipcc not available')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(REPO, 'include'),
                        '-I' + CSRC, '-ffp-contract=on', '-c', os.path.join(CSRC, 'wgrad.hip'), '-o',
                        str(tmp_path / 'wgrad.o'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r'remark: Function Name: ', r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if 'wgrad_kernel' in name or 'wgrad_fused' in name:
            seen += 1
            scratch = int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b).group(1))
            vgprs = int(re.search(r'VGPRs: (\d+)', b).group(1))
            agprs = int(re.search(r'AGPRs: (\d+)', b).group(1))
            assert scratch == 0, '%s spills (%d bytes of scratch per lane)' % (name, scratch)
            assert vgprs + agprs <= 256, '%s needs %d registers: no second workgroup per CU' % (name, vgprs + agprs)
    assert seen >= 3


def test_dense_gemm_kernel_resources(tmp_path):
    """csrc/dense.hip: no scratch, and at most 128 registers -- four workgroups per CU (one wave each per SIMD) overlap one
    another's staging, which is what hides the two barriers per 32-index block."""
    hipcc = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else shutil.which('hipcc')
    if not hipcc:
        pytest.skip('hipcc not available')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(REPO, 'include'),
                        '-I' + CSRC, '-ffp-contract=on', '-c', os.path.join(CSRC, 'dense.hip'), '-o',
                        str(tmp_path / 'dense.o'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = [b for b in re.split(r'remark: Function Name: ', r.stderr)[1:] if 'dense_gemm_kernel' in b.split()[0]]
    assert len(blocks) == 1
    b = blocks[0]
    assert int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b).group(1)) == 0
    assert int(re.search(r'VGPRs: (\d+)', b).group(1)) + int(re.search(r'AGPRs: (\d+)', b).group(1)) <= 128
    assert int(re.search(r'LDS Size \[bytes/block\]: (\d+)', b).group(1)) <= 40 * 1024


def test_fused_edge_backward_with_weight_gradients_resources(tmp_path):
    """csrc/edge_agg.hip global_edge_agg_bwd_wg_kernel (round 5): one 8-wave workgroup per CU -- at most 256 registers, its LDS
    (three piece images + the d e tile + the third weight pieces) inside the 160 KB of a CU, and no more than the 12 bytes of
    scratch the compiler parks OUTSIDE the chunk loop (a spill inside the loop is a vector-memory request that retires in order
    with the loop's loads: the waits there stop being exact -- it cost 15 % when it happened during development)."""
    hipcc = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else shutil.which('hipcc')
    if not hipcc:
        pytest.skip('hipcc not available')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(REPO, 'include'),
                        '-I' + CSRC, '-ffp-contract=on', '-c', os.path.join(CSRC, 'edge_agg.hip'), '-o',
                        str(tmp_path / 'edge_agg.o'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = [b for b in re.split(r'remark: Function Name: ', r.stderr)[1:] if 'global_edge_agg_bwd_wg_kernel' in b.split()[0]]
    assert len(blocks) == 2                                   # accumulate / first-layer instantiations
    for b in blocks:
        assert int(re.search(r'VGPRs: (\d+)', b).group(1)) + int(re.search(r'AGPRs: (\d+)', b).group(1)) <= 256
        assert int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b).group(1)) <= 12
        assert int(re.search(r'LDS Size \[bytes/block\]: (\d+)', b).group(1)) <= 160 * 1024

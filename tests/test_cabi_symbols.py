"""CPU-side checks of the drop-in boundary: libpamnet_hip.so builds for gfx950, loads, and exports every symbol that
include/pamnet_hip.h declares.  No compute calls (there is no GPU in the build container)."""
import ctypes
import os

import pytest

from pamnet_amd import build, lib


@pytest.fixture(scope='module')
def libpath():
    return build.build()


def test_library_builds_and_loads(libpath):
    assert os.path.exists(libpath)
    handle = lib.load()
    assert handle.pamnet_abi_version() >= 1


def test_every_declared_symbol_is_exported(libpath):
    decl = lib.declared_functions()
    assert len(decl) >= 20
    h = ctypes.CDLL(libpath)
    missing = [n for n in decl if not hasattr(h, n)]
    assert not missing, missing


def test_header_has_no_torch_types():
    text = open(lib.HEADER).read()
    assert 'torch' not in text.replace('torch_scatter', '').replace('torch_sparse', '').replace(
        'torch_cluster', '').replace('torch_geometric', '').replace('torch ATen', '').replace('no torch types', '')
    assert 'extern "C"' in text


def test_no_cpu_fallback():
    """The product refuses CPU tensors instead of silently computing elsewhere."""
    import torch
    from pamnet_amd import graph as G
    with pytest.raises(RuntimeError):
        G.exclusive_scan(torch.zeros(4, dtype=torch.int32))
    # the narrow-width operators and a whole narrow model likewise (no silent torch path for CPU tensors)
    from pamnet_amd import narrow
    lin = torch.nn.Linear(16, 16)
    assert not narrow.supported(torch.zeros(3, 16), 16)
    with pytest.raises(RuntimeError):
        narrow.linear(torch.zeros(3, 16), lin)
    import models
    from pamnet_amd import synth
    model = models.PAMNet(models.Config(dataset='QM9', dim=16, n_layer=1, cutoff_l=5.0, cutoff_g=5.0))
    with pytest.raises(RuntimeError):
        model(synth.qm9_batch(1, 0, 2))


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(lib.__file__)))
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, os.path.join(dp, f)

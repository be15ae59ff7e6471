"""physics-aware-multiplex-gnn_amd: MI355X-native hot path of PAMNet behind the reference's `models` API.

The directory name is not a Python identifier; load it with importlib.import_module('physics-aware-multiplex-gnn_amd')
(repo root on sys.path) or -- the drop-in route -- put this directory itself on sys.path and `from models import ...`."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from models import Config, PAMNet, PAMNet_s  # noqa: E402,F401

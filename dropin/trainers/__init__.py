"""`trainers` under the reference's own name (src/depth_train.py:11, src/pose_train.py:11:
`from trainers import *`; src/trainers/__init__.py:5-6).

Put THIS directory's parent (`<repo>/dropin`) in front of sys.path and nothing of the reference but its
`trainers` package is shadowed: `data`, `utils`, `common`, `net_config` still resolve to src/.  The module
object handed out is `lsps_amd.trainers` itself (tests/test_dropin_cpu.py).
"""
import importlib as _importlib
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.append(_root)
_real = _importlib.import_module('lsps_amd.trainers')
for _sub in ('lsps_trainer', 'lsps_nets', 'common_net', 'helpers', 'init'):
    _sys.modules[__name__ + '.' + _sub] = _sys.modules['lsps_amd.trainers.' + _sub]
_sys.modules[__name__] = _real

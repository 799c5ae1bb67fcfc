"""Drop-in `trainers` package for masabdi/LSPS's depth path on MI355X.

`from trainers import *` yields what the reference's package yields (src/trainers/__init__.py:5-6):
LSPSTrainer, the nets and blocks, gaussian_weights_init, get_model_list — and the names its driver
relies on without importing them (Variable, torch, nn, os, np; depth_train.py:135,145,220).

Two ways to reach it under the reference's own name (src/depth_train.py:11 `from trainers import *`),
both covered by tests/test_dropin_cpu.py in a fresh interpreter:
  * `sys.path.insert(0, '<repo>/dropin')`   -> dropin/trainers, a shim that shadows NOTHING but `trainers`
    (recommended: the reference's own `data`, `utils`, `common` packages stay importable);
  * `sys.path.insert(0, '<repo>/lsps_amd')` -> this file imported as top-level `trainers`.
Either way `sys.modules['trainers']` IS `lsps_amd.trainers` (one module object, one set of process-wide
math / Winograd modes, one dlopen of liblsps_hip.so).
"""
import sys as _sys

if __name__ == 'lsps_amd.trainers':
    from .lsps_trainer import *  # noqa: F401,F403
    from .init import *  # noqa: F401,F403
else:
    # Imported as a top-level package (this directory's parent is on sys.path): the relative imports of the
    # submodules (`from .. import ops`) need the real parent package, so hand over to it.  The import system
    # returns whatever sys.modules[name] holds once this file has run (importlib._bootstrap._load_unlocked).
    import importlib as _importlib
    import os as _os
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    _real = _importlib.import_module('lsps_amd.trainers')
    for _sub in ('lsps_trainer', 'lsps_nets', 'common_net', 'helpers', 'init'):
        _sys.modules[__name__ + '.' + _sub] = _sys.modules['lsps_amd.trainers.' + _sub]
    _sys.modules[__name__] = _real

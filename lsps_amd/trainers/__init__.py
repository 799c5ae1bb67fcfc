"""Drop-in `trainers` package for masabdi/LSPS's depth path on MI355X.

`from trainers import *` yields what the reference's package yields (src/trainers/__init__.py:5-6):
LSPSTrainer, the nets and blocks, gaussian_weights_init, get_model_list — and the names its driver
relies on without importing them (Variable, torch, nn, os, np; depth_train.py:135,145,220).
Put this directory's parent (`lsps_amd/`) on sys.path to import it as `trainers`.
"""
from .lsps_trainer import *  # noqa: F401,F403
from .init import *  # noqa: F401,F403

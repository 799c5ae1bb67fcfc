"""Weight initialisers (reference: src/trainers/init.py:8-17)."""
import numpy as np
import torch.nn.init as init


def gaussian_weights_init(m):
    """N(0, 0.02) on every module whose class name STARTS with 'Conv' (init.py:8-12) — i.e. the
    Conv2d / ConvTranspose2d parameter holders of common_net.py, never Linear."""
    if m.__class__.__name__.find('Conv') == 0:
        m.weight.data.normal_(0.0, 0.02)


def xavier_weights_init(m):
    """Unused by the shipped configs (init.py:14-17)."""
    if m.__class__.__name__.find('Conv') != -1:
        init.xavier_uniform_(m.weight, gain=np.sqrt(2))
        init.constant_(m.bias, 0.1)

"""Weight initialisers of the depth path.

`gaussian_weights_init` is applied with `module.apply(...)` by every block and by the trainer
(reference behaviour: src/trainers/init.py:8-12): modules whose CLASS NAME begins with "Conv" — here the
`Conv2d` / `ConvTranspose2d` parameter holders of common_net.py — get N(0, 0.02) weights; everything else
(Linear, placeholders, containers) is left alone.  `xavier_weights_init` exists for API parity
(src/trainers/init.py:14-17); no shipped config uses it.
"""
import math

from torch.nn import init as _init

GAUSSIAN_STD = 0.02


def _is_conv_holder(module, anywhere=False):
    name = type(module).__name__
    return ('Conv' in name) if anywhere else name.startswith('Conv')


def _announce(p):
    """A write through `p.data` is invisible to torch's version counters: tell the flat arena the parameter lives in (if any),
    so that packed weight panels kept across steps are rebuilt (optim.FlatArena.epoch / mark_dirty)."""
    arena = getattr(p, '_lsps_arena', None)
    if arena is not None:
        arena.mark_dirty()


def gaussian_weights_init(m):
    if _is_conv_holder(m):
        m.weight.data.normal_(mean=0.0, std=GAUSSIAN_STD)
        _announce(m.weight)


def xavier_weights_init(m):
    if _is_conv_holder(m, anywhere=True):
        _init.xavier_uniform_(m.weight, gain=math.sqrt(2.0))
        _init.constant_(m.bias, 0.1)
        _announce(m.weight)

"""Checkpoint lister and GAN accuracy counters (reference: src/trainers/helpers.py:9-32)."""
import os

import torch


def get_model_list(dirname, key, idx=-1):
    """Path of the idx-th (sorted) file in `dirname` whose name contains `key` and 'pkl' (helpers.py:9-18)."""
    if not os.path.exists(dirname):
        return None
    names = sorted(os.path.join(dirname, f) for f in os.listdir(dirname)
                   if os.path.isfile(os.path.join(dirname, f)) and key in f and 'pkl' in f)
    return names[idx]


def _flat_count(mask):
    return float(mask.reshape(-1).sum().item()), mask.numel() if mask.dim() == 3 else mask.size(0)


def _compute_true_acc(predictions):
    """fraction of probabilities >= 0.5 (helpers.py:20-25)."""
    hits, n = _flat_count(torch.ge(predictions.detach(), 0.5))
    return hits / (1.0 * n)


def _compute_fake_acc(predictions):
    """fraction of probabilities <= 0.5 (helpers.py:27-32)."""
    hits, n = _flat_count(torch.le(predictions.detach(), 0.5))
    return hits / (1.0 * n)

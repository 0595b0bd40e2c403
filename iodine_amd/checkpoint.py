"""Reading / writing checkpoints in the reference's on-disk format.

``Checkpointer.save`` (lib/utils/checkpoint.py:36-54) writes ``torch.save({'model': state_dict, 'optimizer': ...,
'scheduler'?: ..., **args})`` where ``args`` carries ``epoch`` (and ``iter``); the state_dict keys have a ``module.``
prefix when the model was wrapped in DataParallel (checkpoint.py:43 saves the wrapper).  ``Checkpointer.load``
(:56-80) maps everything to CPU (:118-119).  Parameter names / shapes of ``iodine_amd.IODINE`` are the reference's,
so a reference ``.pth`` loads directly.
"""
from __future__ import annotations

import torch


def strip_module_prefix(state_dict):
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}


def last_checkpoint(save_dir):
    """Path of the newest checkpoint listed in ``save_dir/checkpoint.pkl`` (the index ``Checkpointer.update_checkpoint``
    maintains, lib/utils/checkpoint.py:82-116), or None."""
    import os
    import pickle
    index = os.path.join(save_dir, 'checkpoint.pkl')
    if not os.path.exists(index):
        return None
    with open(index, 'rb') as f:
        names = pickle.load(f)
    return os.path.join(save_dir, names[-1]) if names else None


def load_checkpoint(path, model, optimizer=None, strict=True):
    """Load a reference-format checkpoint into ``model`` (and ``optimizer``).  Returns the extra entries (epoch, iter...)."""
    ckpt = torch.load(path, map_location=torch.device('cpu'))
    if 'model' not in ckpt:
        raise KeyError("not a reference checkpoint: no 'model' entry")
    model.load_state_dict(strip_module_prefix(ckpt.pop('model')), strict=strict)
    if hasattr(model, 'mark_params_dirty'):
        model.mark_params_dirty()                  # re-pack the library's weight copies on the next call
    opt_state = ckpt.pop('optimizer', None)
    if optimizer is not None and opt_state is not None:
        optimizer.load_state_dict(opt_state)
    ckpt.pop('scheduler', None)
    return ckpt


def save_checkpoint(path, model, optimizer=None, data_parallel_prefix=False, **args):
    """Write what ``Checkpointer.save`` writes; ``data_parallel_prefix`` reproduces the ``module.`` key prefix."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if data_parallel_prefix:
        sd = {'module.' + k: v for k, v in sd.items()}
    data = {'model': sd}
    if optimizer is not None:
        data['optimizer'] = optimizer.state_dict()
    data.update(args)
    torch.save(data, path)

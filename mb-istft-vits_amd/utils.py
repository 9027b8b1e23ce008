"""The `hps` half of the drop-in API.

Mirrors what the reference's inference callers use from `utils.py`:
`get_hparams_from_file` (`utils.py:199-205`), `HParams` (`utils.py:243-272`)
and `load_checkpoint(path, model, None)` (`utils.py:22-47`).  Training-side
helpers (logging, tensorboard, plots, wav loading) are out of scope.
"""
import json
import logging
import os

import torch

logger = logging.getLogger("mb_istft_vits_amd")


class HParams:
    """Nested attribute/dict hybrid: ``hps.model.n_heads``, ``hps["model"]``,
    ``**hps.model`` all work, as the callers rely on (`tts_vits.py:77-82`)."""

    def __init__(self, **entries):
        for key, value in entries.items():
            setattr(self, key, HParams(**value) if isinstance(value, dict) else value)

    # mapping protocol (enables ** splatting)
    def keys(self):
        return vars(self).keys()

    def items(self):
        return vars(self).items()

    def values(self):
        return vars(self).values()

    def __len__(self):
        return len(vars(self))

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in vars(self)

    def __iter__(self):
        return iter(vars(self))

    def __repr__(self):
        return repr(vars(self))

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, HParams) else v) for k, v in vars(self).items()}


def get_hparams_from_file(config_path):
    with open(config_path, "r") as f:
        return HParams(**json.load(f))


def builtin_config(name):
    """Path of one of the four BASELINE configs shipped with this package
    (`ljs_mini_mb_istft_vits`, `ljs_mb_istft_vits`, `ljs_ms_istft_vits`,
    `uudb_ms_istft_vits_ms`)."""
    if not name.endswith(".json"):
        name += ".json"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", name)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    return path


def load_checkpoint(checkpoint_path, model, optimizer=None):
    """Key-by-key tolerant load: keys absent from the checkpoint keep the
    model's current values (`utils.py:35-40`).  Returns the reference's
    4-tuple ``(model, optimizer, learning_rate, iteration)``.

    The file is read with ``weights_only=True`` (nothing in it is executed).
    """
    if not os.path.isfile(checkpoint_path):
        raise AssertionError(checkpoint_path)          # reference: bare assert, utils.py:23
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    iteration = ckpt["iteration"]
    learning_rate = ckpt["learning_rate"]
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    saved = ckpt["model"]
    target = model.module if hasattr(model, "module") else model
    merged = {}
    for key, current in target.state_dict().items():
        if key in saved:
            merged[key] = saved[key]
        else:
            logger.info("%s is not in the checkpoint", key)
            merged[key] = current
    target.load_state_dict(merged)
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    return model, optimizer, learning_rate, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, checkpoint_path):
    """Same dict layout as `utils.py:50-60`, so files round-trip with the reference."""
    target = model.module if hasattr(model, "module") else model
    torch.save({"model": target.state_dict(), "iteration": iteration,
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "learning_rate": learning_rate}, checkpoint_path)

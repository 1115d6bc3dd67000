"""Optimizer checkpoints with the reference's file name and dictionary keys (litegs/io_manager/checkpoint.py:4-24): the whole
optimizer object (parameters ride in its param groups) + the lr scheduler + the epoch, one ``torch.save``."""
from __future__ import annotations

import os

import torch

_ORDER = ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")


def save_checkpoint(model_path: str, epoch: int, optimizer, schedular) -> str:
    os.makedirs(model_path, exist_ok=True)
    file_path = os.path.join(model_path, "chkpnt{0}.pth".format(epoch))
    torch.save({"epoch": epoch, "optimizer": optimizer, "schedular": schedular}, file_path)
    return file_path


def load_checkpoint(file_path: str):
    """-> xyz, scale, rot, sh_0, sh_rest, opacity, start_epoch, optimizer, schedular.

    TRUSTED INPUT ONLY: the reference's checkpoints pickle the optimizer and scheduler OBJECTS (litegs/io_manager/__init__.py),
    so this has to unpickle arbitrary classes (``weights_only=False``) -- loading a checkpoint executes whatever its pickle
    says.  Load only files you wrote."""
    d = torch.load(file_path, weights_only=False)
    opt = d["optimizer"]
    by_name = {g["name"]: g["params"][0] for g in opt.param_groups}
    return (*[by_name[n] for n in _ORDER], d["epoch"] + 1, opt, d["schedular"])

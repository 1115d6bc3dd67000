"""Parameter groups and their command line -- host mirror of ``litegs/arguments.py`` + ``litegs/config/__init__.py``: the same
four groups, attribute names, defaults and flag spellings (``-s/--source_path``, ``-m/--model_path``, ``-i/--images``,
``-r/--resolution``, ``--eval`` ...), so the reference's command lines carry over."""
from __future__ import annotations

import dataclasses
from argparse import ArgumentParser, Namespace
from dataclasses import dataclass

from .densify import DensifyParams  # noqa: F401
from .optimizer import OptimizationParams  # noqa: F401
from .render import PipelineParams  # noqa: F401


@dataclass
class ModelParams:
    """Defaults of litegs/arguments.py:57-66."""
    sh_degree: int = 3
    source_path: str = ""
    model_path: str = ""
    images: str = "images"
    resolution: int = -1
    white_background: bool = False
    data_device: str = "cuda"
    eval: bool = False


_SHORTHAND = {"source_path", "model_path", "images", "resolution", "white_background"}      # the reference's underscore-prefixed fields
GROUPS = (ModelParams, OptimizationParams, PipelineParams, DensifyParams)


def add_cmdline_args(parser: ArgumentParser, defaults=None) -> None:
    """one argument group per parameter class; bools are ``store_true`` flags, the rest typed by their default"""
    defaults = defaults or [g() for g in GROUPS]
    for cls, obj in zip(GROUPS, defaults):
        group = parser.add_argument_group(cls.__name__)
        for f in dataclasses.fields(cls):
            value = getattr(obj, f.name)
            names = ["--" + f.name] + (["-" + f.name[0]] if f.name in _SHORTHAND else [])
            if isinstance(value, bool):
                group.add_argument(*names, default=value, action="store_true")
            elif isinstance(value, tuple):
                group.add_argument(*names, default=value, type=int, nargs=len(value))
            else:
                group.add_argument(*names, default=value, type=type(value))


def extract(args: Namespace):
    """-> (lp, op, pp, dp) filled from parsed arguments"""
    out = []
    for cls in GROUPS:
        kw = {f.name: getattr(args, f.name) for f in dataclasses.fields(cls) if hasattr(args, f.name)}
        if "tile_size" in kw:
            kw["tile_size"] = tuple(kw["tile_size"])
        out.append(cls(**kw))
    return tuple(out)


def get_default_arg():
    """litegs/config/__init__.py:3-8"""
    return ModelParams(), OptimizationParams(), PipelineParams(), DensifyParams()

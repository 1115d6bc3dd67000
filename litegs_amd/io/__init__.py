"""Data formats either side of the hot path (reference package ``litegs/io_manager``): COLMAP sparse models,
3DGS point-cloud .ply files and optimizer checkpoints.  Same function names as the reference's package."""
from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401
from .colmap import load_colmap_result  # noqa: F401
from .ply import load_ply, save_ply  # noqa: F401

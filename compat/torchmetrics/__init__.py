"""Stand-in for torchmetrics, limited to the image metrics MooreThreads/LiteGS uses (see compat/README.md)."""
from . import image  # noqa: F401

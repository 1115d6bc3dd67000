from . import lpip, psnr, ssim  # noqa: F401
from .lpip import LearnedPerceptualImagePatchSimilarity  # noqa: F401
from .psnr import PeakSignalNoiseRatio  # noqa: F401
from .ssim import StructuralSimilarityIndexMeasure  # noqa: F401

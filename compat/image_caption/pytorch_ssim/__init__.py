"""Shim for `Image_Caption/pytorch_ssim` (see compat/README.md)."""
import ppv_amd  # noqa: F401
from ppv_amd.ssim import SSIM, ssim  # noqa: F401

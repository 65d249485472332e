"""Shim for `Image_Caption/Camera/Lens.py` (see compat/README.md)."""
import ppv_amd  # noqa: F401
from ppv_amd.camera_lens import OpticsZernike, conv2D  # noqa: F401

"""Shim for `Face-DeId/Camera/Optics.py` (see compat/README.md)."""
import ppv_amd  # noqa: F401
from ppv_amd.camera_optics import Camera  # noqa: F401

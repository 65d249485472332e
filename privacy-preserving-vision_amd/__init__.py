"""MI355X-native Camera + ResNet-101 hot path.  Imported as ``ppv_amd``."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]

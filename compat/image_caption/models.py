"""Shim for `Image_Caption/models.py`: same names, MI355X implementations (see compat/README.md)."""
import torch

import ppv_amd  # noqa: F401  (registers the package alias)
from ppv_amd.encoder import Encoder  # noqa: F401
from ppv_amd.decoder import Attention, DecoderWithAttention  # noqa: F401

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")   # models.py:5 (unused by the shimmed classes)

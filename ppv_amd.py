"""Import alias: ``import ppv_amd`` loads the package that lives in ``privacy-preserving-vision_amd/``
(the directory name carries hyphens, which Python identifiers cannot)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "privacy-preserving-vision_amd")
_spec = importlib.util.spec_from_file_location("ppv_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ppv_amd"] = _mod
_spec.loader.exec_module(_mod)

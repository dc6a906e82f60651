"""Import shim: the package directory is `mtd-gan_amd/` (not a valid Python identifier), so
`import mtd_gan_amd` loads that directory as the package `mtd_gan_amd`."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mtd-gan_amd")
_spec = importlib.util.spec_from_file_location(
    "mtd_gan_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mtd_gan_amd"] = _mod
_spec.loader.exec_module(_mod)

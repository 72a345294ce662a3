"""Import shim: the package directory is named `apex-studio_amd/` (not a valid Python identifier),
so `import apex_studio_amd` loads it from that directory and installs it under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "apex-studio_amd")
_spec = importlib.util.spec_from_file_location(
    "apex_studio_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["apex_studio_amd"] = _mod
_spec.loader.exec_module(_mod)

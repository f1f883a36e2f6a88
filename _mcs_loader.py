"""Import helper: the package directory is `montecarloscattering.jl_amd/` (the
name the build contract fixes); a dot is not legal in a Python module name, so
the package is registered as `mcs_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "montecarloscattering.jl_amd")


def load():
    if "mcs_amd" in sys.modules:
        return sys.modules["mcs_amd"]
    spec = importlib.util.spec_from_file_location(
        "mcs_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["mcs_amd"] = mod
    spec.loader.exec_module(mod)
    return mod

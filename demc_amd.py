"""Import shim: the package directory is named `differentialevolutionmcmc.jl_amd` (with a dot), which Python's
import statement cannot spell.  `import demc_amd` loads that directory as the package `demc_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "differentialevolutionmcmc.jl_amd")
_spec = importlib.util.spec_from_file_location("demc_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["demc_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import alias: `import neorl_industrial_gym_amd as ni`.

The package directory is `neorl-industrial-gym_amd/` (not a valid Python identifier), so
this one-file module loads it under the importable name and replaces itself with it.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neorl-industrial-gym_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

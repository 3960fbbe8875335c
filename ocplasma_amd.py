"""Import alias: `import ocplasma_amd` loads the package that lives in the (non-identifier)
directory `optimal-control-1d-electrostatic-plasma_amd/`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "optimal-control-1d-electrostatic-plasma_amd")
_spec = importlib.util.spec_from_file_location("ocplasma_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ocplasma_amd"] = _mod
_spec.loader.exec_module(_mod)

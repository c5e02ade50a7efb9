"""Import shim: the package directory is `nbody-simulation-parallel_amd/` (hyphenated, after the
reference repository), which Python cannot name in an import statement.  `import nbody_amd` loads
it under the module name `nbody_simulation_parallel_amd` and re-exports its public names."""
import importlib.util
import os
import sys

_NAME = "nbody_simulation_parallel_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nbody-simulation-parallel_amd")

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, "__init__.py"),
                                                   submodule_search_locations=[_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
package = sys.modules[_NAME]
globals().update({k: v for k, v in vars(package).items() if not k.startswith("_")})

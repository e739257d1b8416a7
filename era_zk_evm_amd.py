"""Import shim: the package directory is named `era-zk_evm_amd/` (not a valid Python
identifier), so `import era_zk_evm_amd` loads that directory as a package."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "era-zk_evm_amd")
_spec = importlib.util.spec_from_file_location(
    "era_zk_evm_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["era_zk_evm_amd"] = _mod
_spec.loader.exec_module(_mod)

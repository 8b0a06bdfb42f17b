"""Import helper: the package directory is named `0-kno-vectorx_amd` (not a Python
identifier), so it is loaded under the alias `vectorx_amd`."""
import importlib.util
import os
import sys

_ALIAS = "vectorx_amd"


def load():
    if _ALIAS in sys.modules:
        return sys.modules[_ALIAS]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "0-kno-vectorx_amd")
    spec = importlib.util.spec_from_file_location(_ALIAS, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod

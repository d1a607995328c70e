"""Import helpers: the product package directory is named `quadray-engine_amd` (not an identifier)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_package():
    """Return the `quadray-engine_amd` package module (ctypes binding of libqrhip.so)."""
    name = "quadray_engine_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "quadray-engine_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod

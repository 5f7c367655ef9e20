"""Imports the package directory `ray-tracer_amd/` under the importable name `ray_tracer_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "ray_tracer_amd" in sys.modules:
        return sys.modules["ray_tracer_amd"]
    pkg_dir = os.path.join(ROOT, "ray-tracer_amd")
    spec = importlib.util.spec_from_file_location("ray_tracer_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ray_tracer_amd"] = mod
    spec.loader.exec_module(mod)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return mod

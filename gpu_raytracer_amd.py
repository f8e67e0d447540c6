"""Import shim: the package directory is named ``gpu-raytracer_amd`` (not a valid Python
identifier), so ``import gpu_raytracer_amd`` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu-raytracer_amd")
_spec = importlib.util.spec_from_file_location("gpu_raytracer_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules["gpu_raytracer_amd"] = _module
_spec.loader.exec_module(_module)

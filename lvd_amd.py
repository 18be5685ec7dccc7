"""Import alias: ``import lvd_amd`` loads the package that lives in ``llm-groundedvideodiffusion_amd/``.

The directory name is fixed by the project layout contract and is not a valid Python identifier,
so this one-file loader registers it under the importable name ``lvd_amd``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm-groundedvideodiffusion_amd")
_spec = importlib.util.spec_from_file_location(
    "lvd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lvd_amd"] = _mod
_spec.loader.exec_module(_mod)

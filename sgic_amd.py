"""Import alias: the package directory is `searchable-generative-image-compression_amd/`, which is not a
valid Python identifier, so `import sgic_amd` loads that directory as the package `sgic_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "searchable-generative-image-compression_amd")
_spec = importlib.util.spec_from_file_location("sgic_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sgic_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import alias: the product package lives in the directory
`multi-head-recommendation-with-human-priors_amd/` (not a valid Python identifier);
`import mhr_amd` loads it under this name, sub-modules included (`mhr_amd.ops`, ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-head-recommendation-with-human-priors_amd")
_spec = importlib.util.spec_from_file_location("mhr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mhr_amd"] = _mod
_spec.loader.exec_module(_mod)

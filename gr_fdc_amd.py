"""Import shim: the package directory is named `gr-fdc_amd/` (not a Python identifier), so
`import gr_fdc_amd` loads that directory as the package `gr_fdc_amd`."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gr-fdc_amd")
_spec = importlib.util.spec_from_file_location("gr_fdc_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_m = importlib.util.module_from_spec(_spec)
sys.modules["gr_fdc_amd"] = _m
_spec.loader.exec_module(_m)

"""Import shim: exposes the package directory
``lightweight-human-pose-estimation.pytorch_amd/`` (not a valid Python identifier)
under the importable name ``lwpose_amd``.

    import lwpose_amd
    from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "lightweight-human-pose-estimation.pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "lwpose_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lwpose_amd"] = _mod
_spec.loader.exec_module(_mod)

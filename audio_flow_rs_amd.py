"""Import shim: the package directory is ``audio-flow-rs_amd/`` (the hyphen is
part of the project's name and not a valid Python identifier), so
``import audio_flow_rs_amd`` loads that directory as a package under this name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "audio-flow-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "audio_flow_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["audio_flow_rs_amd"] = _mod
_spec.loader.exec_module(_mod)

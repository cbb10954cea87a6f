"""`import hopmi` -> the product package.

The build contract fixes the package directory name
(`hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd`),
which is not a Python identifier; this loader registers that directory under the name `hopmi`.
"""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd")
_spec = importlib.util.spec_from_file_location("hopmi", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hopmi"] = _mod
_spec.loader.exec_module(_mod)

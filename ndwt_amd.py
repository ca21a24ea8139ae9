"""Importable alias of the package directory `non-decimated_wavelets_amd/` (a hyphen is not a valid identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("non-decimated_wavelets_amd")
sys.modules[__name__] = _pkg

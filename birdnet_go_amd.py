"""Import alias: `import birdnet_go_amd` -> the package in ./birdnet-go_amd/ (hyphenated dir)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("birdnet-go_amd")
sys.modules[__name__] = _pkg

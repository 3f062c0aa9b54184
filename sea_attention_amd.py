"""Import shim: `import sea_attention_amd` loads the package that lives in `sea-attention_amd/`.

The package directory carries the repository's name (with a hyphen), which Python cannot import
directly; this module replaces itself in sys.modules with the real package.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sea-attention_amd")
_spec = importlib.util.spec_from_file_location(
    "sea_attention_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sea_attention_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Importable alias for the hot-path package.

The package directory is named ``preference-guided-image-captioning-alignment_amd``
(hyphens: not a Python identifier).  This shim makes it importable as
``pgca_amd`` by pointing ``__path__`` at that directory and executing its
``__init__``; sub-modules then resolve normally (``import pgca_amd.hip``).
"""
import os as _os

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_PKG_DIR = _os.path.join(_ROOT, "preference-guided-image-captioning-alignment_amd")
__path__ = [_PKG_DIR]
REPO_ROOT = _ROOT
PKG_DIR = _PKG_DIR

with open(_os.path.join(_PKG_DIR, "__init__.py"), "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))

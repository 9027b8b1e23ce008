"""Importable alias for the ``mb-istft-vits_amd/`` package directory.

The product directory carries the reference project's hyphenated name, which
is not a legal Python identifier.  This shim makes ``import mb_istft_vits_amd``
resolve to that directory: it points ``__path__`` at it and executes its
``__init__.py`` in this module's namespace.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "mb-istft-vits_amd")
if not _os.path.isdir(_real):  # pragma: no cover
    raise ImportError("mb-istft-vits_amd/ package directory not found next to %s" % _here)
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))

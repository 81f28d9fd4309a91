"""Importable alias of the product package.

The package directory is named ``constrastive-predictive-coding-audio_amd`` (after the reference repository), which is
not a valid Python identifier; this shim makes it importable as ``cpc_audio_amd`` by pointing the package search path
at that directory.  ``cpc_audio_amd.audio_model`` etc. resolve to the files there.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "constrastive-predictive-coding-audio_amd")
__path__.insert(0, _REAL)

with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f

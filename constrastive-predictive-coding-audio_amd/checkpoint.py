"""Reading the reference's snapshots without the reference's code.

The reference saves WHOLE-MODULE pickles (its SnapshotManager, setup_functions.py:134-164: ``torch.save(model, path)``), which
name its classes (``audio_model.AudioPredictiveCodingModel`` ...).  Unpickling them normally needs those modules on the import
path.  Here every class from a module that is not importable is replaced by an empty ``nn.Module`` stand-in while unpickling:
``nn.Module`` restores its ``_parameters`` / ``_buffers`` / ``_modules`` from the pickled state without calling ``__init__``, so
the stand-in tree answers ``state_dict()`` with the reference's keys, and those are the keys of the classes in this package.

    sd = reference_state_dict("snapshot_12000")            # OrderedDict, the reference's key names
    model = AudioPredictiveCodingModel(...same configuration...)
    load_reference_snapshot(model, "snapshot_12000")       # strict load_state_dict

Plain ``state_dict`` files (``torch.save(model.state_dict(), path)``) load as they are.
"""
import pickle
import types
from collections import OrderedDict

import torch
import torch.nn as nn

_STUBS = {}


def _stand_in(module: str, name: str):
    key = (module, name)
    if key not in _STUBS:
        _STUBS[key] = type(name, (nn.Module,), {"__module__": module, "__doc__": f"stand-in for {module}.{name}"})
    return _STUBS[key]


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)          # (also applies pickle's own Python-2 name fixes)
        except (ImportError, AttributeError):
            return _stand_in(module, name)


_pickle_module = types.ModuleType("cpc_audio_amd._reference_pickle")
_pickle_module.Unpickler = _Unpickler
_pickle_module.load = lambda f, **kw: _Unpickler(f, **kw).load()
_pickle_module.__dict__.update({k: getattr(pickle, k) for k in ("HIGHEST_PROTOCOL", "dump", "dumps", "Pickler", "PickleError",
                                                                "UnpicklingError")})


def reference_state_dict(path, map_location="cpu") -> "OrderedDict[str, torch.Tensor]":
    """The state_dict of a reference snapshot: a whole-module pickle (any of the reference's model classes) or a plain
    state_dict file.  Only tensors and the reference's key names are taken from the file."""
    obj = torch.load(path, map_location=map_location, pickle_module=_pickle_module, weights_only=False)
    if isinstance(obj, nn.Module):
        return obj.state_dict()
    if isinstance(obj, dict) and all(isinstance(v, torch.Tensor) for v in obj.values()):
        return OrderedDict(obj)
    raise TypeError(f"{path}: neither a pickled module nor a state_dict (got {type(obj).__name__})")


def load_reference_snapshot(model: nn.Module, path, strict: bool = True, map_location="cpu"):
    """Loads the parameters / buffers of a reference snapshot into ``model`` (built with the same configuration)."""
    sd = reference_state_dict(path, map_location=map_location)
    model.load_state_dict(sd, strict=strict)
    return model

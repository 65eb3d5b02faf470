"""
Import shim that lets the build container's leftover Anaconda interpreter
(/opt/conda/bin/python3.9: numpy 1.26, numba 0.54) import the reference
package read-only from /root/reference.  Tooling only: contains no reference
code.  Used by make_golden.py; never shipped to / used on the GPU box.
"""
import sys
import types

import numpy as _np

_np.MachAr = type("MachAr", (), {})  # removed in numpy>=1.24, numba 0.54 references it
_m = types.ModuleType("numba.np.ufunc._internal")  # C ext that fails to init against numpy 1.26


class _DUFunc(object):
    def __init__(self, *a, **k):
        pass


_m._DUFunc = _DUFunc
_m.PyUFunc_Zero = 0
_m.PyUFunc_One = 1
_m.PyUFunc_None = -1
_m.PyUFunc_ReorderableNone = -2
sys.modules["numba.np.ufunc._internal"] = _m
_v = _np.__version__
_np.__version__ = "1.20.3"  # pass numba's numpy version gate
import numba  # noqa: E402

_np.__version__ = _v
for _name in ("astropy", "astropy.coordinates", "astropy.time", "astropy.units"):
    sys.modules[_name] = None  # the old conda astropy is broken; make it look absent
if not hasattr(numba.config, "NRT_STATS"):
    numba.config.NRT_STATS = 0

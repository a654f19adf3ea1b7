"""ctypes handle on lib/libgdpt_hip.so.  Fails loudly when the HIP library is missing: there is no
Python/CPU fallback for any op in this package."""
import ctypes as C
import os

from . import _build

_lib = None


class GdptError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("GDPT_LIB") or _build.LIB          # GDPT_LIB: another BUILD of the same HIP sources (the -O1 fence, _build.FENCE_LIB)
        if not os.path.exists(path):
            raise GdptError("HIP library %s is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(no CPU fallback exists)" % path)
        _lib = C.CDLL(path)
        _lib.gdpt_last_error.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise GdptError("gdpt error %d: %s" % (rc, lib().gdpt_last_error().decode()))

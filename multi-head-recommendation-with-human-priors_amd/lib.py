"""ctypes binding of libmhr_hip.so (the C ABI declared in include/mhr.h).

The product path has no CPU or eager fallback: if the HIP library is missing, loading fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmhr_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mhr.h")

F32, BF16 = 0, 1

_p, _i, _l, _f, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64

def _parse_header():
    """name -> ctypes argtypes, derived from the declarations in include/mhr.h so the binding cannot drift."""
    with open(HEADER) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(mhr_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(2), m.group(3).strip()
        if m.group(1) == "int64_t":
            RETURNS_INT64.add(name)
        args = []
        if params and params != "void":
            for prm in params.split(","):
                prm = " ".join(prm.split())
                if "*" in prm:
                    args.append(_p)
                elif prm.startswith("int64_t"):
                    args.append(_l)
                elif prm.startswith("uint64_t"):
                    args.append(_u64)
                elif prm.startswith("float"):
                    args.append(_f)
                elif prm.startswith("int"):
                    args.append(_i)
                else:
                    raise RuntimeError(f"mhr.h: cannot map parameter '{prm}' of {name}")
        sigs[name] = args
    return sigs


RETURNS_INT64 = set()              # size queries (filled from the header: functions declared `int64_t mhr_...`)
SIGNATURES = _parse_header()


def declared_symbols():
    """Every `int mhr_*(` / `const char* mhr_*(` entry point declared in include/mhr.h."""
    with open(HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(mhr_[a-z0-9_]+)\s*\(", text)))


class _Lib:
    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                    "There is no CPU/eager fallback for the product path.")
            dll = ctypes.CDLL(LIB_PATH)
            dll.mhr_last_error.restype = ctypes.c_char_p
            dll.mhr_last_error.argtypes = []
            for name, args in SIGNATURES.items():
                fn = getattr(dll, name, None)
                if fn is None:
                    raise RuntimeError(f"{LIB_PATH} does not export {name} (stale build? run `python __graft_entry__.py`)")
                fn.argtypes = args
                fn.restype = ctypes.c_int64 if name in RETURNS_INT64 else ctypes.c_int
            self._dll = dll
        return self._dll

    def call(self, name, *args):
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {dll.mhr_last_error().decode()}")


_lib = _Lib()
call = _lib.call
load = _lib.load


def check_exports():
    """The shared library must export every symbol the header declares (CPU-checkable, no GPU needed)."""
    dll = load()
    missing = [s for s in declared_symbols() if not hasattr(dll, s)]
    if missing:
        raise RuntimeError(f"libmhr_hip.so does not export: {missing}")
    unsig = [s for s in declared_symbols() if s not in SIGNATURES and s != "mhr_last_error"]
    if unsig:
        raise RuntimeError(f"lib.py has no ctypes signature for: {unsig}")
    return True

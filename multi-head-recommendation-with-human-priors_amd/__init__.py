"""MI355X-native hot path of Multi-Head Recommendation with Human Priors.

`lib`  - ctypes binding of libmhr_hip.so (C ABI: include/mhr.h)
`ops`  - torch-facing wrappers (device pointers + current stream)
The reference-shaped host classes live under `code/REC/` (put `<this dir>/code` on sys.path).
"""
from . import lib  # noqa: F401

__all__ = ["lib"]

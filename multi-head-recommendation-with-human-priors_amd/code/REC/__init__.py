"""Reference-shaped package layout (`REC.model.IDNet.hstu.HSTU`, `REC.evaluator.Collector`, `REC.utils.get_model`,
`REC.trainer.Trainer`) backed by the MI355X kernels.  Put this directory's parent (`.../code`) on sys.path where the
reference's `code/` used to be."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)
import mhr_amd  # noqa: E402,F401  (registers the kernel package under an importable name)

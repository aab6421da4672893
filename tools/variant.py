"""Build / run an experiment variant of the HIP library (extra -D flags), without touching the product library.

    python tools/variant.py build "EXP_NOSREAD EXP_NOTR"          # here: hipcc ... -DEXP_NOSREAD -DEXP_NOTR -> tools/_exp/<name>/libmhr_hip.so
    python tools/variant.py run   "EXP_NOSREAD EXP_NOTR" tools/nce_micro.py   # on the GPU box: the script runs against that library

The EXP_* timing experiments (garbage-value paths that REMOVE a piece of the tile step: no LDS reads, no transposed reads,
no waits) are not in the product sources: `tools/exp_variants.patch` adds them to a scratch copy of csrc/ that this script
builds from whenever a requested define starts with EXP_.
"""
import glob
import os
import runpy
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
defs = sys.argv[2].split()
OUT = os.path.join(ROOT, "tools", "_exp", "_".join(defs) or "base")
LIB = os.path.join(OUT, "libmhr_hip.so")
if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "csrc")
    if any(d.startswith("EXP_") for d in defs):               # the experiments live in a patch, not in the product sources
        import shutil
        pkg = os.path.join(OUT, "src", "pkg")
        shutil.rmtree(os.path.join(OUT, "src"), ignore_errors=True)
        shutil.copytree(csrc, os.path.join(pkg, "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(OUT, "src", "include"))    # csrc includes ../../include/mhr.h
        subprocess.check_call(["patch", "-p0", "-d", pkg, "-i", os.path.join(ROOT, "tools", "exp_variants.patch")])
        csrc = os.path.join(pkg, "csrc")
    # same flags and the same parallel per-file compile as the product build, objects kept per variant (incremental rebuilds)
    ge.build_hip_library(csrc=csrc, build_dir=os.path.join(OUT, "obj"), lib=LIB, extra_flags=["-D" + d for d in defs])
    print("built", LIB)
else:
    sys.path.insert(0, ROOT)
    import mhr_amd.lib as L
    L.LIB_PATH = LIB
    sys.argv = sys.argv[3:]
    runpy.run_path(sys.argv[0], run_name="__main__")

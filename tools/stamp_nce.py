"""In-kernel phase timing of the sampled-softmax backward (s_memtime stamps, cdna guide section 7).

    python tools/stamp_nce.py build     # here (no GPU): hipcc -DMHR_STAMP csrc/*.hip -> tools/_exp/stamp/libmhr_hip.so
    python tools/stamp_nce.py run       # on the GPU box: runs tools/nce_micro.py against that library, prints cycles/tile

The product library never contains the stamps; this script swaps mhr_amd.lib.LIB_PATH before the first load.
"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.environ.get("VARIANT", "")          # e.g. VARIANT="EXP_NOEPI" adds -DEXP_NOEPI and builds into tools/_exp/stamp_EXP_NOEPI
OUT = os.path.join(ROOT, "tools", "_exp", "stamp" + ("_" + VARIANT.replace(" ", "_") if VARIANT else ""))
LIB = os.path.join(OUT, "libmhr_hip.so")
NAMES = ["vmcnt wait", "barrier", "loop top / tail", "s,f MFMAs + epilogue", "E.N (tr) MFMAs + DMA issue", "-"]

if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "csrc", "*.hip")))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    cmd = [ge.HIPCC] + ge.HIP_FLAGS + ["-shared", "-DMHR_STAMP"] + ["-D" + v for v in VARIANT.split()] + ["-I", os.path.join(ROOT, "include"), "-o", LIB] + srcs
    subprocess.check_call(cmd)
    print("built", LIB)
else:
    sys.path.insert(0, ROOT)
    import mhr_amd.lib as L
    L.LIB_PATH = LIB
    import runpy
    runpy.run_path(os.path.join(ROOT, "tools", "nce_micro.py"), run_name="__main__")
    dll = ctypes.CDLL(LIB)
    buf = (ctypes.c_ulonglong * 16)()
    assert dll.mhr_debug_read_stamps(buf) == 0
    n_tiles = int(os.environ.get("TILES", 257))
    tot = sum(buf[:6])
    for k, nm in enumerate(NAMES):
        print(f"{nm:24s} {buf[k] / n_tiles:9.0f} cycles/tile  {100.0 * buf[k] / max(1, tot):5.1f} %")
    print(f"{'total':24s} {tot / n_tiles:9.0f} cycles/tile (s_memtime ticks = shader cycles; 100 MHz-independent)")

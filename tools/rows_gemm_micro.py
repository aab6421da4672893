"""Micro-driver: the encoder's token-rows projections (ops.rows_gemm) against the library GEMM at the cfg1 shapes.
python tools/rows_gemm_micro.py   (M=25600; N,K = 1024,256 / 256,256; also cfg0's 64-wide layers)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhr_amd  # noqa: F401
from mhr_amd import ops

def bench(fn, n=20, reps=5):
    """us per call, back to back inside one replayed hipGraph (no host launch gaps)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=st):
            for _ in range(n): fn()
        g_.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g_.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
SHAPES = [(25600, 1024, 256, False), (25600, 256, 256, True), (51200, 1024, 256, False), (6400, 256, 64, True), (6400, 256, 64, False)]
if os.environ.get("SHAPES"):
    SHAPES = SHAPES[:int(os.environ["SHAPES"])]
for (M, N, K, with_bias) in SHAPES:
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.06).bfloat16()
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16() if with_bias else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = torch.nn.functional.linear(a, w, b)
    ops.rows_gemm(a, w, b, out=out)
    torch.cuda.synchronize()
    exact = (a.float() @ w.float().t() + (b.float() if b is not None else 0)).bfloat16()
    bad = int((out != exact).sum()); bad_lib = int((ref != exact).sum())
    err = float((out.float() - exact.float()).abs().max()) / float(exact.float().abs().max())
    t_own = bench(lambda: ops.rows_gemm(a, w, b, out=out))
    wt_ = w.t().contiguous()
    t_kn = bench(lambda: ops.rows_gemm(a, wt_, b, out=out, w_is_kn=True)) if N % 256 == 0 else float("nan")
    wt = w.t().contiguous()
    t_lib = bench(lambda: torch.nn.functional.linear(a, w, b)) if os.environ.get("LIB", "1") != "0" else float("nan")
    t_lib2 = bench(lambda: a @ wt) if os.environ.get("LIB", "1") != "0" else float("nan")
    nbytes = 2 * (M * K + N * K + M * N)
    print(f"M={M} N={N} K={K} bias={with_bias}: own {t_own:.1f} us [W as K,N: {t_kn:.1f}] ({nbytes / t_own / 1e6:.2f} TB/s, {2 * M * N * K / t_own / 1e6:.0f} TF)  library {t_lib:.1f} us (a @ w_kn: {t_lib2:.1f}); "
          f"elements != fp32-product rounded: own {bad} lib {bad_lib} of {M * N}, max rel err {err:.2e}")

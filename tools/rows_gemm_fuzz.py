"""Fuzz ops.rows_gemm: random (M, N, K, weight layout, bias, strided rows) against the fp32 product of the bf16 operands, and
every case launched several times - a tile consumed before its LDS-DMA pieces landed (the counted vmcnt waits) would show as a
run-to-run difference.  python tools/rows_gemm_fuzz.py [seconds]"""
import os, random, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhr_amd  # noqa: F401
from mhr_amd import ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(4321)
t0, n, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    K = rng.choice([64, 128, 256])
    kn = rng.random() < 0.5
    N = rng.choice([256, 512, 768, 1024]) if kn else rng.choice([8, 72, 200, 256, 520, 1024, 1032])
    M = rng.choice([rng.randint(1, 200), rng.randint(200, 5000), rng.randint(5000, 60000), 25600, 51200])
    pad = rng.choice([0, 0, 8, 256])
    g = torch.Generator(device="cuda").manual_seed(rng.randint(0, 1 << 30))
    a_full = (torch.randn(M, K + pad, device="cuda", generator=g) * 0.5).bfloat16()
    a = a_full[:, :K]
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16() if rng.random() < 0.5 else None
    wd = w.t().contiguous() if kn else w
    ref = a.float() @ w.float().t() + (b.float() if b is not None else 0.0)
    first = None
    for rep in range(6):
        out = torch.full((M + 1, N), 3.0, dtype=torch.bfloat16, device="cuda")
        ops.rows_gemm(a, wd, b, out=out[:M], w_is_kn=kn)
        if first is None:
            first = out
        else:
            assert torch.equal(out, first), ("run-to-run difference", M, N, K, kn, pad)
    torch.cuda.synchronize()
    got = first[:M].float()
    tol = 2 ** -8 * ref.abs().clamp_min(1e-3) + 1e-6
    ratio = float(((got - ref).abs() / tol).max())
    worst = max(worst, ratio)
    assert ratio <= 2.0, (M, N, K, kn, pad, ratio)
    assert bool((first[M] == 3.0).all()), ("guard row written", M, N, K)
    n += 1
print(f"rows_gemm_fuzz: {n} random cases x 6 launches: bitwise repeatable, worst error {worst:.2f} half-ulps of bf16 (limit 2)")

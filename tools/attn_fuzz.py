"""Fuzz the two forms of the HSTU attention against each other: random (B, L, heads, head_dim, key masks); the forward must agree
to the last bit, the backward to bf16 rounding.  python tools/attn_fuzz.py [seconds]"""
import os, sys, time, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhr_amd  # noqa: F401
from mhr_amd import ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(1234)
t0, n, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    hd = rng.choice([8, 16, 24, 32, 64, 128])
    Hh = rng.choice([1, 2, 3, 4, 8])
    L = rng.choice([rng.randint(1, 40), rng.randint(33, 300), rng.randint(257, 640)])
    if 2 * ((L + 31) // 32) * (32 * (2 if hd <= 32 else (4 if hd <= 64 else 8)) * 32 // 1) > 150 * 1024:   # resident form must fit
        L = 200
    B = rng.choice([1, 2, 3, 5])
    D = Hh * hd
    g = torch.Generator(device="cuda").manual_seed(rng.randint(0, 1 << 30))
    h = torch.randn(B * L, 4 * D, device="cuda", generator=g).bfloat16()
    kind = rng.choice(["front", "random", "all", "none_first", "holes"])
    if kind == "front":
        ctx = torch.randint(0, L + 1, (B,), device="cuda", generator=g)
        valid = torch.arange(L, device="cuda")[None, :] >= (L - ctx)[:, None]
    elif kind == "random":
        valid = torch.rand(B, L, device="cuda", generator=g) > rng.random()
    elif kind == "all":
        valid = torch.ones(B, L, dtype=torch.bool, device="cuda")
    elif kind == "none_first":
        valid = torch.ones(B, L, dtype=torch.bool, device="cuda"); valid[0] = False
    else:
        valid = torch.rand(B, L, device="cuda", generator=g) > 0.1
        a = rng.randint(0, max(0, L - 1)); valid[:, a:a + rng.randint(1, 100)] = False
    kv = valid.to(torch.uint8).contiguous()
    d_out = torch.randn(B * L, D, device="cuda", generator=g).bfloat16()
    res = {}
    for mode in ("0", "1"):
        os.environ["MHR_ATTN_STREAM"] = mode
        out, _ = ops.hstu_attn_fwd(h, kv, B, L, Hh, hd, save_act=False)
        dh = torch.zeros_like(h)
        ops.hstu_attn_bwd(h, None, kv, d_out, dh, B, L, Hh, hd)
        res[mode] = (out, dh)
    os.environ["MHR_ATTN_STREAM"] = "0"                     # resident form with the per-batch sequence layout: bit-identical
    lay = ops.attn_seq_layout(kv, B, L, order=rng.random() < 0.5)[:2]
    out, _ = ops.hstu_attn_fwd(h, kv, B, L, Hh, hd, save_act=False, layout=lay)
    dh = torch.full_like(h, 3.0); dh[:, :D] = 0
    ops.hstu_attn_bwd(h, None, kv, d_out, dh, B, L, Hh, hd, layout=lay)
    torch.cuda.synchronize()
    (o0, d0), (o1, d1) = res["0"], res["1"]
    assert torch.equal(out, o0) and torch.equal(dh, d0), (B, L, Hh, hd, kind, "layout")
    assert torch.equal(o0, o1), (B, L, Hh, hd, kind)
    assert bool(torch.isfinite(d0.float()).all()) and bool(torch.isfinite(d1.float()).all()), (B, L, Hh, hd, kind)
    scale = float(d0.float().abs().max()) + 1e-12
    err = float((d0.float() - d1.float()).abs().max()) / scale
    worst = max(worst, err)
    assert err <= 2 ** -6, (B, L, Hh, hd, kind, err)
    n += 1
print(f"attn_fuzz: {n} random cases, forward bit-identical in all, worst backward difference {worst:.2e} of max")

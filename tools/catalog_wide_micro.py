"""Time the full-catalog decode at the wide shapes (cfg2 / TinyLlama / Baichuan2-7B widths); run on the GPU box."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mhr_amd import ops
MFMA = 2500.0
def i32(t):
    t = t & 0xFFFFFFFF
    return torch.where(t >= (1 << 31), t - (1 << 32), t).int()
for D, N, B, H in ((1024, 1 << 20, 64, 8), (2048, 453938, 256, 16), (4096, 453938, 64, 16)):
    g = torch.Generator(device="cuda").manual_seed(D)
    users = torch.nn.functional.normalize(torch.randn(B * H, D, device="cuda", generator=g), dim=-1).bfloat16()
    items = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=-1).bfloat16()
    C = 8
    tags = torch.rand(N, C, device="cuda", generator=g) < 0.375
    tag_bits = i32((tags.long() * (1 << torch.arange(C, device="cuda"))).sum(1) | (1 << 31))
    row_bits = i32(torch.tensor([1 << (r % C) for r in range(B * H)], device="cuda"))
    ops.PROFILE = {"mhr_catalog_score_emit_wide": []}
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = {}
        ov, oi = ops.catalog_topk(users, H, items, tag_bits, row_bits, None, None, 200, n_items=N, stats=st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    full = max(ops.profile_raw("mhr_catalog_score_emit_wide")[-3:]) if ops.PROFILE["mhr_catalog_score_emit_wide"] else float("nan")
    fl = 2.0 * B * H * D * N
    print(f"D={D} N={N} rows={B*H}: decode {dt*1e3:.2f} ms; full emit launch {full:.3f} ms = {fl/full/1e9:.0f} TFLOP/s ({fl/full/1e9/MFMA*100:.1f}% of MFMA peak); {st}", flush=True)

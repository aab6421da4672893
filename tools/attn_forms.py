"""Time the resident and the streamed form of the HSTU attention side by side (MHR_ATTN_STREAM=0/1) at the bench shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
sys.path.insert(0, ROOT)
import mhr_amd  # noqa: E402,F401
from mhr_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, L, Hh, hd in (("cfg1", 128, 200, 8, 32), ("cfg2", 64, 512, 16, 64), ("L1024", 32, 1024, 16, 64), ("L2048", 16, 2048, 8, 32),
                           ("L300x64", 64, 300, 8, 64)):
    D = Hh * hd
    g = torch.Generator(device="cuda").manual_seed(1)
    h = torch.randn(B * L, 4 * D, device="cuda", generator=g).bfloat16()
    ctx = torch.randint(L // 4, L + 1, (B,), device="cuda", generator=g)
    valid = (torch.arange(L, device="cuda")[None, :] >= (L - ctx)[:, None]).to(torch.uint8).contiguous()
    d_out = torch.randn(B * L, D, device="cuda", generator=g).bfloat16()
    dh = torch.zeros_like(h)
    for mode in ("0", "1"):
        os.environ["MHR_ATTN_STREAM"] = mode
        try:
            f = timeit(lambda: ops.hstu_attn_fwd(h, valid, B, L, Hh, hd, save_act=False))
            b = timeit(lambda: ops.hstu_attn_bwd(h, None, valid, d_out, dh, B, L, Hh, hd))
        except Exception as ex:  # noqa: BLE001
            print(name, "stream" if mode == "1" else "resident", "->", str(ex)[:80])
            continue
        print(f"{name:8s} {'stream  ' if mode == '1' else 'resident'} fwd {f:8.1f} us  bwd {b:8.1f} us", flush=True)

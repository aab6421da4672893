"""One shape of the wide catalog decode (for rocprofv3 --pmc passes); D N B H from argv."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mhr_amd import ops
D, N, B, H = (int(a) for a in sys.argv[1:5])
g = torch.Generator(device="cuda").manual_seed(D)
users = torch.nn.functional.normalize(torch.randn(B * H, D, device="cuda", generator=g), dim=-1).bfloat16()
items = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=-1).bfloat16()
row_bits = torch.full((B * H,), -(1 << 31), dtype=torch.int32, device="cuda")
for it in range(2):
    ov, oi = ops.catalog_topk(users, H, items, None, row_bits, None, None, 200, n_items=N)
torch.cuda.synchronize()
print("done")

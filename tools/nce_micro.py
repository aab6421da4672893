"""Micro-driver: the three sampled-softmax kernels at the cfg1 per-category shape (for rocprofv3 / timing)."""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mhr_amd
from mhr_amd import ops
D, n_tok, n_neg, cap, G = int(os.environ.get("D", 256)), int(os.environ.get("NTOK", 68000)), 8192, 204800, int(os.environ.get("GROUPS", 4))
reps = int(os.environ.get("REPS", 3))
g = torch.Generator(device="cuda").manual_seed(0)
q_rows = torch.randn(cap, D, device="cuda", generator=g)
p_rows = torch.randn(cap, D, device="cuda", generator=g)
negs = torch.nn.functional.normalize(torch.randn(G, n_neg, D, device="cuda", generator=g), dim=-1).bfloat16()
idx = torch.arange(cap, dtype=torch.int32, device="cuda")[None].repeat(G, 1).contiguous()
ntd = torch.tensor([n_tok] * G, dtype=torch.int32, device="cuda")
ls = torch.tensor([math.log(20.0)], device="cuda")
w = torch.rand(G, cap, device="cuda")
dq = torch.zeros(cap, D, device="cuda"); dp = torch.zeros(cap, D, device="cuda")
SHARE = bool(os.environ.get("SHARE"))        # the row-sharing path: the streaming kernels in their plain form (nothing suppressed)
FWD_ONLY = bool(os.environ.get("FWD_ONLY"))  # (stamped builds: the forward's stamps are the last ones written)
for it in range(reps):
    sv = ops.nce_fwd(q_rows, idx, p_rows, idx, negs, ntd, cap, ls, 0.99, want_logs=False, share_rows=SHARE)
    if not FWD_ONLY:
        dn, dls = ops.nce_bwd(sv, w, ls, idx, idx, dq, dp)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record(); sv = ops.nce_fwd(q_rows, idx, p_rows, idx, negs, ntd, cap, ls, 0.99, want_logs=False, share_rows=SHARE); ev[1].record()
if not FWD_ONLY:
    dn, dls = ops.nce_bwd(sv, w, ls, idx, idx, dq, dp)
ev[2].record()
torch.cuda.synchronize()
fl = 2.0 * G * n_tok * n_neg * D
print(f"fwd {ev[0].elapsed_time(ev[1]):.3f} ms ({2*fl/ev[0].elapsed_time(ev[1])/1e9:.0f} TF)  bwd(tokens+negs) {ev[1].elapsed_time(ev[2]):.3f} ms ({2*fl/ev[1].elapsed_time(ev[2])/1e9:.0f} TF algorithmic)  n_tok={n_tok} groups={G}")

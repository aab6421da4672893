import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mhr_amd
from mhr_amd import ops
def l2n(x): return x / x.norm(dim=-1, keepdim=True)
D, n_tok, n_neg, cap = int(os.environ.get("DD", 256)), 32, 32, 32
g = torch.Generator().manual_seed(5)
q_rows = torch.randn(cap, D, generator=g); p_rows = torch.randn(cap, D, generator=g)
negs = l2n(torch.randn(n_neg, D, generator=g)).bfloat16()
idx = torch.arange(cap, dtype=torch.int32)
ntd = torch.tensor([n_tok], dtype=torch.int32).cuda(); ls = torch.tensor([math.log(20.0)]).cuda()
sv = ops.nce_fwd(q_rows.cuda(), idx.cuda()[None], p_rows.cuda(), idx.cuda()[None], negs.cuda()[None], ntd, cap, ls, 0.99)
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
w = (torch.rand(cap, generator=g) + 0.5) if mode == "rand" else torch.ones(cap)
if mode == "one":   # single token carries weight
    w = torch.zeros(cap); w[int(sys.argv[2])] = 1.0
dq = torch.zeros(cap, D).cuda(); dp = torch.zeros(cap, D).cuda()
dn, dls = ops.nce_bwd(sv, w.cuda()[None], ls, idx.cuda()[None], idx.cuda()[None], dq, dp)
torch.cuda.synchronize()
def r(t): return t + (t.bfloat16().float() - t).detach()
q = q_rows[:n_tok].clone().requires_grad_(True); p = p_rows[:n_tok].clone().requires_grad_(True); nn_ = negs.float().clone().requires_grad_(True)
qn, pn = r(l2n(q)), r(l2n(p))
neg = qn @ nn_.T
logits = torch.cat([(qn*pn).sum(-1, keepdim=True)*20, neg*20], -1)
loss = torch.logsumexp(logits, -1) - logits[:, 0]
(loss * w[:n_tok]).sum().backward()
G = (torch.softmax(logits, -1)[:, 1:] * w[:n_tok, None]).detach()          # [tok, neg]
ref = 20 * G.T @ qn.detach()
got = dn.cpu()[0]
print("ref vs autograd", float((ref - nn_.grad).abs().max()))
# solve for the effective G the kernel used: got = 20 * Geff^T @ qn  ->  Geff^T = got @ pinv(qn) / 20
Geff = (got @ torch.linalg.pinv(qn.detach()) / 20).T
torch.set_printoptions(precision=3, linewidth=220, sci_mode=False)
ratio = Geff / G
print("Geff/G  rows=tokens 0..31, first 8 negs"); print(ratio[:, :8])
lse = torch.logsumexp(logits, -1).detach()
lw_true = lse * 1.4426950408889634 - torch.log2(w[:n_tok])
lw_used = lw_true - torch.log2(ratio.median(dim=1).values)
for t in range(12):
    d = (lw_true - lw_used[t]).abs()
    print(t, "lw_true %.3f lw_used %.3f  closest token %d (diff %.4f)  lse-only %.3f" % (lw_true[t], lw_used[t], int(d.argmin()), float(d.min()), lse[t] * 1.4426950408889634))

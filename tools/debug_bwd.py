import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mhr_amd
from mhr_amd import ops
def l2n(x): return x / x.norm(dim=-1, keepdim=True)
def run(D, n_tok, n_neg, cap):
    g = torch.Generator().manual_seed(5)
    q_rows = torch.randn(cap, D, generator=g); p_rows = torch.randn(cap, D, generator=g)
    negs = l2n(torch.randn(n_neg, D, generator=g)).bfloat16()
    idx = torch.arange(cap, dtype=torch.int32)
    ntd = torch.tensor([n_tok], dtype=torch.int32).cuda(); ls = torch.tensor([math.log(20.0)]).cuda()
    sv = ops.nce_fwd(q_rows.cuda(), idx.cuda()[None], p_rows.cuda(), idx.cuda()[None], negs.cuda()[None], ntd, cap, ls, 0.99)
    w = torch.rand(cap, generator=g) + 0.5; w[n_tok:] = 0
    dq = torch.zeros(cap, D).cuda(); dp = torch.zeros(cap, D).cuda()
    dn, dls = ops.nce_bwd(sv, w.cuda()[None], ls, idx.cuda()[None], idx.cuda()[None], dq, dp)
    torch.cuda.synchronize()
    def r(t): return t + (t.bfloat16().float() - t).detach()
    q = q_rows[:n_tok].clone().requires_grad_(True); p = p_rows[:n_tok].clone().requires_grad_(True); nn_ = negs.float().clone().requires_grad_(True)
    qn, pn = r(l2n(q)), r(l2n(p))
    neg = qn @ nn_.T; fix = pn @ nn_.T; keep = ~(fix > 0.99)
    logits = torch.cat([(qn*pn).sum(-1, keepdim=True)*20, torch.where(keep, neg*20, torch.full_like(neg, -1e30))], -1)
    loss = torch.logsumexp(logits, -1) - logits[:, 0]
    (loss * w[:n_tok]).sum().backward()
    eq = (dq.cpu()[:n_tok]-q.grad).abs(); en = (dn.cpu()[0]-nn_.grad).abs()
    print(f"D={D} n_tok={n_tok} n_neg={n_neg} cap={cap}: dq err {float(eq.max()):.4g}/{float(q.grad.abs().max()):.3g}  dn err {float(en.max()):.4g}/{float(nn_.grad.abs().max()):.3g}"
          f"  dq beyond n_tok {float(dq.cpu()[n_tok:].abs().max()) if cap>n_tok else 0:.3g}")
    if float(eq.max()) > 0.05 * float(q.grad.abs().max()):
        bad = (eq > 0.05 * float(q.grad.abs().max()))
        print("   dq bad rows:", torch.nonzero(bad.any(1)).flatten().tolist()[:40], " bad cols:", torch.nonzero(bad.any(0)).flatten().tolist()[:40])
    if float(en.max()) > 0.05 * float(nn_.grad.abs().max()):
        bad = (en > 0.05 * float(nn_.grad.abs().max()))
        print("   dn bad rows:", torch.nonzero(bad.any(1)).flatten().tolist()[:40], " bad cols:", torch.nonzero(bad.any(0)).flatten().tolist()[:40])
import sys
cases = [(32, 32, 64, 32), (64, 32, 64, 32), (64, 32, 32, 32), (128, 32, 64, 32), (128, 64, 256, 64)]
for D, nt, nn, cap in cases:
    run(D, nt, nn, cap)

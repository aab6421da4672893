import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mhr_amd
from mhr_amd import ops
from oracle import hstu_oracle as HO
torch.manual_seed(0)
def run(D, n_tok, n_neg, plant):
    g = torch.Generator().manual_seed(5)
    q_rows = torch.randn(n_tok, D, generator=g); p_rows = torch.randn(n_tok, D, generator=g)
    negs = HO.l2n(torch.randn(n_neg, D, generator=g)).bfloat16()
    if plant:
        for t in range(0, n_tok, 5): negs[(t*7) % n_neg] = HO.l2n(p_rows[t][None]).bfloat16()[0]
    idx = torch.arange(n_tok, dtype=torch.int32)
    cap = (n_tok + 3)//4*4
    qi = torch.zeros(cap, dtype=torch.int32); qi[:n_tok] = idx
    ntd = torch.tensor([n_tok], dtype=torch.int32).cuda(); ls = torch.tensor([math.log(20.0)]).cuda()
    sv = ops.nce_fwd(q_rows.cuda(), qi.cuda(), p_rows.cuda(), qi.cuda(), negs.cuda(), ntd, cap, ls, 0.99, want_logs=True)
    w = torch.ones(cap); w[n_tok:] = 0
    dq = torch.zeros(n_tok, D).cuda(); dp = torch.zeros(n_tok, D).cuda()
    dn, dls = ops.nce_bwd(sv, w.cuda(), ls, qi.cuda(), qi.cuda(), dq, dp)
    torch.cuda.synchronize()
    def r(t): return t + (t.bfloat16().float() - t).detach()
    q = q_rows.clone().requires_grad_(True); p = p_rows.clone().requires_grad_(True); nn_ = negs.float().clone().requires_grad_(True)
    qn, pn = r(HO.l2n(q)), r(HO.l2n(p))
    neg = qn @ nn_.T; fix = pn @ nn_.T; keep = ~(fix > 0.99)
    logits = torch.cat([(qn*pn).sum(-1, keepdim=True)*20, torch.where(keep, neg*20, torch.full_like(neg, -1e30))], -1)
    loss = torch.logsumexp(logits, -1) - logits[:, 0]
    loss.sum().backward()
    supp = sv.supp.cpu()
    bits = torch.zeros(n_tok, n_neg, dtype=torch.bool)
    for j in range(n_neg): bits[:, j] = ((supp[j//32, :n_tok] >> (j % 32)) & 1).bool()
    print(f"D={D} n_tok={n_tok} n_neg={n_neg} plant={plant}: supp bits match={bool((bits == ~keep).all())} nsupp={int((~keep).sum())} "
          f"dq err={float((dq.cpu()-q.grad).abs().max()):.4f}/{float(q.grad.abs().max()):.4f} "
          f"dp err={float((dp.cpu()-p.grad).abs().max()):.4f}/{float(p.grad.abs().max()):.4f} "
          f"dn err={float((dn.cpu()-nn_.grad).abs().max()):.4f}/{float(nn_.grad.abs().max()):.4f} loss err={float((sv.loss.cpu()[:n_tok]-loss.detach()).abs().max()):.5f}")
for D, nt, nn, pl in [(16, 37, 30, False), (16, 37, 30, True), (64, 100, 96, False), (256, 130, 512, False), (256, 130, 512, True)]:
    run(D, nt, nn, pl)

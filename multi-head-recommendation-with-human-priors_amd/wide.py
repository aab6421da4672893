"""Generic-width path of the sampled softmax and the catalog decode: feature dims the streaming kernels do not cover.

The row-stationary HIP kernels (csrc/nce.hip, csrc/catalog.hip) keep one operand of the logit GEMM in registers, which
bounds the feature dim at 256 (cfg0 / cfg1).  cfg2 (HSTU size-4, D = 1024) and the HLLM twin (D = 2048) have
GEMMs whose K is large enough for the library GEMM to be the right tool: here the same maths runs as
token-chunked library GEMMs on the GPU (bf16-rounded operands, fp32 products and accumulation: the parity contract) with the epilogues as elementwise passes over one [chunk, n_neg] logit block at
a time - never the reference's full [N_tok, n_neg] tensors.  Same interfaces and saved state as the fused path
(ops.NceSaved), same parity tests; it is selected by ops.nce_fwd / ops.nce_bwd / ops.catalog_topk from the feature dim
alone and is NOT a fallback for a missing library (ops still refuse to run without libmhr_hip.so).

Reference: model/IDNet/hstu.py:600-619 (nce_loss), 697/833 (cross entropy), 965-1015 + trainer.py:724-726 +
collector.py:245 (scores, masks, per-head top-k).
"""
import math

import torch

CHUNK = 8192          # tokens per logit block: 8192 x 8192 fp32 = 256 MB


def _mm(a, b):
    """bf16 x bf16 -> fp32 library GEMM: exact fp32 products of the bf16 operands, fp32 accumulation, fp32 result (MFMA
    rate; an fp32-operand GEMM would run at a fraction of it)."""
    return torch.mm(a, b, out_dtype=torch.float32)


def _scale(logit_scale):
    return torch.exp(logit_scale.detach().float().clamp(0, math.log(100.0))).reshape(())


def _norm_rows(rows, idx):
    x = rows.float()[idx.long().clamp(0, rows.shape[0] - 1)]        # slots past the live count hold arbitrary indices
    inv = 1.0 / x.norm(dim=-1)
    return (x * inv[:, None]).to(torch.bfloat16), inv


def nce_fwd_wide(sv, q_rows, p_rows, negs, logit_scale, want_logs, bucket_idx, loss, n_valid, rank):
    """Fills sv (qn, pn, q_inv, p_inv, s_pos, lse, bucket sums) and loss / n_valid / rank [G, cap] in place."""
    G, cap, thres = sv.groups, sv.cap, sv.thres
    scale = _scale(logit_scale)
    live = torch.arange(cap, device=negs.device)[None, :] < sv.n_tok_dev[:, None]                 # [G, cap]
    for g in range(G):
        qn, qi = _norm_rows(q_rows, sv.q_idx[g])
        pn, pi = _norm_rows(p_rows, sv.p_idx[g])
        sv.qn[g], sv.pn[g], sv.q_inv[g], sv.p_inv[g] = qn, pn, qi, pi
        s_pos = (qn.float() * pn.float()).sum(-1)
        sv.s_pos[g] = s_pos
        ngt = negs[g, :sv.n_neg].t()
        for c0 in range(0, cap, CHUNK):
            sl = slice(c0, min(cap, c0 + CHUNK))
            s = _mm(qn[sl], ngt)                                  # bf16-rounded operands, fp32 products and accumulation
            keep = ~(_mm(pn[sl], ngt) > thres)                    # false-negative suppression
            tot = (torch.exp(scale * (s - 1.0)) * keep).sum(-1) + torch.exp(scale * (s_pos[sl] - 1.0))
            lse = scale + torch.log(tot)
            sv.lse[g, sl] = lse
            loss[g, sl] = torch.where(live[g, sl], lse - scale * s_pos[sl], torch.zeros_like(lse))
            if want_logs:
                n_valid[g, sl] = torch.where(live[g, sl], keep.sum(-1).int() + 1, torch.zeros_like(n_valid[g, sl]))
                rank[g, sl] = torch.where(live[g, sl], (keep & (s > s_pos[sl, None])).sum(-1).int(), torch.zeros_like(rank[g, sl]))
    if bucket_idx is not None:
        flat = (torch.arange(G, device=negs.device)[:, None] * sv.n_buckets + bucket_idx.long().clamp(0, sv.n_buckets - 1)).reshape(-1)
        sv.bucket_sum.view(-1).index_add_(0, flat, loss.reshape(-1))
        sv.bucket_cnt.view(-1).index_add_(0, flat, live.float().reshape(-1))


def nce_bwd_wide(sv, w_tok, logit_scale, dq_rows, dp_rows, d_negs, d_logit_scale):
    """w_tok [G, cap] per-token weights.  Accumulates dq_rows / dp_rows / d_negs / d_logit_scale in place."""
    G, cap, thres = sv.groups, sv.cap, sv.thres
    scale = _scale(logit_scale)
    live = torch.arange(cap, device=dq_rows.device)[None, :] < sv.n_tok_dev[:, None]
    dls = torch.zeros((), dtype=torch.float32, device=dq_rows.device)
    for g in range(G):
        ng = sv.negs[g, :sv.n_neg]
        ngt = ng.t()
        for c0 in range(0, cap, CHUNK):
            sl = slice(c0, min(cap, c0 + CHUNK))
            qn, pn = sv.qn[g, sl], sv.pn[g, sl]
            w = torch.where(live[g, sl], w_tok[g, sl], torch.zeros_like(w_tok[g, sl]))
            s = _mm(qn, ngt)
            keep = ~(_mm(pn, ngt) > thres)
            gmat = (w[:, None] * torch.exp(scale * s - sv.lse[g, sl, None]) * keep)
            gmat = torch.where(live[g, sl, None], gmat, torch.zeros_like(gmat)).to(torch.bfloat16)   # like the fused kernels'
            dq_raw = _mm(gmat, ng)                                                               # sum_j g_ij n_j     # bf16 tile
            if d_negs is not None:
                d_negs[g, :sv.n_neg] += scale * _mm(gmat.t(), qn)
            sp = sv.s_pos[g, sl]
            coef = torch.where(live[g, sl], w * (torch.exp(scale * sp - sv.lse[g, sl]) - 1.0), torch.zeros_like(w))
            qf, pf = qn.float(), pn.float()
            dls = dls + (qf * dq_raw).sum() + (coef * sp).sum()
            dqn = scale * (dq_raw + coef[:, None] * pf)
            dpn = scale * coef[:, None] * qf
            dq = (dqn - qf * (qf * dqn).sum(-1, keepdim=True)) * sv.q_inv[g, sl, None]
            dp = (dpn - pf * (pf * dpn).sum(-1, keepdim=True)) * sv.p_inv[g, sl, None]
            keep_rows = live[g, sl, None]
            qi = torch.where(live[g, sl], sv.q_idx[g, sl].long(), torch.zeros_like(sv.q_idx[g, sl].long()))
            pi = torch.where(live[g, sl], sv.p_idx[g, sl].long(), torch.zeros_like(sv.p_idx[g, sl].long()))
            dq_rows.index_add_(0, qi, torch.where(keep_rows, dq, torch.zeros_like(dq)))
            dp_rows.index_add_(0, pi, torch.where(keep_rows, dp, torch.zeros_like(dp)))
    d_logit_scale += dls * scale


def catalog_topk_wide(users, H, items, n_items, tag_bits, row_bits, hist_ptr, hist_items, k, chunk=65536):
    """Exact per-row top-k (value desc, index asc) by item chunks: GEMM, masks, chunk top-k, running merge."""
    n_rows = users.shape[0]
    dev = users.device
    best_v = torch.full((n_rows, 0), float("-inf"), device=dev)
    best_i = torch.zeros((n_rows, 0), dtype=torch.int64, device=dev)
    rb = row_bits.int()
    hist_mask = None
    if hist_ptr is not None and hist_items is not None and hist_items.numel() > 0:
        lens = (hist_ptr[1:] - hist_ptr[:-1]).long()
        hu = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), lens)
        hist_mask = (hu, hist_items.long())
    for i0 in range(0, n_items, chunk):
        i1 = min(n_items, i0 + chunk)
        sc = _mm(users, items[i0:i1].t())                                                         # [n_rows, chunk], fp32 accumulation
        if tag_bits is not None:
            ok = (tag_bits[i0:i1].int()[None, :] & rb[:, None]) != 0
        else:
            ok = (rb != 0)[:, None].expand(-1, i1 - i0)
        sc = sc.masked_fill(~ok, float("-inf"))
        if i0 == 0:
            sc[:, 0] = float("-inf")                                                              # pad id
        if hist_mask is not None:
            hu, hi = hist_mask
            sel = (hi >= i0) & (hi < i1)
            if bool(sel.any()):
                rows = (hu[sel][:, None] * H + torch.arange(H, device=dev)[None, :]).reshape(-1)
                cols = (hi[sel] - i0)[:, None].expand(-1, H).reshape(-1)
                sc[rows, cols] = float("-inf")
        ids = torch.arange(i0, i1, device=dev)[None, :].expand(n_rows, -1)
        cv = torch.cat([best_v, sc], dim=1)
        ci = torch.cat([best_i, ids], dim=1)
        # value desc, index asc: stable sort by index first, then by value
        order = torch.argsort(ci, dim=1, stable=True)
        cv, ci = torch.gather(cv, 1, order), torch.gather(ci, 1, order)
        order = torch.argsort(cv, dim=1, descending=True, stable=True)[:, :k]
        best_v, best_i = torch.gather(cv, 1, order), torch.gather(ci, 1, order)
    # rows with fewer than k admissible items: complete with (-inf, lowest free ids) like topk_select
    short = ~torch.isfinite(best_v)
    if bool(short.any()):
        for r in torch.nonzero(short.any(dim=1)).flatten().tolist():
            nsel = int(torch.isfinite(best_v[r]).sum())
            taken = set(best_i[r, :nsel].tolist())
            free = [i for i in range(k) if i not in taken][: k - nsel]
            best_i[r, nsel:] = torch.tensor(free, dtype=torch.int64, device=dev)
            best_v[r, nsel:] = float("-inf")
    return best_v, best_i

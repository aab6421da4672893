"""Wide-feature path of the sampled softmax and the catalog decode: feature dims the streaming kernels do not cover.

The row-stationary HIP kernels (csrc/nce.hip, csrc/catalog.hip) keep one operand of the logit GEMM in registers, which
bounds the feature dim at 256 (cfg0 / cfg1).  cfg2 (HSTU size-4, D = 1024) and the HLLM twin (D = 1536 / 2048)
have contractions deep enough for the library GEMM to be the right tool: the logits are produced by
bf16 x bf16 -> fp32 hipBLASLt GEMMs over token / item chunks (exact fp32 products of the bf16-rounded operands, fp32
accumulation: the parity contract of the fused kernels), and everything after the GEMM is ONE hand-written pass over
the fp32 chunk (csrc/wide.hip: loss / lse / log counters; the bf16 softmax-gradient tile; catalog masks + threshold
emit into the candidate lists of the shared exact select) - never the reference's [N_tok, n_neg] / [B, H, N] tensor chains.
Same interfaces and saved state as the fused path (ops.NceSaved), same parity tests; selected by ops.nce_fwd /
ops.nce_bwd / ops.catalog_topk from the feature dim alone and NOT a fallback for a missing library (everything here
calls into libmhr_hip.so).  One host sync per training step: the live-token count bounds the chunk loop (the
reference syncs per prior head, hstu.py:815-816); the streaming path has none.

Reference: model/IDNet/hstu.py:600-629 (nce_loss, logs), 697/833 (cross entropy), 965-1015 + trainer.py:724-726 +
collector.py:245 (scores, masks, per-head top-k).
"""
import math
import os

import torch

from . import lib

_PACKED = None        # (key, packed catalog, packed threshold samples) of the last catalog the wide scorer saw
CHUNK = 8192          # tokens per logit block: 8192 x 8192 fp32 = 256 MB (x2: negatives and false-negative logits)
ITEM_CHUNK = 65536    # items per score block


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _mm(a, b):
    """bf16 x bf16 -> fp32 library GEMM: exact fp32 products of the bf16 operands, fp32 accumulation, fp32 result."""
    return torch.mm(a, b, out_dtype=torch.float32)


def _scale(logit_scale):
    return torch.exp(logit_scale.detach().float().clamp(0, math.log(100.0))).reshape(1).contiguous()


def _norm_rows(rows, idx):
    x = rows.float()[idx.long().clamp(0, rows.shape[0] - 1)]        # slots past the live count hold arbitrary indices
    inv = 1.0 / x.norm(dim=-1).clamp_min(1e-30)                       # (... possibly of all-zero rows: keep them finite)
    return (x * inv[:, None]).to(torch.bfloat16), inv


def _live_cap(sv):
    """Rows worth visiting: the largest live count over the groups, rounded up to 256 (one host sync)."""
    n_max = int(sv.n_tok_dev.max())
    return min(sv.cap, -(-max(n_max, 1) // 256) * 256)


MFMA_NCE = os.environ.get("MHR_NCE_WIDE_MFMA", "1") != "0"     # 0: the library-GEMM + fp32-chunk form of the logit products
# The two PLAIN gradient products of the backward (dQ = G N, dN = G^T Q on the bf16 softmax-gradient tile) run as library GEMMs by
# default: dense products without an epilogue, 4096 - 32768 deep - hipBLASLt does them at 720 TFLOP/s, the own core (mhr_wide_gemm_nt
# on packed operands, MHR_NCE_WIDE_GEMM=own) needs 3x as long once its four operand packs are counted (327 vs 107 us at cfg2's
# shape, tools/nce_wide_micro.py); same bits.
OWN_GEMM = os.environ.get("MHR_NCE_WIDE_GEMM", "lib") == "own"


def _mfma_path(sv, D):
    """The hand-written contraction (csrc/nce_wide.hip) serves every feature dim that is a multiple of 64; REMI's
    hard-negative loss keeps the dense chunk path (its two-pass logsumexp epilogue)."""
    return MFMA_NCE and D % 64 == 0 and D <= 8192 and sv.ihn_beta <= 0


def _pad_rows(x, t_pad):
    """[n] per-token vector -> [t_pad] (the wide kernels read whole 128-token blocks)."""
    if x.shape[0] == t_pad:
        return x.contiguous()
    out = torch.zeros(t_pad, dtype=x.dtype, device=x.device)
    out[:x.shape[0]] = x
    return out


def nce_fwd_wide(sv, q_rows, p_rows, negs, logit_scale, want_logs, bucket_idx, loss, n_valid, rank):
    """Fills sv (qn, pn, q_inv, p_inv, s_pos, lse, bucket sums) and loss / n_valid / rank [G, cap] in place."""
    from . import ops
    G, cap, thres = sv.groups, sv.cap, sv.thres
    scale = _scale(logit_scale)
    sv.scale_dev = scale
    cap_eff = sv.cap_eff = _live_cap(sv)
    loss.zero_()
    sv.lse.zero_()
    if want_logs:
        n_valid.zero_()
        rank.zero_()
    st = _stream()
    D = q_rows.shape[1]
    mfma = _mfma_path(sv, D)
    sv.wide_pack = [None] * G if mfma else None
    dev = negs.device
    for g in range(G):
        qn, qi = _norm_rows(q_rows, sv.q_idx[g, :cap_eff])
        pn, pi = _norm_rows(p_rows, sv.p_idx[g, :cap_eff])
        sv.qn[g, :cap_eff], sv.pn[g, :cap_eff], sv.q_inv[g, :cap_eff], sv.p_inv[g, :cap_eff] = qn, pn, qi, pi
        s_pos = (qn.float() * pn.float()).sum(-1).contiguous()
        sv.s_pos[g, :cap_eff] = s_pos
        if mfma:
            # hand-written contraction: targets x negatives -> false-negative bits, queries x negatives -> the per-token sums
            # (one launch each on the packed tile images; no logit block, no library GEMM)
            n_neg = sv.n_neg
            t_pad = -(-cap_eff // 128) * 128
            n_tiles = -(-n_neg // 256) * 8
            negs_p = ops.pack_tiles(negs[g].contiguous(), n_sel=n_neg, tiles_per_block=8)
            q_p = ops.pack_tiles(qn.contiguous(), tiles_per_block=4)
            p_p = ops.pack_tiles(pn.contiguous(), tiles_per_block=4)
            bits = torch.empty(n_tiles * 2, t_pad, dtype=torch.int16, device=dev)
            lib.call("mhr_nce_wide_fix_bits", p_p.data_ptr(), cap_eff, negs_p.data_ptr(), n_neg, D, float(thres), bits.data_ptr(), st)
            n_lists = 4 * lib.load().mhr_catalog_wide_slices(cap_eff)
            part = torch.empty(3, n_lists, t_pad, dtype=torch.float32, device=dev)
            sp_pad = _pad_rows(s_pos, t_pad)
            lse_g, loss_g = torch.empty(t_pad, dtype=torch.float32, device=dev), torch.empty(t_pad, dtype=torch.float32, device=dev)
            nv_g = torch.empty(t_pad, dtype=torch.int32, device=dev) if want_logs else None
            rk_g = torch.empty(t_pad, dtype=torch.int32, device=dev) if want_logs else None
            ops._timed_call("mhr_nce_wide_fwd", q_p.data_ptr(), cap_eff, negs_p.data_ptr(), n_neg, D, bits.data_ptr(), sp_pad.data_ptr(),
                            scale.data_ptr(), sv.n_tok_dev[g:g + 1].data_ptr(), part[0].data_ptr(), part[1].view(torch.int32).data_ptr(),
                            part[2].view(torch.int32).data_ptr(), lse_g.data_ptr(), loss_g.data_ptr(), ops._ptr(nv_g), ops._ptr(rk_g), st)
            sv.lse[g, :cap_eff], loss[g, :cap_eff] = lse_g[:cap_eff], loss_g[:cap_eff]
            if want_logs:
                n_valid[g, :cap_eff], rank[g, :cap_eff] = nv_g[:cap_eff], rk_g[:cap_eff]
            sv.wide_pack[g] = (negs_p, bits, t_pad)
            continue
        ngt = negs[g, :sv.n_neg].t()
        for c0 in range(0, cap_eff, CHUNK):
            c1 = min(cap_eff, c0 + CHUNK)
            s = _mm(qn[c0:c1], ngt)                               # cos(query, negative)
            fx = _mm(pn[c0:c1], ngt)                              # cos(target, negative): false-negative test
            if sv.ihn_beta > 0:                                   # REMI's hard-negative loss: same chunks, its own epilogue
                lib.call("mhr_ihn_dense_fwd", s.data_ptr(), fx.data_ptr(), s.shape[1], sv.n_neg, s_pos[c0:c1].data_ptr(),
                         scale.data_ptr(), float(thres), float(sv.ihn_beta), sv.n_tok_dev[g:g + 1].data_ptr(), c0, c1 - c0,
                         sv.lse[g, c0:c1].data_ptr(), sv.ihn_num[g, c0:c1].data_ptr(), sv.ihn_imp[g, c0:c1].data_ptr(),
                         loss[g, c0:c1].data_ptr(), n_valid[g, c0:c1].data_ptr() if want_logs else 0,
                         rank[g, c0:c1].data_ptr() if want_logs else 0, st)
                continue
            lib.call("mhr_nce_dense_fwd", s.data_ptr(), fx.data_ptr(), s.shape[1], sv.n_neg, s_pos[c0:c1].data_ptr(),
                     scale.data_ptr(), float(thres), sv.n_tok_dev[g:g + 1].data_ptr(), c0, c1 - c0, sv.lse[g, c0:c1].data_ptr(),
                     loss[g, c0:c1].data_ptr(), n_valid[g, c0:c1].data_ptr() if want_logs else 0,
                     rank[g, c0:c1].data_ptr() if want_logs else 0, st)
    if bucket_idx is not None:
        live = torch.arange(cap, device=negs.device)[None, :] < sv.n_tok_dev[:, None]
        flat = (torch.arange(G, device=negs.device)[:, None] * sv.n_buckets + bucket_idx.long().clamp(0, sv.n_buckets - 1)).reshape(-1)
        sv.bucket_sum.view(-1).index_add_(0, flat, loss.reshape(-1))
        sv.bucket_cnt.view(-1).index_add_(0, flat, live.float().reshape(-1))


def nce_bwd_wide(sv, w_tok, logit_scale, dq_rows, dp_rows, d_negs, d_logit_scale):
    """w_tok [G, cap] per-token weights.  Accumulates dq_rows / dp_rows / d_negs / d_logit_scale in place."""
    G, thres = sv.groups, sv.thres
    scale = sv.scale_dev
    cap_eff = sv.cap_eff
    dev = dq_rows.device
    live = torch.arange(cap_eff, device=dev)[None, :] < sv.n_tok_dev[:, None]
    dls = torch.zeros((), dtype=torch.float32, device=dev)
    st = _stream()
    w_tok = w_tok.contiguous()
    from . import ops
    D = sv.qn.shape[-1]
    for g in range(G):
        ng = sv.negs[g, :sv.n_neg]
        ngt = ng.t()
        pack = sv.wide_pack[g] if getattr(sv, "wide_pack", None) is not None else None
        step = cap_eff if pack is not None else CHUNK          # the hand-written tile producer covers all live rows in one launch
        for c0 in range(0, cap_eff, step):
            c1 = min(cap_eff, c0 + step)
            qn, pn = sv.qn[g, c0:c1], sv.pn[g, c0:c1]
            if pack is not None:
                # softmax-gradient tile straight from the MFMA accumulators (recomputed cosines, saved suppression bits)
                negs_p, bits, t_pad = pack
                ldg = -(-sv.n_neg // 64) * 64            # (whole 64-entry chunks: G is the K-contiguous operand of dQ = G N)
                gfull = torch.empty(cap_eff, ldg, dtype=torch.bfloat16, device=dev)
                q_p = ops.pack_tiles(qn.contiguous(), tiles_per_block=4)
                lse_pad, w_pad = _pad_rows(sv.lse[g, :cap_eff], t_pad), _pad_rows(w_tok[g, :cap_eff], t_pad)   # (named: alive across the call)
                ops._timed_call("mhr_nce_wide_grad_tile", q_p.data_ptr(), cap_eff, negs_p.data_ptr(), sv.n_neg, D, bits.data_ptr(),
                                lse_pad.data_ptr(), w_pad.data_ptr(), scale.data_ptr(), sv.n_tok_dev[g:g + 1].data_ptr(),
                                gfull.data_ptr(), ldg, st)
                gmat = gfull[:, :sv.n_neg]
            else:
                s = _mm(qn, ngt)
                fx = _mm(pn, ngt)
                gmat = torch.empty(c1 - c0, sv.n_neg, dtype=torch.bfloat16, device=dev)
            if pack is not None:
                pass
            elif sv.ihn_beta > 0:
                lib.call("mhr_ihn_dense_bwd", s.data_ptr(), fx.data_ptr(), s.shape[1], sv.n_neg, sv.lse[g, c0:c1].data_ptr(),
                         sv.ihn_num[g, c0:c1].data_ptr(), sv.ihn_imp[g, c0:c1].data_ptr(), w_tok[g, c0:c1].data_ptr(),
                         scale.data_ptr(), float(thres), float(sv.ihn_beta), sv.n_tok_dev[g:g + 1].data_ptr(), c0, c1 - c0,
                         gmat.data_ptr(), sv.n_neg, st)
            else:
                lib.call("mhr_nce_dense_bwd", s.data_ptr(), fx.data_ptr(), s.shape[1], sv.n_neg, sv.lse[g, c0:c1].data_ptr(),
                         w_tok[g, c0:c1].data_ptr(), scale.data_ptr(), float(thres), sv.n_tok_dev[g:g + 1].data_ptr(), c0, c1 - c0,
                         gmat.data_ptr(), sv.n_neg, st)
            if pack is None:
                del s, fx
            if pack is not None and OWN_GEMM:
                # the two plain products on the same LDS-tiled core (mhr_wide_gemm_nt): out[r, i] = sum_k A[i, k] B[r, k]
                #   dQ_raw [tok, d] = sum_n G[tok, n] N[n, d]:  A = N^T (packed transposed), B = G
                #   dN     [n, d]   = sum_t G[t, n] Q[t, d]:    A = Q^T, B = G^T (both packed transposed), alpha = scale, accumulated
                nt_p, k1 = ops.pack_tiles_t(ng.contiguous(), n_sel=D, tiles_per_block=8)
                assert k1 == ldg
                g_p = ops.pack_tiles(gfull, tiles_per_block=4)
                dq_raw = torch.empty(cap_eff, D, dtype=torch.float32, device=dev)
                ops._timed_call("mhr_wide_gemm_nt", nt_p.data_ptr(), D, g_p.data_ptr(), cap_eff, ldg, 0, dq_raw.data_ptr(), D, 0, st)
                if d_negs is not None:
                    qt_p, k2 = ops.pack_tiles_t(qn.contiguous(), n_sel=D, tiles_per_block=8)
                    gt_p, k2b = ops.pack_tiles_t(gfull, n_sel=sv.n_neg, tiles_per_block=4)
                    assert k2 == k2b
                    dn_g = d_negs[g, :sv.n_neg]
                    assert dn_g.is_contiguous()
                    ops._timed_call("mhr_wide_gemm_nt", qt_p.data_ptr(), D, gt_p.data_ptr(), sv.n_neg, k2, scale.data_ptr(), dn_g.data_ptr(),
                                    D, 1, st)
            else:
                dq_raw = _mm(gmat, ng)                                                           # sum_j g_ij n_j
                if d_negs is not None:
                    d_negs[g, :sv.n_neg] += scale * _mm(gmat.t(), qn)
            lv = live[g, c0:c1]
            w = torch.where(lv, w_tok[g, c0:c1], torch.zeros_like(w_tok[g, c0:c1]))
            sp = sv.s_pos[g, c0:c1]
            coef = torch.where(lv, w * (torch.exp(scale * sp - sv.lse[g, c0:c1]) - 1.0), torch.zeros_like(w))
            qf, pf = qn.float(), pn.float()
            dls = dls + (qf * dq_raw).sum() + (coef * sp).sum()
            dqn = scale * (dq_raw + coef[:, None] * pf)
            dpn = scale * coef[:, None] * qf
            dq = (dqn - qf * (qf * dqn).sum(-1, keepdim=True)) * sv.q_inv[g, c0:c1, None]
            dp = (dpn - pf * (pf * dpn).sum(-1, keepdim=True)) * sv.p_inv[g, c0:c1, None]
            zero = torch.zeros((), dtype=torch.int64, device=dev)
            qi = torch.where(lv, sv.q_idx[g, c0:c1].long(), zero)
            pi = torch.where(lv, sv.p_idx[g, c0:c1].long(), zero)
            dq_rows.index_add_(0, qi, torch.where(lv[:, None], dq, torch.zeros_like(dq)))
            dp_rows.index_add_(0, pi, torch.where(lv[:, None], dp, torch.zeros_like(dp)))
    d_logit_scale += dls * scale.reshape(())


# ------------------------------------------------------------------------------------------------
# catalog decode
# ------------------------------------------------------------------------------------------------
def _exact_rows(users, H, items, n_items, tag_bits, row_bits, hist_ptr, hist_items, k, chunk):
    """Exact per-row top-k (value desc, index asc) with every score kept: the dense row scorer (csrc/catalog_dense.hip: fp32
    accumulation of the operands as given, masks in place) + the exact select over the whole row.  Small catalogs and the
    rows the threshold pass could not certify.  hist_ptr / hist_items: the CSR history of THESE users (rows // H)."""
    from . import ops
    rows = torch.arange(users.shape[0], dtype=torch.int32, device=users.device)
    return ops.dense_rows_topk(users.contiguous(), H, items if items.is_contiguous() else items.contiguous(), n_items, tag_bits,
                               row_bits, hist_ptr, hist_items, rows, k)


def _catalog_topk_mfma(users, H, items, N, tag_bits, row_bits, hist_ptr, hist_items, k, chunk, target, stats, k_min=None, tau_out=None,
                       margin=None):
    """Exact per-row top-k with the hand-written wide scorer (csrc/catalog_wide.hip: LDS-tiled MFMA GEMM, threshold emit in
    the epilogue - no score block in memory, no library GEMM).  Same scheme as the register-stationary path
    (ops.catalog_topk): thresholds from two strided sample passes through the SAME kernel, one full pass, the shared exact
    select; rows that cannot be certified (list overflow, too few candidates) are re-run keeping every score."""
    from . import lib, ops
    n_rows, D = users.shape
    dev = users.device
    users, items = users.contiguous(), items.contiguous()
    tb = None if tag_bits is None else tag_bits
    ninf = torch.full((n_rows,), float("-inf"), dtype=torch.float32, device=dev)
    users_p = ops.pack_tiles(users, tiles_per_block=4)                              # once per batch (a few MB)
    SEL = 8192                                                                    # candidates the select kernel holds per row
    k_min = k if k_min is None else min(k, k_min)
    if target is None:
        target = max(512, int(2.5 * k_min))
    s1 = max(1, -(-N // 2048))
    # second sample: dense enough that the threshold is the ~50th largest of the sample (its rank estimate then scatters by
    # ~14 %: a row comes up short of k candidates at > 4 sigma; at N = 2^20 a 1 / 32 sample left 2.4 sigma and flagged rows -
    # each an exact re-run - in most batches)
    s2 = max(1, min(-(-N // 32768), target // 48))
    # the catalog's packed images (whole, and the two strided samples of the threshold passes) depend on the table only:
    # built once per cached catalog (the evaluation normalises and caches it, REC/model/multihead.py:_normalised_items)
    # (keyed on the tensor OBJECT, which the cache keeps alive: an address alone may be a freed table's, reused)
    global _PACKED
    if _PACKED is None or _PACKED[0] is not items or _PACKED[1] != (items._version, N, s1, s2):
        full = ops.pack_tiles(items, n_sel=N)
        samp = None if N <= SEL else (ops.pack_tiles(items[:N], row_stride=s1), ops.pack_tiles(items[:N], row_stride=s2))
        _PACKED = (items, (items._version, N, s1, s2), full, samp)
    items_p, samples = _PACKED[2], _PACKED[3]
    if N <= SEL:                                                                   # small catalogs: every score is a candidate
        cand = ops.catalog_emit_wide(users_p, n_rows, D, items_p, N, tb, row_bits, ninf, 64)     # >= 32 slices: one block per slice at most
        ov, oi, _, _, _ = ops.topk_select_sliced(cand, H, hist_ptr, hist_items, k)
        if tau_out is not None:
            tau_out["tau"] = ninf
        return ov, oi
    t1 = min(1024, max(8, -(-3 * target // s1)))           # first threshold: about 3x looser than the rank aimed at (a tighter one
                                                   # starves the second sample and, as the fallback threshold, the candidates)
    t2 = min(max(k_min // s2 + 1, target // s2), 1024)             # (the select kernel picks at most 1024)
    # pass 1: every s1-th item, all scores (<= 2048 per row, <= 64 per list) -> the t1-th largest bounds the top ~0.4 %
    c1 = ops.catalog_emit_wide(users_p, n_rows, D, samples[0], N, tb, row_bits, ninf, 64, 0, s1)
    _, _, kth1, _, st1 = ops.topk_select_sliced(c1, H, hist_ptr, hist_items, t1)
    # pass 2: every s2-th item above kth1 -> the t2-th largest estimates the score of rank ~target
    c2 = ops.catalog_emit_wide(users_p, n_rows, D, samples[1], N, tb, row_bits, kth1, 32, 0, s2)
    ov2, _, kth2, _, st2 = ops.topk_select_sliced(c2, H, hist_ptr, hist_items, t2)
    ok2 = (st2 == 0) & (st1 == 0)
    tau = torch.where(torch.isfinite(kth2) & ok2, kth2, torch.where(ok2, kth1, ninf)).contiguous()
    if margin is not None:
        # the caller re-ranks everything within `margin` of the k_min-th score (ops.catalog_topk_exact): the threshold must lie
        # below that band, however many items it holds - with (near-)random embeddings at wide feature dims the cosines
        # concentrate (std 1 / sqrt(D)) and the band holds several hundred.  The k_min-th score is estimated from the second
        # sample (rank k_min / s2, taken two standard deviations of its rank noise further down).
        r_k = k_min // s2 + 1
        r_lo = min(t2 - 1, r_k + int(2.0 * r_k ** 0.5) + 1)
        est = ov2[:, r_lo]
        tau_m = torch.where(torch.isfinite(est) & ok2, est - margin, tau)
        tau = torch.minimum(tau, tau_m).contiguous()
        target = max(target, 4096)                   # list capacity for the band (candidates per row the lists can hold)
    n_sl = lib.load().mhr_catalog_wide_slices(n_rows)
    cap_s = max(32, 4 * -(-target // (4 * n_sl)) + 16)
    cand = ops.catalog_emit_wide(users_p, n_rows, D, items_p, N, tb, row_bits, tau, cap_s)
    ov, oi, _, got, stt = ops.topk_select_sliced(cand, H, hist_ptr, hist_items, k)
    flagged = (stt != 0) | ((got < k_min) & (row_bits != 0) & torch.isfinite(tau))
    if stats is not None:
        stats["mean_candidates"] = float(got.float().mean())
        stats["flagged_rows"] = int(flagged.sum())
    if bool(flagged.any()):                               # one host sync per batch; results go to the host anyway
        users_f = torch.nonzero(flagged.view(-1, H).any(dim=1)).flatten()
        rows_f = (users_f[:, None] * H + torch.arange(H, device=dev)[None, :]).flatten()
        sub_ptr, sub_items = ops.sub_history(hist_ptr, hist_items, users_f)
        fv, fi = _exact_rows(users[rows_f].contiguous(), H, items, N, tag_bits, row_bits[rows_f].contiguous(), sub_ptr, sub_items,
                             k, chunk)
        ov[rows_f] = fv
        oi[rows_f] = fi
        tau = tau.clone()
        tau[rows_f] = float("-inf")
    if tau_out is not None:
        tau_out["tau"] = tau
    return ov, oi


def catalog_topk_wide(users, H, items, n_items, tag_bits, row_bits, hist_ptr, hist_items, k, chunk=ITEM_CHUNK, target=None,
                      stats=None, k_min=None, tau_out=None, margin=None):
    """Exact per-row top-k over the catalog at any feature dim.  Thresholds from a strided sample of the catalog; the
    full pass scores item chunks with the library GEMM and emits the few scores above the threshold (csrc/wide.hip) into
    candidate lists; the exact select of the streaming path (topk_select_sliced) picks the top k with the history filter.
    Rows it cannot certify (too few candidates, list overflow) are re-run keeping every score - sampling only affects speed."""
    from . import ops
    n_rows, D = users.shape
    N = int(n_items)
    dev = users.device
    if D % 64 == 0 and users.dtype == torch.bfloat16 and items.dtype == torch.bfloat16:
        return _catalog_topk_mfma(users, H, items, N, tag_bits, row_bits, hist_ptr, hist_items, k, chunk, target, stats, k_min, tau_out,
                                  margin)
    if N <= max(4 * k, 2048) or N <= chunk // 8:
        return _exact_rows(users, H, items, N, tag_bits, row_bits, hist_ptr, hist_items, k, chunk)
    st = _stream()
    tagp = 0 if tag_bits is None else tag_bits.data_ptr()
    if target is None:
        target = max(512, int(2.5 * k))
    # threshold: the score of rank ~target, estimated on every s-th item
    s = max(1, N // 32768)
    t = max(k // s + 1, target // s)
    sample = items[0:N:s].contiguous()
    sc = _mm(users, sample.t())
    lib.call("mhr_catalog_mask_dense", sc.data_ptr(), sc.shape[1], sample.shape[0], 0, s, tagp, row_bits.data_ptr(), n_rows, st)
    tau = torch.topk(sc, min(t, sc.shape[1]), dim=1).values[:, -1].contiguous()                  # -inf: too few admissible -> exact
    del sc
    n_chunks = -(-N // chunk)
    per = 8
    seg = -(-chunk // per)
    n_lists = n_chunks * per
    cap_s = max(32, 4 * -(-target // n_lists) + 16)
    val = torch.empty(n_rows, n_lists, cap_s, dtype=torch.float32, device=dev)
    idx = torch.empty(n_rows, n_lists, cap_s, dtype=torch.int32, device=dev)
    cnt = torch.zeros(n_rows, n_lists, dtype=torch.int32, device=dev)
    for ci, i0 in enumerate(range(0, N, chunk)):
        i1 = min(N, i0 + chunk)
        sc = _mm(users, items[i0:i1].t())
        lib.call("mhr_catalog_emit_dense", sc.data_ptr(), sc.shape[1], i1 - i0, seg, i0, tagp, row_bits.data_ptr(), tau.data_ptr(),
                 n_rows, val.data_ptr(), idx.data_ptr(), cnt.data_ptr(), n_lists, ci * per, cap_s, st)
        del sc
    ov, oi, _, got, stt = ops.topk_select_sliced((val, idx, cnt, n_lists), H, hist_ptr, hist_items, k)
    flagged = (stt != 0) | ((got < k) & (row_bits != 0) & torch.isfinite(tau))
    if stats is not None:
        stats["mean_candidates"] = float(got.float().mean())
        stats["flagged_rows"] = int(flagged.sum())
    if bool(flagged.any()):                               # one host sync per batch; results go to the host anyway
        users_f = torch.nonzero(flagged.view(-1, H).any(dim=1)).flatten()
        rows_f = (users_f[:, None] * H + torch.arange(H, device=dev)[None, :]).flatten()
        sub_ptr, sub_items = ops.sub_history(hist_ptr, hist_items, users_f)
        fv, fi = _exact_rows(users[rows_f].contiguous(), H, items, N, tag_bits, row_bits[rows_f].contiguous(), sub_ptr, sub_items,
                             k, chunk)
        ov[rows_f] = fv
        oi[rows_f] = fi
    return ov, oi

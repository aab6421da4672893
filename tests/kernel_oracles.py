"""Kernel-level oracles of the GPU tests: the PINNED oracle functions (oracle/hstu_oracle.py: `hstu_attention`,
`nce_logits`, `token_ce` - each checked against the reference's own outputs in tests/test_oracle_golden.py) evaluated on
the operand precision of the kernels, i.e. with bf16 straight-through rounding of what the kernels feed to the matrix cores
(`operand_round`).  There is no second formula here: with operand_round = None these ARE the fixture-pinned functions
(tests/test_cpu_host.py::test_kernel_oracles_are_the_pinned_oracle_functions)."""
import torch

from oracle import hstu_oracle as HO


def bf16_round(t):
    """Straight-through bf16 rounding (value rounded, gradient passed)."""
    return t + (t.to(torch.bfloat16).float() - t).detach()


def attn_oracle(h, valid, B, L, Hh, hd, d_out=None, operand_round=bf16_round):
    """h [B*L, 4D] pre-activation u|v|q|k (bf16 values): -> (attention output [B*L, D], d/dh or None).
    reference model/IDNet/hstu.py:244-245 (SiLU of the uvqk product) + 137-160 (attention)."""
    D = Hh * hd
    hp = h.float().clone().requires_grad_(d_out is not None)
    u, v, q, k = torch.split(HO.silu(hp), [D, D, D, D], dim=-1)
    rnd = operand_round if operand_round is not None else (lambda t: t)
    q, k, v = rnd(q).view(B, L, D), rnd(k).view(B, L, D), rnd(v).view(B, L, D)
    out = HO.hstu_attention(q, k, v, valid, Hh, operand_round=operand_round).reshape(B * L, D)
    if d_out is None:
        return out.detach(), None
    out.backward(d_out.float())
    return out.detach(), hp.grad


def nce_oracle(q, p, negs_n, ls, thres, operand_round=bf16_round):
    """-> (per-token loss, logits [T, 1 + Nn], keep [T, Nn] bool, neg cosines (suppressed: -inf), pos cosines [T, 1]).
    reference model/IDNet/hstu.py:600-619 + F.cross_entropy (697)."""
    import math
    logits = HO.nce_logits(q, p, negs_n, ls, thres, operand_round=operand_round)
    loss = HO.token_ce(logits)
    scale = torch.clamp(ls.detach(), 0.0, math.log(100.0)).exp()
    keep = logits[:, 1:].detach() > HO.F32_MIN / 100            # suppressed entries: F32_MIN * scale (-inf once scale > 1)
    cos = logits.detach() / scale
    return loss, logits, keep, cos[:, 1:], cos[:, :1]

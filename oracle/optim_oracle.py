"""CPU oracle of the optimizer step and LR schedule (test infrastructure only).

Reference: `code/REC/trainer/trainer.py:292-299` builds DeepSpeed FusedAdam
(adam_w_mode, betas (0.9, 0.999), eps 1e-8) or torch AdamW with the same
defaults, dense over every parameter including the item table;
`code/REC/utils/lr_scheduler.py:79-116` is the cosine schedule with linear
warm-up (warm-up steps = total_iters * scheduler_args.warmup, trainer.py:457-459).
"""
import math

import torch


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One decoupled-weight-decay Adam step, in place on (p, m, v); `step` is 1-based."""
    if weight_decay != 0.0:
        p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p, m, v


def cosine_warmup_factor(step, warmup_steps, total_steps, num_cycles=0.5):
    """lr multiplier at scheduler step `step` (0-based count of scheduler.step() calls)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    prog = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog)))


def segment_sum_rows(ids, rows, n_rows):
    """Dense [n_rows, D] gradient of an embedding gather: sum `rows` by `ids` (torch index_add)."""
    out = torch.zeros(n_rows, rows.shape[-1], dtype=torch.float32)
    out.index_add_(0, ids.reshape(-1), rows.reshape(-1, rows.shape[-1]).float())
    return out

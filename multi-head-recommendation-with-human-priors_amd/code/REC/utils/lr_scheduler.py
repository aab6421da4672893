"""Cosine schedule with linear warm-up (reference `code/REC/utils/lr_scheduler.py:79-116`)."""
import math

from torch.optim.lr_scheduler import LambdaLR


def cosine_warmup_factor(step, num_warmup_steps, num_training_steps, num_cycles=0.5):
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5, last_epoch=-1):
    return LambdaLR(optimizer, lambda s: cosine_warmup_factor(s, num_warmup_steps, num_training_steps, num_cycles), last_epoch)

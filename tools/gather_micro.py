"""Embedding gather bandwidth at the train-step size and at a size that fills the chip (HIP events, algorithmic bytes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mhr_amd
from mhr_amd import ops
N, D = 453938, 256
table = torch.randn(N, D, device="cuda")
pos = torch.randn(201, D, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
def call(ids, with_x):
    if with_x:
        return ops.embedding_gather(table, ids.view(-1, 208), torch.float32, pos, 200, torch.float32)
    return ops.embedding_gather(table, ids, torch.float32)


for n_ids, with_x in ((26624, True), (59392, False), (1 << 20, False), (1 << 22, False)):
    reps = 20
    idl = [torch.randint(1, N, (n_ids,), device="cuda", generator=g) for _ in range(reps)]     # fresh rows every launch:
    ids = idl[0]                                                                                # the 465 MB table does not fit the 256 MB MALL
    for _ in range(3):
        out = call(ids, with_x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r_ in range(reps):
        out = call(idl[r_], with_x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    nbytes = n_ids * D * 8 + (n_ids // 208 * 200 * D * 4 if with_x else 0)
    print(f"n_ids {n_ids:8d} fused_pos {with_x}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s  ({nbytes / us / 1e3 / 8000:.1%} of 8 TB/s)")

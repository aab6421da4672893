"""Data-parallel plumbing over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

One process per GPU; users (batch rows) are independent, so the only exchanges of a train step are
  (1) an all-gather of negative item *ids* (model side, REC/model/basemodel.py:all_gather_ids),
  (2) one all-reduce of the flat dense-parameter gradient bucket (a few MB), and
  (3) the sparse item-table gradient: the shared-negative block is identically indexed on every rank, so it is a
      plain all-reduce with no indices; the rank-private positive rows travel as one all-gather of (ids, rows).
The table update itself stays dense and local (Adam moments of untouched rows still decay - reference semantics,
trainer.py:292-299), but only touched rows cross xGMI: at cfg1/W=8 about 60 MB per rank instead of the reference's
465 MB dense table gradient.  xGMI is point-to-point (7 links per GPU), so few large collectives are preferred
over many small ones: each of (2) and (3) is a single call.
"""
import os

import torch
import torch.distributed as dist

# dtype of the rank-private gradient rows on the wire.  bf16 halves the step's largest collective (W x 26 624 rows x D at
# cfg1); the per-id sum over ranks is still accumulated in fp32 by the segment-sum kernel.  The reference's DeepSpeed
# bf16 runs reduce the whole gradient in bf16 (trainer.py:292-299, ds_config bf16); MHR_DP_WIRE=fp32 keeps fp32 rows.
ROWS_WIRE_DTYPE = torch.float32 if os.environ.get("MHR_DP_WIRE", "bf16") == "fp32" else torch.bfloat16


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean_(flat):
    """In-place mean all-reduce of one flat bucket."""
    if world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world_size())
    return flat


def exchange_sparse_rows(ids_all, d_rows, n_private, wire_dtype=None):
    """Combine the per-rank embedding-gradient rows of a step.

    ids_all [R] int64 / d_rows [R, D] fp32: the first `n_private` entries are rank-private (positives: every rank has
    its own users), the rest are the shared negatives (same ids, same order on every rank after the id all-gather).
    Returns (ids [W*n_private + n_shared], private rows [W*n_private, D] in the wire dtype, shared rows [n_shared, D] fp32
    summed over ranks); the caller sums duplicates in fp32 and scales by 1/W (DDP's gradient mean) inside the fused Adam.
    """
    W = world_size()
    wire = ROWS_WIRE_DTYPE if wire_dtype is None else wire_dtype
    if W == 1:
        return ids_all, d_rows[:n_private], d_rows[n_private:]
    priv_ids, priv_rows = ids_all[:n_private].contiguous(), d_rows[:n_private].to(wire).contiguous()
    shared_ids, shared_rows = ids_all[n_private:], d_rows[n_private:].contiguous()
    if shared_rows.numel():
        dist.all_reduce(shared_rows, op=dist.ReduceOp.SUM)
    n_sh = shared_ids.numel()
    ids_out = torch.empty(W * n_private + n_sh, dtype=ids_all.dtype, device=ids_all.device)
    rows_priv = torch.empty(W * n_private, d_rows.shape[1], dtype=wire, device=d_rows.device)
    ids_out[W * n_private:] = shared_ids
    if dist.get_backend() == "nccl":          # gather straight into the output (no per-rank list + concatenation)
        dist.all_gather_into_tensor(ids_out[:W * n_private], priv_ids)
        dist.all_gather_into_tensor(rows_priv, priv_rows)
    else:
        g_ids = [torch.empty_like(priv_ids) for _ in range(W)]
        g_rows = [torch.empty_like(priv_rows) for _ in range(W)]
        dist.all_gather(g_ids, priv_ids)
        dist.all_gather(g_rows, priv_rows)
        ids_out[:W * n_private] = torch.cat(g_ids)
        rows_priv.copy_(torch.cat(g_rows))
    return ids_out, rows_priv, shared_rows


def allreduce_metric_sums(values):
    """One packed all-reduce for all metric sums of an evaluation (the reference issues one per key,
    trainer.py:1060-1066, 1110-1115).  `values`: 1-D float64/float32 tensor on the collective's device."""
    if world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values

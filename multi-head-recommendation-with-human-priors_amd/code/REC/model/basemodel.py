"""Base class and collectives shared by the models (reference `code/REC/model/basemodel.py`)."""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_gather(data, group=None, sync_grads=False):
    """Reference semantics (basemodel.py:11-22): stack of every rank's tensor along a new dim 0, differentiable
    when sync_grads.  The fused HSTU path does not use it (it shares negative *ids*, see all_gather_ids)."""
    if _world() > 1:
        from torch.distributed import nn as dnn
        if sync_grads:
            return torch.stack(dnn.functional.all_gather(data, group=group), dim=0)
        with torch.no_grad():
            return torch.stack(dnn.functional.all_gather(data, group=group), dim=0)
    return data.unsqueeze(0)


def all_gather_ids(ids, group=None):
    """[W, *ids.shape] int64: the negatives of every rank, in rank order.  With a replicated item table, gathering
    8-byte ids and re-gathering rows locally is numerically identical to the reference's autograd all_gather of
    fp32 embeddings (hstu.py:673, 755) - the negatives' gradient rows are then summed by the sparse embedding
    reduction - and moves D*4/8 times fewer bytes over xGMI with no backward collective."""
    from mhr_amd import distributed as D_
    if D_.active():
        ids = ids.contiguous()
        out = torch.empty((_world(),) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
        dst, src = out.view(-1), ids.view(-1)
        # straight into the stacked output (RCCL and gloo alike); under the Trainer's step capture a host call of the replay plan
        D_.run_collective(lambda: dist.all_gather_into_tensor(dst, src, group=group))
        return out
    return ids.unsqueeze(0)


def all_gather_pool_ids(neg_items, pools, group=None, extra=None):
    """neg_items [B, n_pools, n_neg] int64, `pools` = the pool indices the loss reads: ONE id all-gather for all of
    them (the reference gathers the embeddings of each pool separately inside its per-category loop,
    hstu.py:669-673, 751-755).  Returns a list of [W*B*n_neg] id vectors, rank-major like all_gather_ids(pool).
    extra (optional int64 vector, e.g. the rank's own item-window ids): rides the same collective; returns (list, [W, extra.numel()]) -
    every rank then knows the ids of ALL ranks' gradient rows in the forward already, so the sparse-row reduction's id sort can run
    early and the backward's exchange moves rows only."""
    # (no `neg_items[:, list]`: indexing with a Python list uploads an index tensor - a blocking copy, 2.3 ms of host stall per step)
    sel = torch.stack([neg_items[:, p] for p in pools], dim=1)                         # [B, G, n_neg]
    if extra is None:
        g = all_gather_ids(sel, group=group)                                           # [W, B, G, n_neg]
        g = g.permute(2, 0, 1, 3).reshape(len(pools), -1)
        return [g[i] for i in range(len(pools))]
    n_sel = sel.numel()
    both = all_gather_ids(torch.cat([sel.reshape(-1), extra.reshape(-1)]), group=group)   # [W, n_sel + n_extra]
    g = both[:, :n_sel].reshape((both.shape[0],) + tuple(sel.shape)).permute(2, 0, 1, 3).reshape(len(pools), -1)
    return [g[i] for i in range(len(pools))], both[:, n_sel:]


def l2_norm(x, eps=1e-6):
    return x / torch.clamp(torch.linalg.norm(x, ord=2, dim=-1, keepdim=True), min=eps)


class BaseModel(nn.Module):
    @property
    def module(self):
        """The reference's trainer reaches the raw model through the DDP/DeepSpeed wrapper's `.module`."""
        return self

    def load_weights(self, path):
        state = torch.load(path, map_location='cpu', weights_only=True)
        state = state.get('state_dict', state)
        self.load_state_dict(state, strict=False)

    def __str__(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + f'\nTrainable parameters: {n}'

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
    if _world() > 1:
        out = [torch.empty_like(ids) for _ in range(_world())]
        dist.all_gather(out, ids.contiguous(), group=group)
        return torch.stack(out, dim=0)
    return ids.unsqueeze(0)


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

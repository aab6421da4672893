"""Multi-head decode collector (reference `code/REC/evaluator/collector.py:58-395`), fused.

`eval_batch_collect` accepts either the reference's dense `scores [B,H,N]` tensor (on the GPU) or the `FusedTopK`
that `HSTU.predict_topk` returns.  The cross-head merge + first-occurrence dedup (reference collector.py:249-275:
a [B,N] bool table and an H*K-step Python loop) and the hit matrix (collector.py:300-316) run in gfx950 kernels;
everything stays on the device until the per-pred_len `[B, K+1]` result is appended.
"""
import copy

import numpy as np
import torch


class DataStruct(object):
    def __init__(self):
        self._tensor_lists = {}
        self._data_dict = {}

    def __getitem__(self, name):
        return self._data_dict[name]

    def __setitem__(self, name, value):
        self._data_dict[name] = value

    def __delitem__(self, name):
        self._data_dict.pop(name)

    def __contains__(self, key):
        return key in self._data_dict

    def get(self, name):
        if name not in self._data_dict:
            raise IndexError("Can not load the data without registration !")
        return self[name]

    def set(self, name, value):
        self._data_dict[name] = value

    def update_tensor(self, name, value):
        # per-batch results stay where they were produced (the device); the reference's `.cpu().clone()` per batch
        # (collector.py:48-54) would serialise the host against the GPU once per batch.
        self._tensor_lists.setdefault(name, []).append(value.detach())

    def finalize_tensors(self):
        # one concatenation and one device->host copy per key for the whole evaluation
        for name, lst in self._tensor_lists.items():
            if lst:
                self._data_dict[name] = torch.cat(lst, dim=0).cpu()
        self._tensor_lists.clear()


class Collector(object):
    def __init__(self, config):
        self.config = config
        self.metrics_pred_len_list = config['metrics_pred_len_list']
        self.eval_pred_len = config['eval_pred_len']
        self.data_struct = {p: DataStruct() for p in self.metrics_pred_len_list}
        self.data_struct[-1] = DataStruct()
        self.topk = config['topk']
        hi = config['head_interaction']
        if hi in ('multiplicative', 'hierarchical'):
            self.medusa_num_heads = config['num_segment_head'] * config['num_prior_head']
        elif hi == 'additive':
            self.medusa_num_heads = config['num_segment_head'] + config['num_prior_head']
        else:
            raise ValueError(f'Unknown head_interaction: {hi}')
        self.split_mode = config['split_mode']
        self.all_tags = None

    def set_all_tags(self, item_tags):
        self.all_tags = item_tags

    def reset_all_tags(self):
        self.all_tags = None

    # ------------------------------------------------------------------------------------------
    def _decode(self, scores, top_k, detail):
        """-> merged item ids [B, top_k] int64 on the device."""
        from mhr_amd import ops
        fused = hasattr(scores, 'indices')
        if fused:
            vals, idx = scores.values, scores.indices
            if vals.shape[-1] < top_k:
                raise ValueError(f"FusedTopK holds k={vals.shape[-1]} < max(topk)={top_k}")
            vals, idx = vals[..., :top_k].contiguous(), idx[..., :top_k].contiguous()
            B, H = vals.shape[:2]
        else:
            if not scores.is_cuda:
                raise RuntimeError("Collector: scores must live on the GPU (no CPU decode path)")
            scores = scores.float()
            B, H = scores.shape[:2]
            if H > 1 and self.split_mode == 'average':
                finite = torch.isfinite(scores)
                avg = torch.sum(scores.masked_fill(~finite, 0), dim=1) / (torch.sum(finite, dim=1) + 1e-8)
                return torch.topk(avg, top_k, dim=-1)[1]
            vals, idx = torch.topk(scores, top_k, dim=-1)
        if H == 1:
            if detail is not None:
                detail['values'], detail['idx'] = vals[:, 0], idx[:, 0]
            return idx[:, 0].contiguous()
        if self.split_mode != 'combine':
            raise ValueError(f"split_mode={self.split_mode} needs dense scores" if fused else f'Unknown split_mode: {self.split_mode}')
        out_idx, out_val, out_src, status = ops.multihead_merge_dedup(vals.contiguous(), idx.contiguous(), B, H, top_k)
        if detail is not None:
            detail.update(values=out_val, head_source=out_src, idx=out_idx, values_by_head=vals, idx_by_head=idx, unique=status)
        return out_idx

    def eval_batch_collect(self, scores_tensor, positive_u, positive_i, tag_category=None, outlier_users=None,
                           log_detailed_results=False):
        from mhr_amd import ops
        if tag_category is not None:
            for p in self.metrics_pred_len_list:
                self.data_struct[p].update_tensor('rec.tgt_tags', torch.any(tag_category[:, :p + 1].bool(), dim=1))
        if outlier_users is not None:
            self.data_struct[self.eval_pred_len - 1].update_tensor('rec.outlier_users', outlier_users)
        top_k = max(self.topk)
        detail = {} if log_detailed_results else None
        topk_idx = self._decode(scores_tensor, top_k, detail)
        dev = topk_idx.device
        positive_i = positive_i.to(dev).contiguous()
        B = topk_idx.shape[0]
        if self.all_tags is not None:
            self.data_struct[-1].update_tensor('rec.rec_tags', self.all_tags.to(dev)[topk_idx])
        # distinct positives among the ascending-sorted targets (reference collector.py:301-304)
        srt, _ = positive_i.sort(dim=1)
        first = torch.ones_like(srt, dtype=torch.bool)
        first[:, 1:] = srt[:, 1:] != srt[:, :-1]
        pos_len_full = first.cumsum(dim=1).int()
        for p in self.metrics_pred_len_list:
            hit = ops.hit_matrix(topk_idx, positive_i, p + 1)                       # cumulative over [0 : p+1]
            self.data_struct[p].update_tensor('rec.topk', torch.cat((hit.int(), pos_len_full[:, p:p + 1]), dim=1))
        if log_detailed_results:
            out = {}
            for k_, v in detail.items():
                out[k_] = v.detach().cpu().numpy() if k_.startswith('values') or k_ in ('head_source', 'unique') else v.detach().cpu().tolist()
            return out
        return {}

    def get_data_struct(self, pred_idx=0):
        self.data_struct[pred_idx].finalize_tensors()
        returned = copy.deepcopy(self.data_struct[pred_idx])
        for key in ('rec.rec_tags', 'rec.tgt_tags', 'rec.outlier_users', 'rec.topk'):
            if key in self.data_struct[pred_idx]:
                del self.data_struct[pred_idx][key]
        return returned

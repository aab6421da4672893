"""REMI on the HSTU encoder, MI355X-native: ComiRec's causal multi-interest read-out with REMI's two
training-time additions (reference `code/REC/model/IDNet/remi.py`, class REMI, lines 14-517; same registry surface,
constructor keys - `lambda_rr`, `beta_ihn`, `interest_num`, `interest_hidden` / `interest_hidden_ratio`,
`attention_net_bias` - and state_dict names):

  * routing regularisation (remi.py:156-201, 342-358): per causal window (b, t) the squared norm of the K variances of the
    routing weights, averaged over the valid positions.  The reference forms the [B*L, K, L] routing matrix of the
    materialised windows; the weight of position m in window t is e[m] / den[t] (the prefix form of the ComiRec read-out),
    so  sum_m A_km^2 = exp(logcumsumexp(2a)[t] - 2 logcumsumexp(a)[t])  and, the weights summing to one over the n_t valid positions,
    variance_k = (sum_m A_km^2 - 1 / n_t) / D  - three prefix sums over [B, L, K], autograd through torch.
  * interest-aware hard negatives (remi.py:203-288): loss = log(exp(l+) + Neg) - l+ with Neg = sum_j exp((beta + 1) l_j) /
    mean_j exp(beta l_j) over the sampled pool: the dense path of the sampled softmax (library GEMM logit chunks) with the
    `ihn_dense` epilogue kernels of csrc/wide.hip; the same false-negative rule and top-k logs as the plain loss.

Prediction is ComiRec's (remi.py:439-506): one interest per "head" over the whole sequence -> the fused multi-head decode.
"""
import torch
import torch.nn as nn

from REC.model.IDNet.comirec import ComiRec
from REC.model.IDNet.hstu import truncated_normal


class REMI(ComiRec):
    def __init__(self, config, dataload):
        super().__init__(config, dataload)
        D = self._hstu_embedding_dim
        self.lambda_rr = config.get('lambda_rr', 100.0)
        self.beta_ihn = config.get('beta_ihn', 1.0)
        self._ihn_beta = float(self.beta_ihn) if self.beta_ihn > 0 else 0.0
        hidden = config.get('interest_hidden', int(D * config.get('interest_hidden_ratio', 0.5)))
        if hidden != self.interest_hidden or not config.get('attention_net_bias', True):
            self.interest_hidden = hidden
            self.attention_net = nn.Sequential(nn.Linear(D, hidden, bias=config.get('attention_net_bias', True)), nn.Tanh(),
                                               nn.Dropout(self._linear_dropout_rate), nn.Linear(hidden, self.num_interest, bias=False))
            for p in self.attention_net.parameters():
                truncated_normal(p.data, mean=0.0, std=0.02)

    def _routing_loss(self, a, den, valid, model_out):
        if not self.lambda_rr > 0:
            return 0.0
        D = self._hstu_embedding_dim
        vf = valid.float()
        n = torch.cumsum(vf, dim=1).clamp_min(1.0)                                            # valid positions up to t
        # sum_m A_km^2 of window t in log space (a peaky routing spans hundreds of e-folds: the plain prefix sums underflow)
        af = a.masked_fill(~valid[..., None], -1.0e4)
        sq = torch.exp(torch.logcumsumexp(2.0 * af, dim=1) - 2.0 * torch.logcumsumexp(af, dim=1))
        var = (sq - 1.0 / n[..., None]) / D                                                   # [B,L,K]
        norm_sq = (var * var).sum(-1)                                                         # [B,L]
        rr = (norm_sq * vf).sum() / vf.sum().clamp_min(1.0)
        model_out["rr_loss"] = rr.detach()
        return self.lambda_rr * rr

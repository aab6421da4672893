"""Decoding-head block (reference `code/REC/model/llm_heads.py:5-40`): y = x + SiLU(Linear(x)), with an optional
affine LayerNorm in front.  Parameter names (`linear`, `norm`) match the reference's state_dict."""
import torch
from torch import nn


class ResBlock(nn.Module):
    def __init__(self, hidden_size, use_norm=False, zero_init=True):
        super().__init__()
        self.use_norm = use_norm
        if use_norm:
            self.norm = nn.LayerNorm(hidden_size)
        self.linear = nn.Linear(hidden_size, hidden_size)
        if zero_init:
            nn.init.zeros_(self.linear.weight)
        else:
            nn.init.trunc_normal_(self.linear.weight, std=0.02)
        self.act = nn.SiLU()

    def forward(self, x):
        if self.use_norm:
            x = self.norm(x)
        return x + self.act(self.linear(x))

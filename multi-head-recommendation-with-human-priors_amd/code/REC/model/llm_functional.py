"""Autograd glue between the Llama-style decoder module (REC/model/HLLM/modeling_llama.py) and the gfx950 kernels.

Same mixed-precision contract as the HSTU path (Fabric `bf16-mixed`): fp32 master weights and residual stream, bf16
GEMM operands and branch activations, fp32 statistics.  The dense projections are library GEMMs (SplitKLinearFn in
hstu_functional.py); RMSNorm (+ residual add), RoPE, causal softmax attention and SwiGLU are hand-written kernels.
"""
import torch
from torch.autograd import Function

from mhr_amd import ops


class RMSNormFn(Function):
    """(x fp32, weight) -> bf16 normalised rows (reference model/HLLM/modeling_llama.py:266-280)."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        x = x.contiguous()
        _, y, rstd = ops.rmsnorm_fwd(x, weight, None, eps)
        ctx.save_for_backward(x, weight, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, rstd = ctx.saved_tensors
        dx, _, dw = ops.rmsnorm_bwd(dy.contiguous(), x, weight, rstd)
        return dx, dw, None


class AddRMSNormFn(Function):
    """(x fp32, branch bf16, weight) -> (x + branch fp32, RMSNorm(x + branch) bf16): the residual add of one block
    fused with the norm that follows it (modeling_llama.py:779+783, 785+768 of the next layer, 1108)."""

    @staticmethod
    def forward(ctx, x, res, weight, eps):
        x_out, y, rstd = ops.rmsnorm_fwd(x.contiguous(), weight, res.contiguous(), eps)
        ctx.save_for_backward(x_out, weight, rstd)
        return x_out, y

    @staticmethod
    def backward(ctx, d_xout, dy):
        x_out, weight, rstd = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros(x_out.shape, dtype=torch.bfloat16, device=x_out.device)
        d_xout = None if d_xout is None else d_xout.contiguous().float()
        dx, dres, dw = ops.rmsnorm_bwd(dy.contiguous(), x_out, weight, rstd, d_xout, want_dres=True)
        return dx, dres, dw, None


class SwigluFn(Function):
    """[gate | up] bf16 -> silu(gate) * up (modeling_llama.py:484)."""

    @staticmethod
    def forward(ctx, gate_up):
        gate_up = gate_up.contiguous()
        ctx.save_for_backward(gate_up)
        return ops.swiglu_fwd(gate_up)

    @staticmethod
    def backward(ctx, d_act):
        (gate_up,) = ctx.saved_tensors
        return ops.swiglu_bwd(gate_up, d_act.contiguous())


class RopeAttentionFn(Function):
    """Packed qkv rows [T, (nh + 2 nkv) hd] bf16 -> causal softmax attention output [T, nh hd] bf16, with the rotary
    embedding applied to q and k in place first (modeling_llama.py:622-682).  The qkv matrix is the output of the
    projection GEMM and is not needed unrotated by anyone, so it is rotated in place and kept for the backward; the
    backward un-rotates the q / k gradients in place."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, positions, key_valid, cu_seqlens, n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale):
        qkv = qkv.contiguous()
        ops.rope_inplace(qkv, n_heads + n_kv_heads, head_dim, cos, sin, positions, seq_len=max_len)
        out, lse = ops.softmax_attn_fwd(qkv, n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale, cu_seqlens, key_valid)
        ctx.save_for_backward(qkv, out, lse, cos, sin, positions, key_valid, cu_seqlens)
        ctx.cfg = (n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, out, lse, cos, sin, positions, key_valid, cu_seqlens = ctx.saved_tensors
        n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale = ctx.cfg
        dqkv = ops.softmax_attn_bwd(qkv, out, d_out.contiguous(), lse, n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale,
                                    cu_seqlens, key_valid)
        ops.rope_inplace(dqkv, n_heads + n_kv_heads, head_dim, cos, sin, positions, seq_len=max_len, inverse=True)
        return (dqkv,) + (None,) * 11

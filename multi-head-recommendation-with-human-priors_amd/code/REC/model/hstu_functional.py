"""Autograd glue between the reference-shaped HSTU module and the gfx950 kernels (mhr_amd.ops).

Each Function's forward/backward enqueues hand-written HIP kernels on the current stream; the dense
projections between them are plain library GEMMs (torch.matmul -> hipBLASLt) as SURVEY.md section 2.3 plans.
Mixed precision follows the reference's Fabric `bf16-mixed` run: fp32 master weights and residual stream,
bf16 GEMM operands, fp32 statistics / loss.
"""
import os
from collections import OrderedDict

import torch
from torch.autograd import Function

from mhr_amd import ops

_SPLITK_MAX = int(os.environ.get("MHR_SPLITK_MAX", "16"))
_SUM_KERNEL = os.environ.get("MHR_SUM_KERNEL", "1") != "0"


class LayerNormFn(Function):
    """Affine-free LayerNorm, fp32 in -> bf16 out (reference model/IDNet/hstu.py:213-214, 241)."""

    @staticmethod
    def forward(ctx, x, eps, out=None):
        y, mean, rstd = ops.layernorm_fwd(x, torch.bfloat16, eps, out=out)
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        return ops.layernorm_bwd(dy.contiguous(), x, mean, rstd, dx_dtype=torch.float32), None, None


class LayerNormResidualFn(Function):
    """x -> (x, LN(x)): the first layer's LayerNorm together with the residual branch that leaves x for the first residual add
    (reference hstu.py:241 and 286-287).  One node for both uses of x, so that the backward adds the LayerNorm's gradient INTO the
    residual branch's (mhr_layernorm_bwd, accumulate) instead of autograd summing two [B L, D] fp32 tensors in a pass of its own.
    residual_grad_is_fresh: the residual branch's gradient is a tensor nobody else reads (it comes out of AddLayerNormFn's
    backward), so it may be accumulated into in place."""

    @staticmethod
    def forward(ctx, x, eps, residual_grad_is_fresh, out=None):
        y, mean, rstd = ops.layernorm_fwd(x, torch.bfloat16, eps, out=out)
        ctx.save_for_backward(x, mean, rstd)
        ctx.inplace = bool(residual_grad_is_fresh)
        return x, y

    @staticmethod
    def backward(ctx, d_x, d_y):
        x, mean, rstd = ctx.saved_tensors
        if d_y is None:
            return d_x, None, None, None
        if d_x is not None and ctx.inplace and d_x.dtype == torch.float32 and d_x.is_contiguous():
            return ops.layernorm_bwd(d_y.contiguous(), x, mean, rstd, dx=d_x, accumulate=True), None, None, None
        g = ops.layernorm_bwd(d_y.contiguous(), x, mean, rstd, dx_dtype=torch.float32)
        return (g if d_x is None else g + d_x), None, None, None


class AddLayerNormFn(Function):
    """(x, y) -> (x + y, LN(x + y)): the residual add of one HSTU layer fused with the LayerNorm of the next
    (reference hstu.py:286-287, 241).  x fp32, y bf16 -> x_out fp32, xn bf16."""

    @staticmethod
    def forward(ctx, x, y, eps, xn_out=None, dy_out=None, dead=None):
        # dead = (first_row, L): rows in front of a sequence's first valid key are not loaded (ops._dead_args) - forward and backward
        x_out, xn, mean, rstd = ops.add_layernorm_fwd(x.contiguous(), y.contiguous(), eps, xn_out=xn_out, dead=dead)
        ctx.save_for_backward(x_out, mean, rstd)
        ctx.dy_out = dy_out                      # (the caller's buffer for the gradient of y: WeightGradStack)
        ctx.dead = dead
        return x_out, xn

    @staticmethod
    def backward(ctx, d_xout, d_xn):
        x_out, mean, rstd = ctx.saved_tensors
        if d_xout is None:
            d_xout = torch.zeros_like(x_out)
        if d_xn is None:
            d_xn = torch.zeros(x_out.shape, dtype=torch.bfloat16, device=x_out.device)
        dx, dy = ops.add_layernorm_bwd(d_xn.contiguous(), x_out, mean, rstd, d_xout.contiguous().float(), dy_out=ctx.dy_out,
                                       dead=ctx.dead)
        return dx, dy, None, None, None, None


class AddCastFn(Function):
    """(x fp32, y bf16) -> (x + y fp32, bf16(x + y)): the residual add behind the last encoder layer together with the bf16
    copy the decoding heads' GEMM reads (reference hstu.py:286-288; llm_heads.py:30-40 under autocast).  One pass each way."""

    @staticmethod
    def forward(ctx, x, y, dy_out=None):
        ctx.dy_out = dy_out                      # (the caller's buffer for the gradient of y: WeightGradStack)
        return ops.add_cast(x.contiguous(), y.contiguous())

    @staticmethod
    def backward(ctx, d_out, d_out16):
        if d_out is None and d_out16 is None:
            return None, None, None
        if d_out16 is None:
            return d_out, d_out.to(torch.bfloat16), None
        if d_out is None:
            return d_out16.float(), d_out16, None
        # dx = d_out + d_out16 (fp32), dy = bf16(dx)
        return ops.add_cast(d_out.contiguous(), d_out16.contiguous(), out16=ctx.dy_out) + (None,)


class WeightGradStack:
    """The operands of the encoder layers' weight-gradient products side by side, so that the L layers' products run as ONE
    batched split-K GEMM per projection at the optimizer step instead of L small ones inside the backward.

    Nothing in the backward reads a weight gradient, and each of these products is too small for the chip on its own:
    [B L, D]^T [B L, 4 D] as 16 slices is 256 workgroups, 27 us per layer at cfg1; [B L, D]^T [B L, D] 64 workgroups, 19 us.
    Stacked over the 8 layers they take 124 us instead of 220 and 40 us instead of 158 (hipBLASLt, measured in isolation).
    The producers write straight into the stack: xn[l] (LayerNorm in front of the uvqk projection), dh[l] (gradient of its
    output, HSTUCoreFn.backward), o[l] (gate output = input of the output projection), dy[l] (gradient of that projection's
    output: AddLayerNormFn / AddCastFn backward).  `flush()` - queued with the fused optimizer, run first thing in its step -
    forms [L S, ., .] bf16 partials with two batched GEMMs and queues their per-layer sums into the parameters' fp32 gradients
    (one `mhr_sum_rows_many` launch per projection).  Needs the optimizer's one-backward-per-step mode (its partial arena)."""

    def __init__(self, opt, uvqk_params, o_params, R, D, device):
        L = len(uvqk_params)
        self.opt, self.L, self.R, self.D = opt, L, R, D
        self.params = {"uvqk": list(uvqk_params), "o": list(o_params)}
        bf = dict(dtype=torch.bfloat16, device=device)
        self.xn, self.dh = torch.empty(L, R, D, **bf), torch.empty(L, R, 4 * D, **bf)
        self.o, self.dy = torch.empty(L, R, D, **bf), torch.empty(L, R, D, **bf)
        self.seen = {"uvqk": set(), "o": set()}
        opt.defer(self.flush)

    def operands(self, kind, layer, x, dy):
        """The backward of layer `layer`'s projection hands over its operands; they are the stack's own slices unless somebody
        re-materialised them on the way (then they are copied in)."""
        xs, ds = (self.xn, self.dh) if kind == "uvqk" else (self.o, self.dy)
        if x.data_ptr() != xs[layer].data_ptr():
            xs[layer].copy_(x.reshape(xs[layer].shape))
        if dy.data_ptr() != ds[layer].data_ptr():
            ds[layer].copy_(dy.reshape(ds[layer].shape))
        self.seen[kind].add(layer)

    @staticmethod
    def _splits(R, prefer):
        return next((s for s in prefer if R % s == 0 and R // s >= 256), 1)

    def flush(self):
        n = {k: len(v) for k, v in self.seen.items()}
        if n["uvqk"] == 0 and n["o"] == 0:
            return                                # a forward whose backward never ran
        if n["uvqk"] != self.L or n["o"] != self.L:
            raise RuntimeError(f"WeightGradStack: {n} of {self.L} layers handed their operands over")
        L, R, D = self.L, self.R, self.D
        S1 = self._splits(R, (8, 4, 2))           # (slices per layer: measured best at cfg1 - 8 for the wide product, 16 for the square)
        w1 = torch.bmm(self.xn.view(L * S1, R // S1, D).transpose(1, 2), self.dh.view(L * S1, R // S1, 4 * D))      # [L S1, D, 4D]
        S2 = self._splits(R, (16, 8, 4, 2))
        w2 = torch.bmm(self.dy.view(L * S2, R // S2, D).transpose(1, 2), self.o.view(L * S2, R // S2, D))           # [L S2, D(out), D(in)]
        for l in range(L):
            self.opt.queue_rows_sum(w1[l * S1:(l + 1) * S1].view(S1, -1), self.params["uvqk"][l].grad)
            self.opt.queue_rows_sum(w2[l * S2:(l + 1) * S2].view(S2, -1), self.params["o"][l].grad)
        self.seen = {"uvqk": set(), "o": set()}


ROWS_GEMM = os.environ.get("MHR_ROWS_GEMM", "1") != "0"
ROWS_GEMM_MIN_M = int(os.environ.get("MHR_ROWS_GEMM_MIN_M", "8192"))    # rows from which the stationary operand's load is amortised


def _rows_gemm_pays(x, w_kn):
    """The hand-written token-rows projection (ops.rows_gemm) where it beats the library product: the wide-output, K = 256
    uvqk shape with enough rows to amortise the stationary operand's load (measured at cfg1: 23 us against 31 us; the
    [M, 256] x [256, 256] products and K = 64 are faster in the library)."""
    M, K = x.shape
    N = w_kn.shape[1]
    return (K == 256 and N >= 1024 and M >= ROWS_GEMM_MIN_M and x.dtype == torch.bfloat16 and w_kn.dtype == torch.bfloat16
            and x.stride(1) == 1 and w_kn.is_contiguous() and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and w_kn.data_ptr() % 16 == 0
            and ops.rows_gemm_supported(M, N, K, True))


class SplitKLinearFn(Function):
    """y = x @ W (w_is_nk=False, W [K, N]) or x @ W.T + b (w_is_nk=True, W [N, K], the nn.Linear layout), bf16 GEMMs on
    fp32 master weights (reference hstu.py:236-239 under bf16-mixed autocast).

    The weight gradient contracts over all B*L tokens into a 256 x 1024 (or 256 x 256) result: as one GEMM that is 64
    output tiles on a 256-CU part (measured 94-137 us per call, ~2.2 ms per step at cfg1).  It is issued as a
    split-K batched GEMM instead - S slices of the token axis, S x more tiles - and the S partials are summed in fp32."""

    @staticmethod
    def forward(ctx, x, w, b, w_is_nk, w_bf16, defer=None):
        ctx.defer = defer                        # (WeightGradStack, kind, layer): the weight gradient is formed for all layers at once
        if w_bf16 is None:
            sh = getattr(w, "_mhr_bf16", None)       # shadow kept by the fused optimizer, valid while nobody edited w
            w_bf16 = sh if (sh is not None and w._version == w._mhr_ver) else None
        wb = w.to(torch.bfloat16) if w_bf16 is None else w_bf16
        bb = None
        if b is not None:
            sh = getattr(b, "_mhr_bf16", None)
            bb = sh if (sh is not None and b._version == b._mhr_ver) else b.to(torch.bfloat16)
        if w_is_nk:
            y = torch.nn.functional.linear(x, wb, bb)
        elif ROWS_GEMM and _rows_gemm_pays(x, wb):
            y = ops.rows_gemm(x, wb, None, w_is_kn=True)     # the uvqk projection: weight stationary, token rows streamed
        else:
            y = x @ wb
        ctx.save_for_backward(x, wb)
        ctx.w_is_nk, ctx.has_bias = w_is_nk, b is not None
        # parameters whose .grad is a view of the optimizer's flat gradient buffer receive their gradient in place
        ctx.w_leaf = w if getattr(w, "_mhr_direct_grad", False) and w.grad is not None else None
        ctx.b_leaf = b if (b is not None and getattr(b, "_mhr_direct_grad", False) and b.grad is not None) else None
        return y

    @staticmethod
    def _into(leaf, parts):
        """leaf.grad (a view of the optimizer's flat fp32 gradient buffer, zeroed once per step) += sum of `parts` over
        dim 0: one kernel for the split-K reduction / the bias column sum, no autograd accumulate pass."""
        g = leaf.grad
        if _SUM_KERNEL and parts.dtype == torch.bfloat16 and parts.is_contiguous() and g.is_contiguous() and g.numel() % 8 == 0:
            ops.sum_rows_into(parts, g)
        else:
            g.add_(torch.sum(parts, 0, dtype=torch.float32).view_as(g))

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ wb if ctx.w_is_nk else dy @ wb.t()
        R = x.shape[0]
        if ctx.defer is not None and ctx.w_leaf is not None:
            stack, kind, layer = ctx.defer
            stack.operands(kind, layer, x, dy)    # (both are slices of the stack already; copied in if somebody re-made them)
            if ctx.has_bias and ctx.b_leaf is not None:
                bopt = getattr(ctx.b_leaf, "_mhr_opt", None)
                if not (bopt is not None and bopt.queue_bias_sum(dy, ctx.b_leaf)):
                    SplitKLinearFn._into(ctx.b_leaf, dy)
                return dx, None, None, None, None, None
            return dx, None, (torch.sum(dy, 0, dtype=torch.float32) if ctx.has_bias else None), None, None, None
        # split the token axis only as far as the output needs it: a [1024, 256] gradient is 4 macro-tiles on a 256-CU part,
        # a [11264, 2048] one (LLM decoder) already has 352 and splitting it would only add partials to write and sum
        tiles = -(-wb.shape[0] // 256) * -(-wb.shape[1] // 256)
        s_max = min(max(1, 256 // tiles), _SPLITK_MAX)
        S = next((s for s in (64, 32, 16, 8, 4, 2) if s <= s_max and R % s == 0 and R // s >= 256), 1)
        xs, dys = x.view(S, R // S, -1), dy.view(S, R // S, -1)
        # with the optimizer's partial arena on, the S partials land in the parameter's columns of ONE [S, total] buffer that
        # the optimizer sums with one launch at the step (optim.FusedAdamW.enable_partial_arena)
        opt = getattr(ctx.w_leaf, "_mhr_opt", None) if ctx.w_leaf is not None else None
        slot = opt.partial_view(ctx.w_leaf, S) if (opt is not None and S > 1 and _SUM_KERNEL) else None
        if ctx.w_is_nk:
            dw = torch.bmm(dys.transpose(1, 2), xs, out=slot) if slot is not None else torch.bmm(dys.transpose(1, 2), xs)   # [S, N, K]
        else:
            dw = torch.bmm(xs.transpose(1, 2), dys, out=slot) if slot is not None else torch.bmm(xs.transpose(1, 2), dys)   # [S, K, N]
        if slot is not None:
            dw = None
        elif ctx.w_leaf is not None:                              # reduction of the split-K partials straight into p.grad
            SplitKLinearFn._into(ctx.w_leaf, dw)
            dw = None
        else:
            dw = torch.sum(dw, 0, dtype=torch.float32) if S > 1 else dw[0].float()      # cast fused into the reduction
        db = None
        if ctx.has_bias:
            if ctx.b_leaf is not None:
                bopt = getattr(ctx.b_leaf, "_mhr_opt", None)
                if not (bopt is not None and _SUM_KERNEL and bopt.queue_bias_sum(dy, ctx.b_leaf)):     # all such sums: one launch at the step
                    SplitKLinearFn._into(ctx.b_leaf, dy)
            else:
                db = torch.sum(dy, 0, dtype=torch.float32)
        return dx, dw, db, None, None, None


class FusedHeadsLinearFn(Function):
    """z = x @ [W_0; ...; W_{H-1}]^T + [b_0; ...; b_{H-1}] for H heads whose parameters the fused optimizer laid out back to
    back (mhr_amd.optim.fused_views / fused_flat): operands are views of the flat bf16 shadow, gradients are reduced straight
    into the flat fp32 gradient views - no concatenation, no casts, no per-head accumulate kernels (14 small launches per
    step at cfg1).  The parameters themselves do not pass through autograd."""

    @staticmethod
    def forward(ctx, x, w16, b16, gw, gb):
        ctx.save_for_backward(x, w16)
        ctx.gw, ctx.gb = gw, gb
        return torch.nn.functional.linear(x, w16, b16)

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w16
        R = x.shape[0]
        tiles = -(-w16.shape[0] // 256) * -(-w16.shape[1] // 256)
        s_max = min(max(1, 256 // tiles), _SPLITK_MAX)
        S = next((s for s in (64, 32, 16, 8, 4, 2) if s <= s_max and R % s == 0 and R // s >= 256), 1)
        dw = torch.bmm(dy.view(S, R // S, -1).transpose(1, 2), x.view(S, R // S, -1))          # [S, N, K] bf16 partials
        ops.sum_rows_into(dw, ctx.gw)
        ops.sum_rows_into(dy, ctx.gb)
        return dx, None, None, None, None


class HeadsResidualFn(Function):
    """out[b, h, l] = x[b, l] + silu(z[b, l, h]) in the [B, H, L, D] layout the loss reads (reference llm_heads.py:5-40 per
    head + the stack / permute of hstu.py:665-667); x [B*L, D] fp32, z [B*L, H*D] bf16 (the heads' one concatenated GEMM)."""

    @staticmethod
    def forward(ctx, x, z, B, L, H):
        ctx.save_for_backward(z)
        ctx.cfg = (B, L, H)
        return ops.heads_residual_fwd(x.contiguous(), z.contiguous(), B, L, H)

    @staticmethod
    def backward(ctx, d_out):
        (z,) = ctx.saved_tensors
        B, L, H = ctx.cfg
        dz, dx = ops.heads_residual_bwd(d_out.contiguous().float(), z, B, L, H)
        return dx, dz, None, None, None


class FusedParamLinearFn(Function):
    """y = x @ [W0; W1; ...]^T for parameters the fused optimizer laid out back to back (mhr_amd.optim.fused_views): the
    operand is the bf16 shadow view of the flat weight buffer and the weight gradient goes straight into the flat gradient
    view - no concatenation, no cast, no autograd accumulation (a 1.1 B-parameter decoder otherwise moves ~24 GB per step
    for those three).  `leaf` is the first parameter of the group (carries the per-step first-write marker)."""

    @staticmethod
    def forward(ctx, x, w16, gview, leaf):
        ctx.save_for_backward(x, w16)
        ctx.gview, ctx.leaf = gview, leaf
        return torch.nn.functional.linear(x, w16)

    @staticmethod
    def backward(ctx, dy):
        from mhr_amd import optim
        x, w16 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w16
        dw = torch.mm(dy.t(), x, out_dtype=torch.float32)              # fp32 straight out of the MFMA accumulators
        if getattr(ctx.leaf, "_mhr_epoch_f", -1) != optim.GRAD_EPOCH:
            ctx.leaf._mhr_epoch_f = optim.GRAD_EPOCH
            ctx.gview.copy_(dw)
        else:
            ctx.gview.add_(dw)
        return dx, None, None, None


class MoveRowsFn(Function):
    """Rows between the window layout [B L, D] and the packed layout [capacity, D] (ops.seq_pack_maps): y[r] = x[fwd_idx[r]]
    (zeros where fwd_idx[r] < 0).  The map is injective, so the backward is the same gather through the inverse map."""

    @staticmethod
    def forward(ctx, x, fwd_idx, bwd_idx):
        ctx.bwd_idx = bwd_idx
        return ops.rows_gather_masked(x.contiguous(), fwd_idx)

    @staticmethod
    def backward(ctx, dy):
        return ops.rows_gather_masked(dy.contiguous(), ctx.bwd_idx), None, None


class HSTUCoreFn(Function):
    """h = LN(x) @ W_uvqk (pre-activation, [B*L, 4D] bf16)  ->  o = silu(u) * LN(attn(silu(q), silu(k), silu(v))) * drop.

    Fuses reference hstu.py:244-285 minus the two GEMMs: the SiLU of the uvqk product is applied on load inside
    the attention and gate kernels, the [B,H,L,L] score tensor is never formed, and the backward writes all four
    column blocks of dh in place (du from the gate kernel, dv|dq|dk from the attention kernel).
    """

    @staticmethod
    def forward(ctx, h, key_valid, B, L, n_heads, head_dim, eps, dropout_p, seed, seed_dev=None, o_out=None, dh_out=None):
        D = n_heads * head_dim
        # the activated q / k / v are NOT saved: the backward recomputes silu() while it stages them (h is kept anyway)
        layout = getattr(key_valid, "_mhr_layout", None)         # (HSTU._encode: once per batch, shared by the layers)
        a, _ = ops.hstu_attn_fwd(h, key_valid, B, L, n_heads, head_dim, apply_silu=True, save_act=False, layout=layout)
        dead = (layout[2], L) if (ops.DEAD_ROWS and layout is not None and len(layout) > 2 and layout[2] is not None
                                  and getattr(key_valid, "_mhr_dead_ok", True)) else None
        o, mean, rstd = ops.ln_gate_fwd(h, a, D, torch.bfloat16, eps, dropout_p, seed, seed_dev, out=o_out, dead=dead)
        ctx.dh_out = dh_out                      # (the caller's buffer for the gradient of h: WeightGradStack)
        ctx.save_for_backward(h, key_valid, a, mean, rstd)
        ctx.cfg = (B, L, n_heads, head_dim, dropout_p, seed, seed_dev)
        ctx.layout, ctx.dead = layout, dead
        return o

    @staticmethod
    def backward(ctx, d_o):
        h, key_valid, a, mean, rstd = ctx.saved_tensors
        B, L, n_heads, head_dim, dropout_p, seed, seed_dev = ctx.cfg
        D = n_heads * head_dim
        dh = torch.empty_like(h) if ctx.dh_out is None else ctx.dh_out.view(h.shape)
        da = ops.ln_gate_bwd(d_o.contiguous(), h, a, mean, rstd, dh, D, dropout_p, seed, seed_dev, dead=ctx.dead)
        ops.hstu_attn_bwd(h, None, key_valid, da, dh, B, L, n_heads, head_dim, apply_silu=True, layout=ctx.layout)
        return (dh,) + (None,) * 11


class SparseRowGrad:
    """Gradient of the item table as (sorted ids, summed rows): what `EmbeddingGatherFn.backward` leaves on the
    module instead of a dense [item_num, D] tensor.  `rows[i]` is valid where `sorted_ids[i]` starts a segment;
    `row_slot[id]` points at it (and is reset by the fused Adam)."""
    __slots__ = ("sorted_ids", "rows", "row_slot", "n_rows")

    def __init__(self, sorted_ids, rows, row_slot, n_rows):
        self.sorted_ids, self.rows, self.row_slot, self.n_rows = sorted_ids, rows, row_slot, n_rows

    def to_dense(self):
        """Dense [n_rows, D] view of the gradient (tests / reference-style optimizers)."""
        head = torch.ones_like(self.sorted_ids, dtype=torch.bool)
        head[1:] = self.sorted_ids[1:] != self.sorted_ids[:-1]
        dense = torch.zeros(self.n_rows, self.rows.shape[1], dtype=torch.float32, device=self.rows.device)
        dense[self.sorted_ids[head]] = self.rows[head]
        return dense


class EmbeddingGatherFn(Function):
    """rows = table[ids_all] (fp32) and x = table[items[:, :L]] + pos[:L] in one pass over the ids
    (reference hstu.py:637-643, 670, 752).  ids_all = [items.flatten(), negative ids...].

    Backward: deterministic sparse reduction (sort + segment-sum kernel) into `holder.sparse_grad`; with
    holder.dense_grad=True the reference's dense `weight.grad` is produced instead (float-atomic scatter-add).
    """

    @staticmethod
    def forward(ctx, table, pos_table, ids_all, n_item_ids, L, window, holder, negs_elsewhere=False):
        D = table.shape[1]
        n_neg = ids_all.numel() - n_item_ids
        ctx.save_for_backward(ids_all)
        ctx.meta = (n_item_ids, L, window, holder, table.shape[0], D)
        if negs_elsewhere and pos_table is not None:
            # ONE launch (mhr_embedding_gather_step): item windows as fp32 rows + position-added x, the negative pools gathered
            # AND L2-normalised to bf16 with no fp32 copy of their rows.  The normalised pools enter autograd through
            # GatherL2NormFn; this Function's second output is a memory-less stand-in for the pools' raw rows - the gradient
            # GatherL2NormFn returns for it arrives here as d_negs, exactly like the gradient of real rows would
            rows, x, negs_n, norms = ops.embedding_gather_step(table, pos_table, ids_all, n_item_ids, L, window)
            holder._gathered_negs = (negs_n, norms)           # picked up by GatherL2NormFn.forward right behind this call
            return rows, torch.zeros((), dtype=torch.float32, device=table.device).expand(n_neg, D), x
        rows = torch.empty(ids_all.numel(), D, dtype=torch.float32, device=table.device)
        # ONE launch: item windows (rows + position-added x) and negative-pool ids (rows only), written in place
        if pos_table is not None:
            _, x = ops.embedding_gather(table, ids_all, torch.float32, pos_table, L, torch.float32, out=rows, window=window,
                                        n_x_ids=n_item_ids)
        else:                                  # projection tower in between: the position add happens after it
            ops.embedding_gather(table, ids_all, torch.float32, out=rows)
            x = torch.zeros(1, device=table.device)
        # two outputs over ONE buffer (item windows | negative pools): their gradients arrive as two tensors, so the backward
        # has no slice-backward zero fills and no accumulate pass over the [rows, D] gradient
        return rows[:n_item_ids], rows[n_item_ids:], x

    @staticmethod
    def backward(ctx, d_items, d_negs, d_x):
        (ids_all,) = ctx.saved_tensors
        n_item_ids, L, window, holder, n_rows, D = ctx.meta
        dev = ids_all.device
        n_neg_ids = ids_all.numel() - n_item_ids
        d_items = d_items.contiguous() if d_items is not None else torch.zeros(n_item_ids, D, device=dev)
        early_shared = getattr(holder, "_shared_pending", None) is not None        # (data parallel: the negatives' gradient is elsewhere)
        if n_neg_ids and not early_shared:
            d_negs = d_negs.contiguous() if d_negs is not None else torch.zeros(n_neg_ids, D, device=dev)
        d_x = d_x.contiguous() if (d_x is not None and d_x.dim() == 3) else None
        d_pos = None
        if d_x is not None:
            pos = holder.position_embedding.weight
            pg = pos.grad if getattr(pos, "_mhr_direct_grad", False) else None
            if pg is not None and pg.is_contiguous() and pg.dtype == torch.float32 and (L * D) % 8 == 0 and pg.data_ptr() % 16 == 0:
                ops.sum_rows_into(d_x.view(d_x.shape[0], L * D), pg[:L].view(-1))     # sum over the batch straight into pos.grad[:L]
            else:
                d_pos = torch.zeros(pos.shape, dtype=torch.float32, device=dev)
                d_pos[:L] = d_x.sum(dim=0)
        if getattr(holder, "dense_embedding_grad", False):
            gt = torch.zeros(n_rows, D, dtype=torch.float32, device=dev)
            ops.embedding_scatter_add(d_items, ids_all[:n_item_ids].contiguous(), gt)
            if n_neg_ids:
                ops.embedding_scatter_add(d_negs, ids_all[n_item_ids:].contiguous(), gt)
            if d_x is not None:
                ops.embedding_scatter_add(d_x.view(-1, D), ids_all[:n_item_ids].view(-1, window)[:, :L].contiguous().view(-1), gt)
            return gt, d_pos, None, None, None, None, None, None
        if holder._row_slot is None or holder._row_slot.numel() != n_rows:
            holder._row_slot = torch.full((n_rows,), -1, dtype=torch.int32, device=dev)
        from mhr_amd import distributed as dist_
        if dist_.active() or getattr(holder, "accumulate_rows", False):
            # data parallel / gradient accumulation: fold the input-side gradient into the rows and defer the reduction
            # until the optimizer asks for it (HSTU.finish_sparse_grad: cross-rank exchange, then ONE segment-sum over the
            # rows of every micro-batch since the last step)
            shared, holder._shared_pending = getattr(holder, "_shared_pending", None), None
            if shared is not None or not n_neg_ids:
                # the shared-negative block already left for its all-reduce (GatherL2NormFn / L2NormFn backward): only the
                # rank-private rows are handed on - no zero block, no concatenation (33 + 60 MB at cfg1)
                d_rows = d_items
            else:
                d_rows = torch.cat([d_items, d_negs])
            if d_x is not None:
                d_rows[:n_item_ids].view(-1, window, D)[:, :L] += d_x
            if holder._pending_rows is None:
                holder._pending_rows = []
            holder._pending_rows.append((ids_all, d_rows, n_item_ids, shared))
            holder.sparse_grad = None
            return None, d_pos, None, None, None, None, None, None
        pre, holder._presorted = getattr(holder, "_presorted", None), None
        if pre is not None and pre[0].data_ptr() == ids_all.data_ptr() and pre[0].numel() == ids_all.numel():        # sorted (and the row buffer zeroed) early, underneath the encoder forward
            _, sorted_ids, perm, out_rows = pre
        else:
            sorted_ids, perm = torch.sort(ids_all)
            out_rows = torch.zeros(ids_all.numel(), D, dtype=torch.float32, device=dev)
        ops.sparse_rows_segment_sum(sorted_ids, perm, d_items, d_negs if n_neg_ids else None, d_x, L, window, out_rows, holder._row_slot)
        holder.sparse_grad = SparseRowGrad(sorted_ids, out_rows, holder._row_slot, n_rows)
        return None, d_pos, None, None, None, None, None, None


def begin_pending_rows(holder):
    """Data-parallel / accumulation tail of the embedding backward, first half: the exchange goes on the wire (RCCL,
    asynchronous).  Several micro-batches (accumulate_grad > 1, reference trainer.py:521-533) are exchanged as one set:
    their rank-private rows back to back, then their shared-negative blocks."""
    from mhr_amd import distributed as dist_
    pend = holder._pending_rows
    if len(pend) == 1:
        ids_all, d_rows, n_private, shared = pend[0]
    else:
        shared_rows = []
        for ids_i, rows_i, n_i, sp in pend:
            if sp is None:
                shared_rows.append(rows_i[n_i:])
            else:
                if sp[1] is not None:
                    sp[1].wait()
                shared_rows.append(sp[0])
        any_reduced = any(p[3] is not None for p in pend)
        if any_reduced and not all(p[3] is not None for p in pend):
            raise RuntimeError("micro-batches of one step mix early-reduced and plain shared-negative blocks")
        n_private = sum(p[2] for p in pend)
        ids_all = torch.cat([p[0][:p[2]] for p in pend] + [p[0][p[2]:] for p in pend])
        d_rows = torch.cat([p[1][:p[2]] for p in pend] + shared_rows)
        shared = (d_rows[n_private:], None) if any_reduced else None
    ids_pre = getattr(holder, "_dp_ids", None) if len(pend) == 1 else None          # exchanged in the forward (HSTU.forward)
    holder._row_exchange = dist_.begin_row_exchange(ids_all, d_rows, n_private, shared_pending=shared, ids_pre=ids_pre)


def reduce_pending_rows(holder):
    """Second half: the ids are sorted while the rows are still travelling, then the deterministic segment-sum."""
    if getattr(holder, "_row_exchange", None) is None:
        begin_pending_rows(holder)
    ex, holder._row_exchange, holder._pending_rows = holder._row_exchange, None, None
    ids = ex.wait_ids()
    pre, holder._presorted = getattr(holder, "_presorted", None), None
    if pre is not None and pre[0] is ids:             # all ranks' ids were exchanged in the forward: sorted (and the row buffer zeroed) early
        _, sorted_ids, perm, out_rows = pre
        rows_priv, rows_shared = ex.wait_rows()
    else:
        sorted_ids, perm = torch.sort(ids)
        rows_priv, rows_shared = ex.wait_rows()
        out_rows = torch.zeros(sorted_ids.numel(), rows_shared.shape[1], dtype=torch.float32, device=rows_shared.device)
    ops.sparse_rows_segment_sum(sorted_ids, perm, rows_priv.contiguous(), rows_shared.contiguous() if rows_shared.numel() else None,
                                None, 0, 0, out_rows, holder._row_slot)
    holder.sparse_grad = SparseRowGrad(sorted_ids, out_rows, holder._row_slot, holder._row_slot.numel())
    return holder.sparse_grad


# Full-precision gradient hand-over between two of our own autograd Functions.  autograd casts a gradient to the dtype of the
# tensor it belongs to: the fp32 gradient of the bf16 normalised negatives would be rounded to bf16 on the way from the loss to
# the normalisation's backward (one cast kernel there, one back to fp32 here, 17 MB each way at cfg1 - and 8 bits of the
# gradient).  The loss deposits the fp32 tensor under the data pointer of the bf16 tensor and returns a stride-0 placeholder.
# An open slot holds the bf16 tensor itself (its address cannot be reused while the slot exists); at most four are kept.
_F32_GRAD = OrderedDict()          # data_ptr of the bf16 tensor -> [that tensor, deposited fp32 gradient or None]
_PLACEHOLDER = {}


def _open_f32_grad_slot(y):
    _F32_GRAD[y.data_ptr()] = [y, None]
    while len(_F32_GRAD) > 4:      # forwards whose backward never came
        _F32_GRAD.popitem(last=False)
    return y.data_ptr()


def _deposit_f32_grad(of_tensor, grad):
    slot = _F32_GRAD.get(of_tensor.data_ptr())
    if slot is None or grad is None or grad.numel() != slot[0].numel() or grad.numel() != of_tensor.numel():
        return grad
    slot[1] = grad if slot[1] is None else slot[1] + grad
    ph = _PLACEHOLDER.get(grad.device)
    if ph is None:
        ph = _PLACEHOLDER[grad.device] = torch.zeros((), dtype=torch.bfloat16, device=grad.device)
    return ph.expand(of_tensor.shape)


class L2NormFn(Function):
    """y = x / ||x|| in fp32, emitted as bf16 for the MFMA operand (reference hstu.py:672, 754).

    `holder` (data parallel, x = rows of the item table gathered for the shared negatives): the backward hands its
    result to the cross-rank all-reduce right away - asynchronously, on RCCL's stream - and returns no gradient to
    autograd; `reduce_pending_rows` picks the summed block up from `holder._shared_pending`."""

    @staticmethod
    def forward(ctx, x, holder=None):
        y, norms = ops.l2norm_rows(x, torch.bfloat16, want_norms=True)
        ctx.save_for_backward(x, norms)
        ctx.holder = holder
        ctx.key = _open_f32_grad_slot(y) if ctx.needs_input_grad[0] else None     # (NceLossFn.backward deposits there)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, norms = ctx.saved_tensors
        slot = _F32_GRAD.pop(ctx.key, None)
        g32 = slot[1] if slot is not None else None
        if g32 is not None:                           # the loss left its fp32 gradient; dy is the placeholder unless others add to it
            dy = g32 if all(st == 0 for st in dy.stride()) else g32 + dy.float().reshape(g32.shape)
        if x.dtype == torch.float32 and x.is_contiguous():
            g = ops.l2norm_rows_bwd(dy.reshape(x.shape).float().contiguous(), x, norms)
        else:
            n = x / norms[:, None]
            dy = dy.float()
            g = (dy - n * (n * dy).sum(-1, keepdim=True)) / norms[:, None]
        from mhr_amd import distributed as dist_
        if ctx.holder is not None and dist_.active() and dist_.OVERLAP and not getattr(ctx.holder, "dense_embedding_grad", False):
            ctx.holder._shared_pending = (g, dist_.allreduce_sum_begin(g))
            return None, None
        return g, None


class GatherL2NormFn(Function):
    """y = table[ids] / |table[ids]| as bf16 (reference hstu.py:670-672, 752-754: gather the negative pools, then normalise),
    produced by EmbeddingGatherFn's own launch (mhr_embedding_gather_step) - the fp32 copy of the gathered rows that
    EmbeddingGatherFn + L2NormFn wrote and read back never exists (33.5 MB each way per step at cfg1); the values are bitwise
    theirs.  This Function is the autograd face of that half of the launch.  `rows_standin` is EmbeddingGatherFn's
    memory-less second output: the backward returns the gradient w.r.t. the gathered rows for it, so it reaches the sparse
    reduction of the embedding backward as d_negs.  The backward re-reads the rows from the table (no optimizer step lies
    between a step's forward and backward).  Data parallel with `holder`: as L2NormFn, the gradient leaves for its all-reduce
    from here."""

    @staticmethod
    def forward(ctx, rows_standin, table, ids, holder):
        got, holder._gathered_negs = getattr(holder, "_gathered_negs", None), None
        if got is None or got[0].shape[0] != ids.numel():
            raise RuntimeError("GatherL2NormFn: no gathered negatives on the holder (call it right behind EmbeddingGatherFn with negs_elsewhere)")
        y, norms = got
        ctx.save_for_backward(ids, norms)
        ctx.table = table
        ctx.holder = holder
        ctx.key = _open_f32_grad_slot(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, norms = ctx.saved_tensors
        slot = _F32_GRAD.pop(ctx.key, None)
        g32 = slot[1] if slot is not None else None
        if g32 is not None:                           # the loss left its fp32 gradient; dy is the placeholder unless others add to it
            dy = g32 if all(st == 0 for st in dy.stride()) else g32 + dy.float().reshape(g32.shape)
        g = ops.l2norm_rows_indexed_bwd(dy.reshape(ids.numel(), -1).float().contiguous(), ctx.table, ids, norms)
        ctx.table = None
        from mhr_amd import distributed as dist_
        if ctx.holder is not None and dist_.active() and dist_.OVERLAP and not getattr(ctx.holder, "dense_embedding_grad", False):
            ctx.holder._shared_pending = (g, dist_.allreduce_sum_begin(g))
            return None, None, None, None
        return g, None, None, None


class NceLossFn(Function):
    """Sampled-softmax loss of G groups (negative pools / prior categories) in one launch per kernel
    (reference hstu.py:600-619 + cross_entropy, called once per prior head there).

    q_rows [Rq, D] / p_rows [Rp, D] fp32 (raw head embeddings / target embeddings, shared by the groups); token t of
    group g pairs q_rows[q_idx[g,t]] with p_rows[p_idx[g,t]]; negs [G, n_neg, D] bf16 normalised.
    Without buckets: returns the per-token loss [G, tok_cap] fp32 (zeros beyond n_tok[g]).
    With bucket_idx [G, tok_cap] int32 / n_buckets: returns the MEAN loss per (group, bucket) [G, n_buckets] - the
    reference's per-offset mean (hstu.py:697-700) - straight from the finalize kernel; the backward turns d(mean) into
    per-bucket token weights, so no per-token tensor passes through torch.  No [N_tok, n_neg] tensor is ever materialised.
    share_rows: consecutive tokens of a group that use the same q row (the P prediction offsets of one position) share the
    negative-pool products (csrc/nce_shared.hip); same loss and gradients.
    """

    @staticmethod
    def forward(ctx, q_rows, p_rows, negs, logit_scale, q_idx, p_idx, n_tok_dev, tok_cap, thres, want_logs, logs_out,
                bucket_idx=None, n_buckets=0, log_group=-1, p_row_mask=None, share_rows=False, window=None, ihn_beta=0.0,
                prep=None, bucket_weight=None, n_segments=1, exclusive_q_rows=False):
        sv = ops.nce_fwd(q_rows, q_idx, p_rows, p_idx, negs, n_tok_dev, tok_cap, logit_scale.detach().view(1), thres,
                         want_logs=want_logs, for_backward=True, bucket_idx=bucket_idx, n_buckets=n_buckets,
                         log_group=log_group, p_row_mask=p_row_mask, share_rows=share_rows, window=window, ihn_beta=ihn_beta,
                         prep=prep)
        ctx.sv = sv
        ctx.exclusive_q_rows = bool(exclusive_q_rows)
        ctx.save_for_backward(q_idx, p_idx, logit_scale)
        ctx.shapes = (q_rows.shape, p_rows.shape)
        if want_logs and logs_out is not None:
            logs_out["n_valid"], logs_out["rank"] = sv.n_valid, sv.rank
        if bucket_idx is None:
            return sv.loss
        if logs_out is not None:
            logs_out["bucket_cnt"] = sv.bucket_cnt
        ctx.bucket_weight = bucket_weight
        if bucket_weight is not None:
            # (total, [per_gp | seg_all | g_tot | seg_sum]): the weighted total and the logged partial sums in ONE launch
            total, red = ops.loss_reduce(sv.bucket_sum, sv.bucket_cnt, bucket_weight, n_segments)
            ctx.mark_non_differentiable(red)
            return total, red
        return sv.bucket_sum / sv.bucket_cnt.clamp_min(1.0)

    @staticmethod
    def backward(ctx, d_out, _d_red=None):
        q_idx, p_idx, logit_scale = ctx.saved_tensors
        sv = ctx.sv
        q_shape, p_shape = ctx.shapes
        bufs, sv.bwd_bufs = getattr(sv, "bwd_bufs", None), None
        d_negs0 = d_ls0 = lw_row = None
        if bufs is not None and bufs[0].shape == q_shape and bufs[1].shape == p_shape:     # zeroed early (ops.nce_shared_prepare)
            dq, dp, d_negs0, d_ls0, lw_row = bufs
            if not ctx.needs_input_grad[2]:
                d_negs0 = None
        else:
            dq = torch.zeros(q_shape, dtype=torch.float32, device=d_out.device)
            dp = torch.zeros(p_shape, dtype=torch.float32, device=d_out.device)
        if ctx.bucket_weight is not None:
            w = ops.loss_reduce_bwd(d_out, sv.bucket_cnt, ctx.bucket_weight)      # d(total)/d(loss_t) = weight / count of the bucket
        else:
            w = d_out.contiguous().float()
            if sv.bucket_idx is not None:
                w = w / sv.bucket_cnt.clamp_min(1.0)                      # d(mean)/d(loss_t) = 1 / count of the bucket
        d_negs, d_ls = ops.nce_bwd(sv, w, logit_scale.detach().view(1), q_idx, p_idx, dq, dp, d_negs=d_negs0, d_logit_scale=d_ls0,
                                   want_negs=ctx.needs_input_grad[2], lw_row=lw_row, exclusive_q_rows=ctx.exclusive_q_rows)
        ctx.sv = None
        if d_negs is not None:
            d_negs = _deposit_f32_grad(sv.negs, d_negs)
        return (dq, dp, d_negs, d_ls.view(logit_scale.shape)) + (None,) * 18

"""Thin torch-facing wrappers over the C ABI (include/mhr.h).

PyTorch is plumbing here: device memory (`Tensor.data_ptr()`), the current HIP stream and
`torch.distributed`.  Every function enqueues hand-written gfx950 kernels from libmhr_hip.so on the
current stream; none of them synchronises, and none has a CPU or eager fallback.
"""
import os

import torch

from . import lib

F32, BF16 = lib.F32, lib.BF16

# Optional per-launch timing (bench.py): when PROFILE is a dict, calls named in it are bracketed by HIP events
# recorded on the stream the kernel is launched on; durations are read after a sync by `profile_summary`.
PROFILE = None


def _timed_call(name, *args):
    if PROFILE is not None and name in PROFILE:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.call(name, *args)
        e1.record()
        PROFILE[name].append((e0, e1))
    else:
        lib.call(name, *args)


def profile_summary(with_max=False):
    """name -> (launches, mean ms, total ms[, max ms]); call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in (PROFILE or {}).items():
        if evs:
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms) / len(ms), sum(ms)) + ((max(ms),) if with_max else ())
    return out


def profile_raw(name):
    """Every measured duration (ms) of one entry point, in launch order."""
    return [a.elapsed_time(b) for a, b in (PROFILE or {}).get(name, [])]


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())




def zeros_many(dev, *specs):
    """Several zero-initialised 4-byte tensors from ONE allocation and ONE fill launch (each small `torch.zeros` is its own
    fill kernel: about 4.5 us of device timeline apiece, and a training step used to issue some forty of them).
    specs: (shape, dtype) with dtype float32 / int32; every tensor starts 16-byte aligned."""
    sizes = []
    for shape, dtype in specs:
        n = 1
        for d in shape:
            n *= int(d)
        sizes.append((n + 3) // 4 * 4)
    buf = torch.zeros(max(sum(sizes), 4), dtype=torch.int32, device=dev)
    out, off = [], 0
    for (shape, dtype), sz in zip(specs, sizes):
        n = 1
        for d in shape:
            n *= int(d)
        t = buf[off:off + n]
        out.append((t if dtype == torch.int32 else t.view(dtype)).view(*shape))
        off += sz
    return out


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor; the MI355X path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


# ------------------------------------------------------------------------------------------------
# embedding
# ------------------------------------------------------------------------------------------------
def embedding_gather(table, ids, out_dtype=torch.bfloat16, pos_table=None, seq_len=0, x_dtype=torch.float32,
                     want_rows=True, out=None, window=None, n_x_ids=0):
    """rows = table[ids] (-> [*ids.shape, D]); with pos_table also x = table[ids[:, :seq_len]] + pos[:seq_len].
    `out`: optional preallocated contiguous [ids.numel(), D] destination (e.g. a slice of a larger row matrix)."""
    _chk(table, "table", torch.float32)
    _chk(ids, "ids", torch.int64)
    D = table.shape[1]
    n = ids.numel()
    if out is not None:
        _chk(out, "out")
        assert out.numel() == n * D
    elif want_rows:
        out = torch.empty(*ids.shape, D, dtype=out_dtype, device=table.device)
    x = None
    # `window` / `n_x_ids`: flat ids whose first n_x_ids entries are [n_x_ids / window, window] item windows (they get
    # the position-added x), followed by ids that only need their rows (the negative pools): one launch for both
    if window is None:
        window = ids.shape[-1] if pos_table is not None else 0
    if pos_table is not None:
        _chk(pos_table, "pos_table", torch.float32)
        if n_x_ids:
            x = torch.empty(n_x_ids // window, seq_len, D, dtype=x_dtype, device=table.device)
        else:
            x = torch.empty(*ids.shape[:-1], seq_len, D, dtype=x_dtype, device=table.device)
    _timed_call("mhr_embedding_gather_fwd", table.data_ptr(), table.shape[0], D, ids.data_ptr(), n,
             _ptr(out), _dt(out) if out is not None else F32, _ptr(pos_table), seq_len, window,
             _ptr(x), _dt(x) if x is not None else F32, int(n_x_ids), _stream())
    return out, x


def embedding_scatter_add(grad_rows, ids, grad_table):
    _chk(grad_rows, "grad_rows")
    _chk(ids, "ids", torch.int64)
    _chk(grad_table, "grad_table", torch.float32)
    lib.call("mhr_embedding_scatter_add_bwd", grad_rows.data_ptr(), _dt(grad_rows), ids.data_ptr(), ids.numel(),
             grad_table.data_ptr(), grad_table.shape[0], grad_table.shape[1], _stream())
    return grad_table


def sparse_rows_segment_sum(sorted_ids, perm, grad_a, grad_b, x_grad, seq_len, window_len, out_rows, row_slot):
    D = out_rows.shape[-1]
    n_a = 0 if grad_a is None else grad_a.numel() // D
    n_b = 0 if grad_b is None else grad_b.numel() // D
    _timed_call("mhr_sparse_rows_segment_sum", sorted_ids.data_ptr(), perm.data_ptr(), sorted_ids.numel(),
             _ptr(grad_a), _dt(grad_a) if grad_a is not None else F32, n_a,
             _ptr(grad_b), _dt(grad_b) if grad_b is not None else F32, n_b,
             _ptr(x_grad), seq_len, window_len, out_rows.data_ptr(), row_slot.data_ptr(), row_slot.numel(), D, _stream())


def adam_rows(w, m, v, grad_rows, row_slot, step, lr, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    _timed_call("mhr_adam_rows", w.data_ptr(), m.data_ptr(), v.data_ptr(), w.shape[0], w.shape[1], grad_rows.data_ptr(),
             _ptr(row_slot), grad_scale, lr, betas[0], betas[1], eps, weight_decay, step, _stream())


def adam_flat(w, g, m, v, step, lr, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_bf16=None,
              hist=None, step_dev=None, hist_len=64):
    """hist [hist_len, 4] f32 + step_dev int64[1] (device): the step's constants come from device memory (hipGraph replay)."""
    lib.call("mhr_adam_flat", w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), grad_scale, lr,
             betas[0], betas[1], eps, weight_decay, step, _ptr(w_bf16), _ptr(hist), 0 if hist is None else hist_len,
             _ptr(step_dev), _stream())


def heads_residual_fwd(x, z, B, L, H):
    """out [B, H, L, D] f32 = x[b, l] + silu(z[b, l, h]);  x [B*L, D] f32, z [B*L, H*D] bf16."""
    _chk(x, "x", torch.float32)
    _chk(z, "z", torch.bfloat16)
    D = x.shape[-1]
    out = torch.empty(B, H, L, D, dtype=torch.float32, device=x.device)
    lib.call("mhr_heads_residual_fwd", x.data_ptr(), z.data_ptr(), out.data_ptr(), B * L, L, H, D, _stream())
    return out


def heads_residual_bwd(d_out, z, B, L, H):
    """(dz [B*L, H*D] bf16, dx [B*L, D] f32) from d_out [B, H, L, D] f32."""
    _chk(d_out, "d_out", torch.float32)
    D = d_out.shape[-1]
    dz = torch.empty_like(z)
    dx = torch.empty(B * L, D, dtype=torch.float32, device=z.device)
    lib.call("mhr_heads_residual_bwd", d_out.data_ptr(), z.data_ptr(), dz.data_ptr(), dx.data_ptr(), B * L, L, H, D, _stream())
    return dz, dx


def sum_rows_many(ptr_table, n, rows, cols, keep=None):
    """out_i += x_i.sum(0) for n equally shaped bf16 matrices in one launch; ptr_table: device int64 [2 n] (sources, then fp32
    destinations).  `keep`: the tensors behind the addresses (held by the caller until the launch is enqueued)."""
    _chk(ptr_table, "ptr_table", torch.int64)
    _timed_call("mhr_sum_rows_many", ptr_table.data_ptr(), int(n), int(rows), int(cols), _stream())


def bad_id_count(reset=True):
    """Ids outside the table that embedding gathers met since the last reset (they are clamped on the device, where the
    reference's nn.Embedding raises).  Synchronises: call it where the host reads device results anyway."""
    import ctypes
    c = ctypes.c_int64(0)
    lib.call("mhr_bad_id_count", ctypes.addressof(c), 1 if reset else 0)
    return int(c.value)


def sum_rows_into(x, out):
    """out [cols] fp32 += column sums of x [rows, cols] bf16 or fp32 (contiguous)."""
    _chk(out, "out", torch.float32)
    cols = out.numel()
    assert x.numel() % cols == 0
    if x.dtype == torch.float32:
        _chk(x, "x", torch.float32)
        lib.call("mhr_sum_rows_f32_into", x.data_ptr(), x.numel() // cols, cols, out.data_ptr(), _stream())
        return out
    _chk(x, "x", torch.bfloat16)
    _timed_call("mhr_sum_rows_into", x.data_ptr(), x.numel() // cols, cols, out.data_ptr(), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# normalisation / gate
# ------------------------------------------------------------------------------------------------
def _out_like(out, shape, dtype, device, name):
    """`out` (a caller's buffer: same element count and dtype, contiguous) or a fresh tensor."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= int(d)
    if out.dtype != dtype or out.numel() != n or not out.is_contiguous() or out.device != device:
        raise ValueError(f"{name}: expected a contiguous {dtype} buffer of {n} elements on {device}")
    return out


def layernorm_fwd(x, out_dtype=torch.bfloat16, eps=1e-6, out=None):
    _chk(x, "x")
    D = x.shape[-1]
    rows = x.numel() // D
    y = _out_like(out, x.shape, out_dtype, x.device, "layernorm_fwd out")
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib.call("mhr_layernorm_fwd", x.data_ptr(), _dt(x), y.data_ptr(), _dt(y), mean.data_ptr(), rstd.data_ptr(), rows, D,
             eps, _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, dx=None, accumulate=False, dx_dtype=torch.float32):
    _chk(dy, "dy")
    D = x.shape[-1]
    rows = x.numel() // D
    if dx is None:
        dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)
    lib.call("mhr_layernorm_bwd", dy.data_ptr(), _dt(dy), x.data_ptr(), _dt(x), mean.data_ptr(), rstd.data_ptr(),
             dx.data_ptr(), _dt(dx), 1 if accumulate else 0, rows, D, _stream())
    return dx


def add_cast(x, y, out16=None):
    """(x + y fp32, its bf16 copy) for x fp32 and y bf16 of one shape, one pass (mhr_add_cast)."""
    _chk(x, "x", torch.float32)
    _chk(y, "y", torch.bfloat16)
    assert x.shape == y.shape
    out = torch.empty_like(x)
    out16 = _out_like(out16, y.shape, torch.bfloat16, y.device, "add_cast out16")
    lib.call("mhr_add_cast", x.data_ptr(), y.data_ptr(), out.data_ptr(), out16.data_ptr(), x.numel(), _stream())
    return out, out16


def _dead_args(dead, rows):
    """(first_row pointer, seq_len) of a `dead=(first_row [B] int32, L)` row-liveness descriptor (attn_seq_layout): rows in front
    of a sequence's first valid key are not loaded by the row-wise encoder kernels (they read as zeros, zeros are written)."""
    if dead is None:
        return 0, 0
    first_row, L = dead
    if first_row.dtype != torch.int32 or first_row.numel() * L != rows:
        raise ValueError("dead rows: first_row must be int32 [B] with B * L == rows")
    return first_row.data_ptr(), int(L)


def add_layernorm_fwd(x, y, eps=1e-6, xn_out=None, dead=None):
    """x_out = x + y (fp32 + bf16), xn = LN(x_out) bf16.  Returns (x_out, xn, mean, rstd)."""
    _chk(x, "x", torch.float32)
    _chk(y, "y", torch.bfloat16)
    D = x.shape[-1]
    rows = x.numel() // D
    x_out = torch.empty_like(x)
    xn = _out_like(xn_out, x.shape, torch.bfloat16, x.device, "add_layernorm_fwd xn_out")
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib.call("mhr_add_layernorm_fwd", x.data_ptr(), y.data_ptr(), x_out.data_ptr(), xn.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), rows, D, eps, *_dead_args(dead, rows), _stream())
    return x_out, xn, mean, rstd


def add_layernorm_bwd(d_xn, x_out, mean, rstd, d_xout, dy_out=None, dead=None):
    """-> (dx f32, dy bf16), both = d_xout + LN'(d_xn)."""
    _chk(d_xn, "d_xn", torch.bfloat16)
    _chk(d_xout, "d_xout", torch.float32)
    D = x_out.shape[-1]
    rows = x_out.numel() // D
    dx = torch.empty_like(x_out)
    dy = _out_like(dy_out, x_out.shape, torch.bfloat16, x_out.device, "add_layernorm_bwd dy_out")
    lib.call("mhr_add_layernorm_bwd", d_xn.data_ptr(), x_out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), d_xout.data_ptr(),
             dx.data_ptr(), dy.data_ptr(), rows, D, *_dead_args(dead, rows), _stream())
    return dx, dy


def ln_gate_fwd(h, a, dim, out_dtype=None, eps=1e-6, dropout_p=0.0, seed=0, seed_dev=None, out=None, dead=None):
    """o = silu(h[:, :dim]) * LN(a) * dropmask.  h [rows, stride] pre-activation, a [rows, dim].
    seed_dev (device int64[1], optional): step counter of a hipGraph-replayed step; `seed` is then the per-layer part."""
    rows = a.numel() // dim
    o = _out_like(out, (rows, dim), out_dtype or a.dtype, a.device, "ln_gate_fwd out").view(rows, dim)
    mean = torch.empty(rows, dtype=torch.float32, device=a.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
    assert h.dtype == a.dtype
    lib.call("mhr_ln_gate_fwd", h.data_ptr(), h.stride(0), a.data_ptr(), _dt(a), o.data_ptr(), _dt(o), mean.data_ptr(),
             rstd.data_ptr(), rows, dim, eps, dropout_p, seed, _ptr(seed_dev), *_dead_args(dead, rows), _stream())
    return o, mean, rstd


def ln_gate_bwd(d_o, h, a, mean, rstd, dh, dim, dropout_p=0.0, seed=0, seed_dev=None, dead=None):
    """writes du into dh[:, :dim]; returns da."""
    rows = a.numel() // dim
    da = torch.empty_like(a)
    lib.call("mhr_ln_gate_bwd", d_o.data_ptr(), _dt(d_o), h.data_ptr(), h.stride(0), a.data_ptr(), _dt(a), mean.data_ptr(),
             rstd.data_ptr(), dh.data_ptr(), dh.stride(0), da.data_ptr(), rows, dim, dropout_p, seed, _ptr(seed_dev),
             *_dead_args(dead, rows), _stream())
    return da


def l2norm_rows(x, out_dtype=torch.bfloat16, want_norms=False):
    _chk(x, "x")
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    norms = torch.empty(rows, dtype=torch.float32, device=x.device) if want_norms else None
    lib.call("mhr_l2norm_rows", x.data_ptr(), _dt(x), y.data_ptr(), _dt(y), _ptr(norms), rows, D, _stream())
    return (y, norms) if want_norms else y


def l2norm_rows_bwd(dy, x, norms):
    """dx of y = x / |x| (fp32 rows)."""
    _chk(dy, "dy", torch.float32)
    _chk(x, "x", torch.float32)
    D = x.shape[-1]
    dx = torch.empty_like(x)
    lib.call("mhr_l2norm_rows_bwd", dy.data_ptr(), x.data_ptr(), norms.data_ptr(), dx.data_ptr(), x.numel() // D, D, _stream())
    return dx


def embedding_gather_step(table, pos_table, ids_all, n_item_ids, seq_len, window):
    """One launch for a training step's table reads: item windows -> (rows fp32 [n_item_ids, D], x fp32 [B, seq_len, D] with the
    position add); negative-pool ids -> (normalised bf16 rows [n_neg, D], norms fp32 [n_neg])."""
    _chk(table, "table", torch.float32)
    _chk(pos_table, "pos_table", torch.float32)
    _chk(ids_all, "ids_all", torch.int64)
    D = table.shape[1]
    dev = table.device
    n = ids_all.numel()
    n_neg = n - n_item_ids
    rows = torch.empty(n_item_ids, D, dtype=torch.float32, device=dev)
    x = torch.empty(n_item_ids // window, seq_len, D, dtype=torch.float32, device=dev)
    negs = torch.empty(n_neg, D, dtype=torch.bfloat16, device=dev)
    norms = torch.empty(n_neg, dtype=torch.float32, device=dev)
    _timed_call("mhr_embedding_gather_step", table.data_ptr(), table.shape[0], D, ids_all.data_ptr(), n, n_item_ids, rows.data_ptr(),
                pos_table.data_ptr(), seq_len, window, x.data_ptr(), negs.data_ptr(), norms.data_ptr(), _stream())
    return rows, x, negs, norms


def l2norm_rows_indexed_bwd(dy, table, ids, norms):
    """dx of y = table[ids] / |table[ids]| w.r.t. the gathered rows (fp32 [n, D]); the rows are re-read from the table."""
    _chk(dy, "dy", torch.float32)
    _chk(table, "table", torch.float32)
    _chk(ids, "ids", torch.int64)
    D = table.shape[1]
    dx = torch.empty(ids.numel(), D, dtype=torch.float32, device=table.device)
    lib.call("mhr_l2norm_rows_indexed_bwd", dy.data_ptr(), table.data_ptr(), table.shape[0], ids.data_ptr(), norms.data_ptr(),
             dx.data_ptr(), ids.numel(), D, _stream())
    return dx


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def rows_gemm_supported(M, N, K, w_is_kn=False):
    return K in (64, 128, 256) and N > 0 and N % (256 if w_is_kn else 8) == 0 and M > 0          # (= mhr_rows_gemm_supported)


def rows_gemm(a, w, bias=None, out=None, w_is_kn=False):
    """out [M, N] bf16 = a [M, K] @ w.T (w [N, K]; w_is_kn: a @ w with w [K, N]) (+ bias [N] bf16): the encoder's token-rows
    projections (hstu.py:236-239, 281-288) with the weight stationary in registers and the token rows streaming through LDS."""
    for t_, nm in ((a, "a"), (w, "w")):                       # (rows may be a column block of a wider buffer: row strides are passed on)
        if not (t_.is_cuda and t_.dtype == torch.bfloat16 and t_.dim() == 2 and t_.stride(1) == 1):
            raise ValueError(f"rows_gemm: {nm} must be a 2-d bf16 device tensor with unit inner stride")
    M, K = a.shape
    N = w.shape[1] if w_is_kn else w.shape[0]
    if (w.shape[0] if w_is_kn else w.shape[1]) != K:
        raise ValueError("rows_gemm: inner dimensions differ")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    elif out.shape != (M, N) or out.dtype != torch.bfloat16 or out.stride(1) != 1:
        raise ValueError("rows_gemm: out must be [M, N] bf16 with unit inner stride")
    if bias is not None:
        _chk(bias, "bias", torch.bfloat16)
    _timed_call("mhr_rows_gemm", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), 1 if w_is_kn else 0,
                bias.data_ptr() if bias is not None else 0, out.data_ptr(), out.stride(0), M, N, K, _stream())
    return out


def seq_pack_maps(key_valid, B, L, capacity):
    """(cu_rows [B+1], src_of [capacity], row_of [B*L], overflow [1]) int32 of a batch of masks (mhr_seq_pack_maps)."""
    _chk(key_valid, "key_valid", torch.uint8)
    dev = key_valid.device
    cu = torch.empty(B + 1, dtype=torch.int32, device=dev)
    src_of = torch.empty(capacity, dtype=torch.int32, device=dev)
    row_of = torch.empty(B * L, dtype=torch.int32, device=dev)
    overflow = torch.empty(1, dtype=torch.int32, device=dev)
    lib.call("mhr_seq_pack_maps", key_valid.data_ptr(), B, L, int(capacity), cu.data_ptr(), src_of.data_ptr(), row_of.data_ptr(),
             overflow.data_ptr(), _stream())
    return cu, src_of, row_of, overflow


def rows_gather_masked(src, idx, out=None):
    """out[r] = src[idx[r]] where idx[r] >= 0, zeros elsewhere (rows of f32 or bf16; idx int32)."""
    _chk(src, "src")
    _chk(idx, "idx", torch.int32)
    n, dim = idx.numel(), src.shape[-1]
    if out is None:
        out = torch.empty(n, dim, dtype=src.dtype, device=src.device)
    lib.call("mhr_rows_gather_masked", src.data_ptr(), _dt(src), idx.data_ptr(), out.data_ptr(), n, dim, _stream())
    return out


PACK_ROWS = os.environ.get("MHR_PACK_ROWS", "1") != "0"          # encoder over the valid rows only when the batch carries a row capacity
DEAD_ROWS = os.environ.get("MHR_DEAD_ROWS", "1") != "0"          # row-wise encoder kernels do not load rows in front of a sequence's first valid key
SEQ_LAYOUT = os.environ.get("MHR_ATTN_SEQ_LAYOUT", "1") != "0"   # skip leading all-padding blocks + longest-sequences-first launch order


def attn_seq_layout(key_valid, B, L, order=True):
    """(first_block [B] int32, seq_order [B] int32 | None, first_row [B] int32) of a batch of masks: the 32-row block holding each
    sequence's first valid key, the sequences ordered by it (most live blocks first), and the index of that key (L when none).
    Computed once per batch, read by every layer's attention launches (`layout=` of hstu_attn_fwd / hstu_attn_bwd) and row-wise
    kernels (`dead=(first_row, L)`)."""
    _chk(key_valid, "key_valid", torch.uint8)
    first = torch.empty(B, dtype=torch.int32, device=key_valid.device)
    first_row = torch.empty(B, dtype=torch.int32, device=key_valid.device)
    order_t = torch.empty(B, dtype=torch.int32, device=key_valid.device) if order else None
    lib.call("mhr_attn_seq_layout", key_valid.data_ptr(), B, L, first.data_ptr(), order_t.data_ptr() if order else 0,
             first_row.data_ptr(), _stream())
    return first, order_t, first_row


def _layout_ptrs(layout):
    """(first_block, seq_order, cu_rows) pointers of a layout tuple (first_block | None, seq_order | None, first_row, cu_rows)."""
    if layout is None:
        return 0, 0, 0
    first, order = layout[0], layout[1]
    cu = layout[3] if len(layout) > 3 else None
    return (first.data_ptr() if first is not None else 0), (order.data_ptr() if order is not None else 0), (cu.data_ptr() if cu is not None else 0)


def hstu_attn_fwd(h, key_valid, B, L, n_heads, head_dim, apply_silu=True, save_act=True, layout=None):
    """h [B*L, 4*D] bf16 pre-activation uvqk (column blocks u|v|q|k).  Returns (out [B*L, D], act [B*L, 3*D] (q|k|v))."""
    _chk(h, "h", torch.bfloat16)
    _chk(key_valid, "key_valid", torch.uint8)
    D = n_heads * head_dim
    stride = h.stride(0)
    esz = 2
    base = h.data_ptr()
    v_ptr, q_ptr, k_ptr = base + D * esz, base + 2 * D * esz, base + 3 * D * esz
    R = h.shape[0]                               # B * L windows, or the capacity of a packed batch (layout[3] = cu_rows)
    out = torch.empty(R, D, dtype=torch.bfloat16, device=h.device)
    act = torch.empty(R, 3 * D, dtype=torch.bfloat16, device=h.device) if save_act else None
    aq = act.data_ptr() if save_act else 0
    fb, so, cu = _layout_ptrs(layout)
    _timed_call("mhr_hstu_attn_fwd_seq", q_ptr, k_ptr, v_ptr, stride, key_valid.data_ptr(), out.data_ptr(),
             aq, aq + D * esz if save_act else 0, aq + 2 * D * esz if save_act else 0, 3 * D,
             B, L, n_heads, head_dim, 1 if apply_silu else 0, fb, so, cu, R if cu else 0, _stream())
    return out, act


def hstu_attn_bwd(h, act, key_valid, d_out, dh, B, L, n_heads, head_dim, apply_silu=True, layout=None):
    """writes dv|dq|dk into dh[:, D:4D] (pre-activation gradients).  act = None: the activated q / k / v are recomputed
    from h inside the kernel (apply_silu only)."""
    D = n_heads * head_dim
    esz = 2
    base, dbase = h.data_ptr(), dh.data_ptr()
    abase = act.data_ptr() if act is not None else 0
    fb, so, cu = _layout_ptrs(layout)
    _timed_call("mhr_hstu_attn_bwd_seq", base + 2 * D * esz, base + 3 * D * esz, base + D * esz, h.stride(0),
             abase, abase + D * esz if act is not None else 0, abase + 2 * D * esz if act is not None else 0,
             act.stride(0) if act is not None else 0, key_valid.data_ptr(), d_out.data_ptr(),
             dbase + 2 * D * esz, dbase + 3 * D * esz, dbase + D * esz, dh.stride(0),
             B, L, n_heads, head_dim, 1 if apply_silu else 0, fb, so, cu, h.shape[0] if cu else 0, _stream())
    return dh


# ------------------------------------------------------------------------------------------------
# LLM decoder blocks (HLLM twin): RMSNorm, SwiGLU, RoPE, causal softmax attention
# ------------------------------------------------------------------------------------------------
def rmsnorm_fwd(x, weight, res=None, eps=1e-6):
    """x f32 [rows, D] (+ res bf16) -> (x_out f32 (x itself without res), y bf16, rstd f32 [rows])."""
    _chk(x, "x", torch.float32)
    _chk(weight, "weight", torch.float32)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    x_out = x
    if res is not None:
        _chk(res, "res", torch.bfloat16)
        x_out = torch.empty_like(x)
    _timed_call("mhr_rmsnorm_fwd", x.data_ptr(), _ptr(res), weight.data_ptr(), x_out.data_ptr(), y.data_ptr(), rstd.data_ptr(),
                rows, D, float(eps), _stream())
    return x_out, y, rstd


def rmsnorm_bwd(dy, x_out, weight, rstd, d_xout=None, want_dres=False):
    """-> (dx f32, dres bf16 or None, dw f32 [D])."""
    _chk(dy, "dy", torch.bfloat16)
    _chk(x_out, "x_out", torch.float32)
    D = x_out.shape[-1]
    rows = x_out.numel() // D
    dx = torch.empty_like(x_out)
    dres = torch.empty(x_out.shape, dtype=torch.bfloat16, device=x_out.device) if want_dres else None
    parts = lib.load().mhr_rmsnorm_bwd_parts(rows)
    dw_part = torch.empty(parts, D, dtype=torch.float32, device=x_out.device)
    if d_xout is not None:
        _chk(d_xout, "d_xout", torch.float32)
    _timed_call("mhr_rmsnorm_bwd", dy.data_ptr(), x_out.data_ptr(), weight.data_ptr(), rstd.data_ptr(), _ptr(d_xout),
                dx.data_ptr(), _ptr(dres), dw_part.data_ptr(), rows, D, _stream())
    return dx, dres, dw_part.sum(0)


def swiglu_fwd(gate_up):
    _chk(gate_up, "gate_up", torch.bfloat16)
    F2 = gate_up.shape[-1]
    rows = gate_up.numel() // F2
    act = torch.empty(*gate_up.shape[:-1], F2 // 2, dtype=torch.bfloat16, device=gate_up.device)
    _timed_call("mhr_swiglu_fwd", gate_up.data_ptr(), act.data_ptr(), rows, F2 // 2, _stream())
    return act


def swiglu_bwd(gate_up, d_act):
    _chk(gate_up, "gate_up", torch.bfloat16)
    _chk(d_act, "d_act", torch.bfloat16)
    F2 = gate_up.shape[-1]
    rows = gate_up.numel() // F2
    d = torch.empty_like(gate_up)
    _timed_call("mhr_swiglu_bwd", gate_up.data_ptr(), d_act.data_ptr(), d.data_ptr(), rows, F2 // 2, _stream())
    return d


def rope_inplace(x, n_heads, head_dim, cos, sin, positions=None, seq_len=0, inverse=False):
    """Rotate the first n_heads heads of every row of x [T, stride] bf16 in place."""
    _chk(x, "x", torch.bfloat16)
    _chk(cos, "cos", torch.float32)
    _chk(sin, "sin", torch.float32)
    if positions is not None:
        _chk(positions, "positions", torch.int32)
    assert cos.shape == sin.shape and cos.shape[1] == head_dim // 2 and n_heads * head_dim <= x.shape[-1]
    T = x.numel() // x.shape[-1]
    _timed_call("mhr_rope_inplace", x.data_ptr(), x.shape[-1], _ptr(positions), cos.data_ptr(), sin.data_ptr(), T, int(seq_len),
                n_heads, head_dim, cos.shape[0], 1 if inverse else 0, _stream())
    return x


def softmax_attn_fwd(qkv, n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale, cu_seqlens=None, key_valid=None):
    """qkv [T, (n_heads + 2 n_kv_heads) * head_dim] bf16 (q | k | v column blocks) -> (out [T, n_heads*head_dim] bf16, lse)."""
    _chk(qkv, "qkv", torch.bfloat16)
    T, stride = qkv.shape
    assert stride == (n_heads + 2 * n_kv_heads) * head_dim
    if cu_seqlens is not None:
        _chk(cu_seqlens, "cu_seqlens", torch.int32)
        assert cu_seqlens.numel() == n_seqs + 1
    else:
        assert T == n_seqs * max_len
    if key_valid is not None:
        _chk(key_valid, "key_valid", torch.uint8)
        assert key_valid.numel() == T
    out = torch.empty(T, n_heads * head_dim, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(T, n_heads, dtype=torch.float32, device=qkv.device)
    e = qkv.element_size()
    q, k = qkv.data_ptr(), qkv.data_ptr() + n_heads * head_dim * e
    v = k + n_kv_heads * head_dim * e
    _timed_call("mhr_softmax_attn_fwd", q, k, v, stride, _ptr(cu_seqlens), _ptr(key_valid), out.data_ptr(), lse.data_ptr(),
                n_seqs, max_len, n_heads, n_kv_heads, head_dim, float(scale), _stream())
    return out, lse


def softmax_attn_bwd(qkv, out, d_out, lse, n_seqs, max_len, n_heads, n_kv_heads, head_dim, scale, cu_seqlens=None,
                     key_valid=None):
    """-> dqkv [T, (n_heads + 2 n_kv_heads) * head_dim] bf16 (KV-head gradients summed over their query-head group in fp32)."""
    _chk(qkv, "qkv", torch.bfloat16)
    _chk(out, "out", torch.bfloat16)
    _chk(d_out, "d_out", torch.bfloat16)
    T, stride = qkv.shape
    dqkv = torch.empty_like(qkv)
    e = qkv.element_size()
    q, k = qkv.data_ptr(), qkv.data_ptr() + n_heads * head_dim * e
    v = k + n_kv_heads * head_dim * e
    group = n_heads // n_kv_heads
    if group == 1:
        dk_ptr = dqkv.data_ptr() + n_heads * head_dim * e
        dv_ptr = dk_ptr + n_kv_heads * head_dim * e
        dkv_stride, slabs = stride, None
    else:
        slabs = torch.empty(2, T, n_heads * head_dim, dtype=torch.bfloat16, device=qkv.device)
        dk_ptr, dv_ptr, dkv_stride = slabs[0].data_ptr(), slabs[1].data_ptr(), n_heads * head_dim
    _timed_call("mhr_softmax_attn_bwd", q, k, v, stride, _ptr(cu_seqlens), _ptr(key_valid), out.data_ptr(), d_out.data_ptr(),
                lse.data_ptr(), dqkv.data_ptr(), stride, dk_ptr, dv_ptr, dkv_stride, n_seqs, max_len, n_heads, n_kv_heads,
                head_dim, float(scale), _stream())
    if slabs is not None:
        red = torch.sum(slabs.view(2, T, n_kv_heads, group, head_dim), dim=3, dtype=torch.float32)      # [2, T, n_kv, hd]
        dqkv[:, n_heads * head_dim:].view(T, 2, n_kv_heads * head_dim).copy_(red.view(2, T, -1).transpose(0, 1))
    return dqkv


# ------------------------------------------------------------------------------------------------
# sampled softmax
# ------------------------------------------------------------------------------------------------
def token_compact(mask, q_all, p_all, o_all, tok_cap=None, slot_map=False):
    """Ordered compaction of live (group, slot) pairs.  mask [G, n_slots] bool/uint8; q_all [G, n_slots] int32;
    p_all, o_all [n_slots] int32.  Returns (q_idx, p_idx, o_idx [G, cap] int32 - entries beyond n_tok undefined -,
    n_tok [G] int32) with cap = tok_cap (default n_slots) rounded up to a multiple of 32.  No host sync.
    slot_map: also return tok_of_slot [G, n_slots] int32 (list position of a live slot, -1 otherwise)."""
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    _chk(mask, "mask", torch.uint8)
    _chk(q_all, "q_all", torch.int32)
    _chk(p_all, "p_all", torch.int32)
    _chk(o_all, "o_all", torch.int32)
    G, n_slots = mask.shape
    assert q_all.shape == (G, n_slots) and p_all.numel() == n_slots and o_all.numel() == n_slots
    cap = ((n_slots if tok_cap is None else tok_cap) + 31) // 32 * 32
    dev = mask.device
    q_idx = torch.empty(G, cap, dtype=torch.int32, device=dev)
    p_idx = torch.empty(G, cap, dtype=torch.int32, device=dev)
    o_idx = torch.empty(G, cap, dtype=torch.int32, device=dev)
    n_tok = torch.empty(G, dtype=torch.int32, device=dev)
    scratch = torch.empty(G, (n_slots + 4095) // 4096, dtype=torch.int32, device=dev)
    tos = torch.empty(G, n_slots, dtype=torch.int32, device=dev) if slot_map else None
    lib.call("mhr_token_compact", mask.data_ptr(), q_all.data_ptr(), p_all.data_ptr(), o_all.data_ptr(), G, n_slots, cap,
             q_idx.data_ptr(), p_idx.data_ptr(), o_idx.data_ptr(), n_tok.data_ptr(), scratch.data_ptr(), _ptr(tos), _stream())
    if slot_map:
        return q_idx, p_idx, o_idx, n_tok, tos
    return q_idx, p_idx, o_idx, n_tok


def loss_reduce(bucket_sum, bucket_cnt, weight, n_segments):
    """[G,P] per-offset loss sums / token counts / weights -> (total [], flat [G P + G S + G + S] = per_gp | seg_all | g_tot | seg_sum)
    (mhr_loss_reduce: one launch for the mean, the weighting and every logged partial sum)."""
    for t, n in ((bucket_sum, "bucket_sum"), (bucket_cnt, "bucket_cnt"), (weight, "weight")):
        _chk(t, n, torch.float32)
    G, P = bucket_sum.shape
    assert bucket_cnt.shape == (G, P) and weight.shape == (G, P) and P % n_segments == 0
    total = torch.empty((), dtype=torch.float32, device=bucket_sum.device)
    out = torch.empty(G * P + G * n_segments + G + n_segments, dtype=torch.float32, device=bucket_sum.device)
    lib.call("mhr_loss_reduce", bucket_sum.data_ptr(), bucket_cnt.data_ptr(), weight.data_ptr(), G, P, int(n_segments),
             total.data_ptr(), out.data_ptr(), _stream())
    return total, out


def loss_reduce_bwd(d_total, bucket_cnt, weight):
    """Token weight per (group, offset) bucket for the backward kernels: d_total * weight / max(cnt, 1)."""
    d_total = d_total.reshape(1).float().contiguous()
    w = torch.empty_like(weight)
    lib.call("mhr_loss_reduce_bwd", d_total.data_ptr(), bucket_cnt.data_ptr(), weight.data_ptr(), weight.numel(), w.data_ptr(), _stream())
    return w


def nce_log_counters(n_valid, rank, o_idx, n_tok_dev, group, ks):
    """-> f32 [1 + len(ks)]: mean n_valid and mean(rank < k) over the live offset-0 tokens of `group` (hstu.py:621-629)."""
    import ctypes
    G, cap = o_idx.shape
    for t in (n_valid, rank, o_idx):
        _chk(t, "nce_log_counters input", torch.int32) if t.is_contiguous() else None
        assert t.shape == (G, cap) and t.stride(0) == t.shape[1] * t.stride(1) or t.is_contiguous(), "token tables must share one [G, cap] layout"
        assert t.stride(0) == cap and t.stride(1) == 1 and t.dtype == torch.int32
    out = torch.empty(1 + len(ks), dtype=torch.float32, device=o_idx.device)
    ks_arr = (ctypes.c_int32 * max(1, len(ks)))(*ks)
    key = str(o_idx.device)
    if key not in _COUNTER_SCRATCH:       # self-cleaning (the kernel zeroes it again), one per device, never freed: graph-safe
        _COUNTER_SCRATCH[key] = torch.zeros(8, dtype=torch.int64, device=o_idx.device)
    lib.call("mhr_nce_log_counters", n_valid.data_ptr(), rank.data_ptr(), o_idx.data_ptr(), n_tok_dev.data_ptr(), int(group), cap,
             ctypes.addressof(ks_arr), len(ks), _COUNTER_SCRATCH[key].data_ptr(), out.data_ptr(), _stream())
    return out


_COUNTER_SCRATCH = {}


STREAM_DIMS = (16, 32, 64, 128, 256)      # feature dims of the register-stationary streaming kernels


HOIST_FALSE_NEGATIVE_TEST = os.environ.get("MHR_NCE_HOIST", "1") != "0"


class NceSaved:
    """Tensors the forward keeps for the backward (all preallocated at token capacity)."""
    __slots__ = ("qn", "pn", "supp", "q_inv", "p_inv", "s_pos", "lse", "loss", "n_valid", "rank", "negs",
                 "n_tok_dev", "tok_cap", "cap", "thres", "dim", "n_neg", "groups", "q_idx", "p_idx", "bucket_idx", "n_buckets",
                 "bucket_sum", "bucket_cnt", "u", "wide", "scale_dev", "cap_eff", "wide_pack",
                 # query-row sharing (nce_shared.hip): row-level state of the streaming kernels + the maps between rows and tokens
                 "shared", "tok2row", "row_first", "n_row_dev", "row_cap", "fix_words", "fix_slot", "fix_any", "n_p_rows", "row_q",
                 "window", "bwd_bufs",
                 # REMI's interest-aware hard-negative loss (dense path): beta and the two saved log-sums per token
                 "ihn_beta", "ihn_num", "ihn_imp")


_ROW_IOTA = {}


SHARE_ROWS = os.environ.get("MHR_NCE_SHARE_ROWS", "1") != "0"
DETERMINISTIC = False


def set_deterministic(on):
    """Order-independent reductions in the loss backward (include/mhr.h: deterministic mode): fixed-point accumulators for the
    negative-side gradient, ordered partial sums for d(logit_scale), single-range column sums, a fixed-order fold of the per-offset
    loss sums.  Bitwise reproducible steps (run to run, replayed vs host-issued) for the row-sharing sampled softmax (loss =
    'prior' with pred_len > 1: cfg1); a few per cent slower.  Process-wide; also MHR_DETERMINISTIC=1 in the environment."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    lib.call("mhr_set_deterministic", 1 if on else 0)


if os.environ.get("MHR_DETERMINISTIC", "0") == "1":
    try:
        set_deterministic(True)
    except RuntimeError:          # (library not built yet: lib.load() raises at the first real use anyway)
        DETERMINISTIC = True
_ZERO_FIX = {}


def _row_maps(q_idx, n_tok_dev, cap, row_cap):
    """Runs of equal query rows in the (offset-fastest) token lists -> (row list [G, row_cap], tok2row [G, cap],
    row_first [G, row_cap], n_row [G]); all on the device, no host sync (mhr_row_maps: count + scan, like the compaction)."""
    G = q_idx.shape[0]
    dev = q_idx.device
    r_q, r_first = zeros_many(dev, ((G, row_cap), torch.int32), ((G, row_cap), torch.int32))
    tok2row = torch.empty(G, cap, dtype=torch.int32, device=dev)
    n_row = torch.empty(G, dtype=torch.int32, device=dev)
    scratch = torch.empty(G, (cap + 4095) // 4096, dtype=torch.int32, device=dev)
    lib.call("mhr_row_maps", q_idx.data_ptr(), n_tok_dev.data_ptr(), G, cap, row_cap, r_q.data_ptr(), r_first.data_ptr(),
             tok2row.data_ptr(), n_row.data_ptr(), scratch.data_ptr(), _stream())
    return r_q, tok2row, r_first, n_row


def _fix_bits_launch(p_rows, negs, n_neg, D, G, thres, p_row_mask):
    dev = negs.device
    n_p_rows = p_rows.shape[0]
    rp_pad = (n_p_rows + 255) // 256 * 256
    n_tiles = (n_neg + 31) // 32
    fix_words = torch.empty(G, n_tiles, rp_pad, dtype=torch.int32, device=dev)
    fix_any, slot_of_row = zeros_many(dev, ((G, rp_pad), torch.int32), ((G, n_p_rows), torch.int32))
    row_list = n_list = None
    if p_row_mask is None:
        slot_of_row = None
    else:
        assert p_row_mask.shape == (G, n_p_rows)
        key = (G, n_p_rows, str(dev))
        if key not in _ROW_IOTA:
            ar = torch.arange(n_p_rows, dtype=torch.int32, device=dev)
            _ROW_IOTA[key] = (ar[None].expand(G, -1).contiguous(), ar)
        iota_g, iota = _ROW_IOTA[key]
        row_list, _, _, n_list = token_compact(p_row_mask.contiguous(), iota_g, iota, iota, tok_cap=rp_pad)
    lib.call("mhr_nce_fix_bits", p_rows.data_ptr(), _dt(p_rows), n_p_rows, negs.data_ptr(), n_neg, D, G, float(thres),
             fix_words.data_ptr(), _ptr(row_list), _ptr(n_list), _ptr(slot_of_row), fix_any.data_ptr(), _stream())
    return fix_words, fix_any, slot_of_row, (row_list, n_list)


def _fix_bits_tables(p_rows, negs, n_neg, D, G, thres, p_row_mask):
    tabs = _fix_bits_launch(p_rows, negs, n_neg, D, G, thres, p_row_mask)
    return tabs[0], tabs[1], tabs[2]


def nce_shared_prepare_stages(q_idx, p_idx, n_tok_dev, p_rows, negs, thres, p_row_mask, want_logs, n_q_rows=None):
    """Everything the row-sharing forward needs that depends on the BATCH only (token lists, target rows, negatives) and not
    on the query rows: the row maps of the token lists, the false-negative bit table of the target rows, the first target of
    every row, the normalised target rows and the zeroed accumulators (with n_q_rows, the number of query rows, the backward's
    accumulators as well).  `nce_fwd(share_rows=True)` builds it itself; a model may build it EARLY - on a second stream
    underneath the sequence encoder, a stage at a time between the encoder's layers - and hand it in (`prep=`).
    Returns (prep, stages): the dict the stages fill, and the stages (callables, to be run in order on one stream); (None, [])
    when the shapes are not the row-sharing path's (ragged capacity / pool sizes, feature dims it does not take)."""
    G, cap = q_idx.shape
    n_neg, D = negs.shape[1], negs.shape[2]
    if not SHARE_ROWS or D not in STREAM_DIMS or cap % 32 or n_neg % 32:
        return None, []
    return _shared_prepare_stages(q_idx, p_idx, n_tok_dev, p_rows, negs, n_neg, thres, p_row_mask, want_logs, n_q_rows)


def _shared_prepare_stages(q_idx, p_idx, n_tok_dev, p_rows, negs, n_neg, thres, p_row_mask, want_logs, n_q_rows=None):
    """(n_neg: the pool's own size; `negs` may be padded to whole 32-row tiles behind it)"""
    G, cap = q_idx.shape
    D = negs.shape[2]
    dev = negs.device
    row_cap = cap + 32
    prep = {"key": (q_idx.data_ptr(), p_idx.data_ptr(), p_rows.data_ptr(), negs.data_ptr(), float(thres), bool(want_logs)),
            "keep": (q_idx, p_idx, p_rows, negs)}

    def row_maps():
        prep["row_maps"] = _row_maps(q_idx, n_tok_dev, cap, row_cap)

    def fix_bits():                 # the real false-negative bit table, per target row
        prep["fix"] = _fix_bits_tables(p_rows, negs, n_neg, D, G, thres, p_row_mask)

    def rows_and_zeros():
        prep["r_p"] = torch.gather(p_idx, 1, prep["row_maps"][2].long().clamp_(max=cap - 1)).contiguous()
        prep["z"] = zeros_many(dev, ((G, row_cap), torch.float32), ((G, row_cap), torch.int32), ((G, row_cap), torch.int32),
                               ((G, cap), torch.int32), ((G, cap), torch.int32))
        # the normalised target is a property of the TARGET ROW (shared by every token and group that points at it)
        pn_rows, p_norm = l2norm_rows(p_rows.contiguous(), torch.bfloat16, want_norms=True)
        prep["pn"] = (pn_rows, 1.0 / p_norm)

    def backward_buffers():         # zero / +inf fills that wait for nothing: (dq, dp, d_negs, d_scale, lw_row)
        dq, dp, dn, dls = zeros_many(dev, ((int(n_q_rows), D), torch.float32), ((p_rows.shape[0], D), torch.float32),
                                     ((G, n_neg, D), torch.float32), ((1,), torch.float32))
        prep["bwd"] = (dq, dp, dn, dls, torch.full((G, row_cap), float("inf"), dtype=torch.float32, device=dev))

    return prep, [row_maps, fix_bits, rows_and_zeros] + ([backward_buffers] if n_q_rows is not None else [])


def nce_shared_prepare(q_idx, p_idx, n_tok_dev, p_rows, negs, thres, p_row_mask, want_logs, n_q_rows=None):
    """`nce_shared_prepare_stages` run in one go on the current stream: the finished prep (or None)."""
    prep, stages = nce_shared_prepare_stages(q_idx, p_idx, n_tok_dev, p_rows, negs, thres, p_row_mask, want_logs, n_q_rows)
    for f in stages:
        f()
    return prep


def _nce_fwd_shared(sv, q_rows, p_rows, negs, logit_scale, thres, want_logs, bucket_idx, n_buckets, log_group, p_row_mask, loss,
                    window=None, prep=None):
    """Query-row sharing (csrc/nce_shared.hip): the streaming kernels see each distinct query row once."""
    dev = negs.device
    G, n_neg, D = sv.groups, sv.n_neg, sv.dim              # negs itself is padded to whole 32-row tiles
    cap, tok_cap, n_tok_dev = sv.cap, sv.tok_cap, sv.n_tok_dev
    q_idx, p_idx = sv.q_idx, sv.p_idx
    row_cap = cap + 32
    n_p_rows = p_rows.shape[0]
    rp_pad = (n_p_rows + 255) // 256 * 256
    n_tiles = (n_neg + 31) // 32
    st = _stream()
    key = (q_idx.data_ptr(), p_idx.data_ptr(), p_rows.data_ptr(), negs.data_ptr(), float(thres), bool(want_logs))
    if prep is not None and prep["key"] != key:
        raise ValueError("nce_fwd: prep was built for other token lists / target rows / negatives than this call's")
    if prep is None:
        prep, stages = _shared_prepare_stages(q_idx, p_idx, n_tok_dev, p_rows, negs, n_neg, thres, p_row_mask, want_logs)
        for f in stages:
            f()
    r_q, tok2row, r_first, n_row = prep["row_maps"]
    # (1) the real false-negative bit table, per target row
    fix_words, fix_any, slot_of_row = prep["fix"]
    # (2) the fused streaming forward over the ROWS with NOTHING suppressed (mhr_nce_fwd's plain form: no bit table, no
    #     suppression words, no normalised-target rows written); its positive is the target of the row's first token, so the
    #     log counters of offset-0 tokens come out of this launch.  Pools that are not whole 32-negative tiles take the
    #     general form with an all-zero bit table that is never written.
    plain = n_neg % 32 == 0
    r_p, z = prep["r_p"], prep["z"]
    sum_row = z[0]
    nv_row, rk_row = (z[1], z[2]) if want_logs else (None, None)
    qn_row = torch.empty(G, row_cap, D, dtype=torch.bfloat16, device=dev)
    q_inv_row = torch.empty(G, row_cap, dtype=torch.float32, device=dev)
    p_inv_row = torch.empty(G, row_cap, dtype=torch.float32, device=dev)
    s_pos_row = torch.empty(G, row_cap, dtype=torch.float32, device=dev)
    u_row = torch.empty(G, row_cap, D, dtype=torch.float32, device=dev)
    if plain:
        pn_row = supp_row = None
        fix_args = (0, 0, 0, 0)
    else:
        zkey = (G, n_tiles, n_p_rows, str(dev))
        if zkey not in _ZERO_FIX:         # grow-only: a captured hipGraph of the step keeps reading the entry of its shape
            _ZERO_FIX[zkey] = (torch.zeros(G, n_tiles, rp_pad, dtype=torch.int32, device=dev),
                               torch.zeros(G, rp_pad, dtype=torch.int32, device=dev), torch.zeros(G, dtype=torch.int32, device=dev),
                               torch.zeros(G, n_p_rows, dtype=torch.int32, device=dev))
        fix_args = tuple(t.data_ptr() for t in _ZERO_FIX[zkey])
        pn_row = torch.empty(G, row_cap, D, dtype=torch.bfloat16, device=dev)
        supp_row = torch.empty(G, n_tiles, row_cap, dtype=torch.int32, device=dev)
    _timed_call("mhr_nce_fwd", q_rows.data_ptr(), r_q.data_ptr(), p_rows.data_ptr(), r_p.data_ptr(), _dt(q_rows),
                negs.data_ptr(), n_neg, D, G, n_row.data_ptr(), row_cap, logit_scale.data_ptr(), float(thres),
                sum_row.data_ptr(), _ptr(nv_row), _ptr(rk_row), qn_row.data_ptr(), _ptr(pn_row), _ptr(supp_row),
                q_inv_row.data_ptr(), p_inv_row.data_ptr(), s_pos_row.data_ptr(), int(log_group), u_row.data_ptr(), n_p_rows,
                *fix_args, st)
    # (3) per token: s+, sums and counters with the token's own suppressed negatives taken out.  The normalised target is a
    #     property of the TARGET ROW (shared by every token and group that points at it): one l2norm pass over p_rows
    pn_rows, sv.p_inv = prep["pn"]
    sv.pn = pn_rows
    ssum = torch.empty(G, cap, dtype=torch.float32, device=dev)
    n_valid, rank = (z[3], z[4]) if want_logs else (None, None)
    _timed_call("mhr_nce_shared_fwd_tokens", pn_rows.data_ptr(), n_p_rows, p_idx.data_ptr(), tok2row.data_ptr(), G,
                n_tok_dev.data_ptr(), cap, row_cap, qn_row.data_ptr(), sum_row.data_ptr(), _ptr(nv_row), _ptr(rk_row),
                negs.data_ptr(), n_neg, D, logit_scale.data_ptr(), fix_words.data_ptr(), _ptr(slot_of_row), fix_any.data_ptr(),
                sv.s_pos.data_ptr(), ssum.data_ptr(), _ptr(n_valid), _ptr(rank), st)
    lib.call("mhr_nce_finalize", ssum.data_ptr(), sv.s_pos.data_ptr(), G, n_tok_dev.data_ptr(), cap,
             logit_scale.data_ptr(), loss.data_ptr(), sv.lse.data_ptr(), _ptr(n_valid), _ptr(bucket_idx), int(n_buckets),
             _ptr(sv.bucket_sum), _ptr(sv.bucket_cnt), st)
    sv.shared = True
    sv.bwd_bufs = prep.get("bwd")
    sv.qn, sv.u, sv.q_inv, sv.supp = qn_row, u_row, q_inv_row, supp_row
    sv.tok2row, sv.row_first, sv.n_row_dev, sv.row_cap, sv.row_q = tok2row, r_first, n_row, row_cap, r_q
    sv.window = window                      # (tok_of_slot [G, n_slots], L, P) of window-structured lists, or None
    sv.fix_words, sv.fix_slot, sv.fix_any, sv.n_p_rows = fix_words, slot_of_row, fix_any, n_p_rows
    sv.loss = loss[:, :tok_cap]
    sv.n_valid = None if n_valid is None else n_valid[:, :tok_cap]
    sv.rank = None if rank is None else rank[:, :tok_cap]
    return sv


def nce_fwd(q_rows, q_idx, p_rows, p_idx, negs, n_tok_dev, tok_cap, logit_scale, thres=0.99, want_logs=False,
            for_backward=True, bucket_idx=None, n_buckets=0, log_group=-1, p_row_mask=None, share_rows=False, window=None,
            ihn_beta=0.0, prep=None):
    """Grouped sampled softmax.  q_rows/p_rows [*, D] (bf16 or f32, same dtype, shared by all groups);
    q_idx/p_idx [G, tok_cap] int32; negs [G, n_neg, D] bf16 normalised; n_tok_dev [G] int32.
    (1-D q_idx / 2-D negs are accepted as a single group.)  Saved tensors carry the leading group axis.
    p_row_mask [G, p_rows.shape[0]] bool/uint8 (optional): per group, a superset of the rows of p_rows that live tokens
    point at - the hoisted false-negative test then visits only those rows.
    share_rows: tokens with the same query row are neighbours in the lists (several prediction offsets of one position):
    the negative-pool products run once per distinct row (csrc/nce_shared.hip).  Same results.
    window = (tok_of_slot, L, P) (with share_rows): the lists are the compaction of (b, l, p) window slots with
    p_idx = b (L + P) + l + 1 + p (token_compact(slot_map=True)): the backward then needs no per-token atomics.
    ihn_beta > 0: REMI's interest-aware hard-negative loss (remi.py:203-288) instead of the plain sampled softmax; runs on the
    dense path (library GEMM + the ihn_dense epilogues of csrc/wide.hip) at every feature dim.
    prep: `nce_shared_prepare(...)` of exactly these lists / rows / negatives, built earlier (row-sharing path only)."""
    if q_idx.dim() == 1:
        q_idx, p_idx, negs, n_tok_dev = q_idx[None], p_idx[None], negs[None], n_tok_dev.view(1)
    _chk(negs, "negs", torch.bfloat16)
    _chk(q_idx, "q_idx", torch.int32)
    _chk(p_idx, "p_idx", torch.int32)
    _chk(n_tok_dev, "n_tok_dev", torch.int32)
    _chk(logit_scale, "logit_scale", torch.float32)
    assert q_rows.dtype == p_rows.dtype
    dev = negs.device
    G, n_neg, D = negs.shape
    assert q_idx.shape == (G, tok_cap) and p_idx.shape == (G, tok_cap) and n_tok_dev.numel() == G
    if n_neg % 32:                                    # pools are stored [G, round_up(n_neg, 32), D]: whole tiles stream unclamped
        negs = torch.nn.functional.pad(negs, (0, 0, 0, 32 - n_neg % 32)).contiguous()
    # the kernels stream whole 32-token tiles: capacities are rounded up (production shapes already are multiples of 32)
    cap = (tok_cap + 31) // 32 * 32
    if cap != tok_cap:
        q_idx = torch.nn.functional.pad(q_idx, (0, cap - tok_cap)).contiguous()
        p_idx = torch.nn.functional.pad(p_idx, (0, cap - tok_cap)).contiguous()
        if bucket_idx is not None:
            bucket_idx = torch.nn.functional.pad(bucket_idx, (0, cap - tok_cap)).contiguous()
    sv = NceSaved()
    sv.shared = False
    sv.bwd_bufs = None
    sv.q_idx, sv.p_idx = q_idx, p_idx
    sv.ihn_beta = float(ihn_beta)
    sv.wide = D not in STREAM_DIMS or sv.ihn_beta > 0     # feature dims beyond the register-stationary kernels (and the IHN loss): wide.py
    sv.bucket_idx, sv.n_buckets, sv.bucket_sum, sv.bucket_cnt = bucket_idx, int(n_buckets), None, None
    shared_path = share_rows and SHARE_ROWS and for_backward and D in STREAM_DIMS and float(ihn_beta) <= 0
    zf = zeros_many(dev, ((2, G, max(n_buckets, 1)), torch.float32), ((G, cap), torch.float32), ((G, cap), torch.float32),
                    ((G, cap) if (want_logs and not shared_path) else (1,), torch.int32),
                    ((G, cap) if (want_logs and not shared_path) else (1,), torch.int32))
    if bucket_idx is not None:       # per-(group, bucket) loss sums and token counts come out of the finalize kernel
        _chk(bucket_idx, "bucket_idx", torch.int32)
        assert bucket_idx.shape == (G, cap)
        sv.bucket_sum, sv.bucket_cnt = zf[0][0], zf[0][1]
    loss = zf[1]
    sv.lse = zf[2]
    n_valid = zf[3] if want_logs else None           # (the row-sharing path brings its own per-token counters)
    rank = zf[4] if want_logs else None
    sv.s_pos = torch.empty(G, cap, dtype=torch.float32, device=dev)
    if sv.ihn_beta > 0:
        sv.ihn_num = torch.zeros(G, cap, dtype=torch.float32, device=dev)
        sv.ihn_imp = torch.zeros(G, cap, dtype=torch.float32, device=dev)
    if share_rows and SHARE_ROWS and for_backward and not sv.wide:
        sv.negs = negs
        sv.n_tok_dev, sv.tok_cap, sv.cap, sv.thres, sv.dim, sv.n_neg, sv.groups = n_tok_dev, tok_cap, cap, float(thres), D, n_neg, G
        return _nce_fwd_shared(sv, q_rows, p_rows, negs, logit_scale, thres, want_logs, bucket_idx, n_buckets, log_group,
                               p_row_mask, loss, window, prep)
    if for_backward or sv.wide:
        sv.qn = torch.empty(G, cap, D, dtype=torch.bfloat16, device=dev)
        sv.pn = torch.empty(G, cap, D, dtype=torch.bfloat16, device=dev)
        sv.q_inv = torch.empty(G, cap, dtype=torch.float32, device=dev)
        sv.p_inv = torch.empty(G, cap, dtype=torch.float32, device=dev)
        sv.supp = sv.u = None
        if not sv.wide:
            sv.supp = torch.empty(G, (n_neg + 31) // 32, cap, dtype=torch.int32, device=dev)
            sv.u = torch.empty(G, cap, D, dtype=torch.float32, device=dev)  # unnormalised token-side gradient (fused forward)
    else:
        sv.qn = sv.pn = sv.supp = sv.q_inv = sv.p_inv = sv.u = None
    sv.negs = negs
    sv.n_tok_dev, sv.tok_cap, sv.cap, sv.thres, sv.dim, sv.n_neg, sv.groups = n_tok_dev, tok_cap, cap, float(thres), D, n_neg, G
    if sv.wide:
        from . import wide
        wide.nce_fwd_wide(sv, q_rows, p_rows, negs, logit_scale, want_logs, bucket_idx, loss, n_valid, rank)
        sv.loss = loss[:, :tok_cap]
        sv.n_valid = None if n_valid is None else n_valid[:, :tok_cap]
        sv.rank = None if rank is None else rank[:, :tok_cap]
        return sv
    ssum = torch.zeros(G, cap, dtype=torch.float32, device=dev)
    st = _stream()
    # training path: the false-negative test runs once per (group, target row, negative) into a bit table (see mhr.h)
    n_p_rows = p_rows.shape[0]
    fix_words = None
    row_list = n_list = slot_of_row = None
    if sv.u is not None and HOIST_FALSE_NEGATIVE_TEST:
        rp_pad = (n_p_rows + 255) // 256 * 256
        fix_words = torch.empty(G, (n_neg + 31) // 32, rp_pad, dtype=torch.int32, device=dev)
        if p_row_mask is not None:
            assert p_row_mask.shape == (G, n_p_rows)
            key = (G, n_p_rows, str(dev))
            if key not in _ROW_IOTA:
                ar = torch.arange(n_p_rows, dtype=torch.int32, device=dev)
                _ROW_IOTA[key] = (ar[None].expand(G, -1).contiguous(), ar)
            iota_g, iota = _ROW_IOTA[key]
            row_list, _, _, n_list = token_compact(p_row_mask.contiguous(), iota_g, iota, iota, tok_cap=rp_pad)
            slot_of_row = torch.zeros(G, n_p_rows, dtype=torch.int32, device=dev)     # zero: any lookup stays in bounds
    _timed_call("mhr_nce_fwd", q_rows.data_ptr(), q_idx.data_ptr(), p_rows.data_ptr(), p_idx.data_ptr(), _dt(q_rows),
                negs.data_ptr(), n_neg, D, G, n_tok_dev.data_ptr(), cap, logit_scale.data_ptr(), float(thres),
                ssum.data_ptr(), _ptr(n_valid), _ptr(rank), _ptr(sv.qn), _ptr(sv.pn),
                _ptr(sv.supp), _ptr(sv.q_inv), _ptr(sv.p_inv), _ptr(sv.s_pos), int(log_group), _ptr(sv.u), n_p_rows,
                _ptr(fix_words), _ptr(row_list), _ptr(n_list), _ptr(slot_of_row), st)
    lib.call("mhr_nce_finalize", ssum.data_ptr(), sv.s_pos.data_ptr(), G, n_tok_dev.data_ptr(), cap,
             logit_scale.data_ptr(), loss.data_ptr(), sv.lse.data_ptr(), _ptr(n_valid), _ptr(bucket_idx), int(n_buckets),
             _ptr(sv.bucket_sum), _ptr(sv.bucket_cnt), st)
    sv.loss = loss[:, :tok_cap]
    sv.n_valid = None if n_valid is None else n_valid[:, :tok_cap]
    sv.rank = None if rank is None else rank[:, :tok_cap]
    return sv


def nce_bwd(sv, w, logit_scale, q_idx, p_idx, dq_rows, dp_rows, d_negs=None, d_logit_scale=None, want_negs=True, lw_row=None,
            exclusive_q_rows=False):
    """w [G, tok_cap] f32 = dLoss/dloss[g, t] - or [G, n_buckets] when the forward was given bucket_idx (every token
    of a bucket then has the same weight).  Accumulates into dq_rows [Rq, D] / dp_rows [Rp, D] (f32, the
    forward's shared row spaces); returns (d_negs [G, n_neg, D] f32, d_logit_scale [1]).  q_idx / p_idx are the
    forward's index lists (the padded copies saved by nce_fwd are what the kernels read).  want_negs=False skips the
    negative-side product (frozen negatives, e.g. the HLLM twin's cached item tower) and returns d_negs = None.
    exclusive_q_rows: no two (group, query-row) pairs of the forward name the same row of dq_rows (the groups read disjoint
    decoding heads): the row-wise backward then adds with plain read-modify-writes instead of float atomics."""
    dev = sv.negs.device
    D, cap, G = sv.dim, sv.cap, sv.groups
    if w.dim() == 1:
        w = w[None]
    bucketed = sv.bucket_idx is not None and w.shape == (G, sv.n_buckets) and sv.n_buckets != sv.tok_cap
    if d_negs is None and want_negs:
        d_negs = torch.zeros(G, sv.n_neg, D, dtype=torch.float32, device=dev)
    if d_logit_scale is None:
        d_logit_scale = torch.zeros(1, dtype=torch.float32, device=dev)
    if cap != sv.tok_cap and not bucketed:
        w = torch.nn.functional.pad(w, (0, cap - sv.tok_cap))
    w = w.contiguous()
    _chk(w, "w", torch.float32)
    _chk(dq_rows, "dq_rows", torch.float32)
    _chk(dp_rows, "dp_rows", torch.float32)
    if sv.wide:
        from . import wide
        w_tok = torch.gather(w, 1, sv.bucket_idx.long().clamp(0, sv.n_buckets - 1)) if bucketed else w
        wide.nce_bwd_wide(sv, w_tok, logit_scale, dq_rows, dp_rows, d_negs, d_logit_scale)
        return d_negs, d_logit_scale
    st = _stream()
    if sv.shared:
        wb_ptr, nb = (sv.bucket_idx.data_ptr(), sv.n_buckets) if bucketed else (0, 0)
        dn_ptr = d_negs.data_ptr() if want_negs else 0
        if lw_row is None:                   # (+inf: a row the kernels do not visit contributes nothing to the negative-side product)
            lw_row = torch.full((G, sv.row_cap), float("inf"), dtype=torch.float32, device=dev)
        assert lw_row.shape == (G, sv.row_cap)
        dn_fix = dls_part = None
        if DETERMINISTIC and sv.window is not None:
            dn_fix = torch.zeros(d_negs.shape, dtype=torch.int64, device=dev) if want_negs else None
            dls_part = torch.zeros(G * 1024, dtype=torch.float32, device=dev)
        if sv.window is not None:            # window-structured lists: sums formed where they land, no per-token atomics
            tos, L_, P_ = sv.window
            _timed_call("mhr_nce_shared_bwd_rows", sv.qn.data_ptr(), sv.u.data_ptr(), sv.q_inv.data_ptr(), sv.row_q.data_ptr(),
                        sv.row_first.data_ptr(), sv.n_row_dev.data_ptr(), sv.row_cap, sv.pn.data_ptr(), D, G, cap,
                        logit_scale.data_ptr(), sv.lse.data_ptr(), w.data_ptr(), sv.s_pos.data_ptr(), sv.p_idx.data_ptr(),
                        dq_rows.data_ptr(), d_logit_scale.data_ptr(), lw_row.data_ptr(), wb_ptr, nb, sv.negs.data_ptr(), sv.n_neg,
                        sv.fix_words.data_ptr(), sv.n_p_rows, _ptr(sv.fix_slot), sv.fix_any.data_ptr(), dn_ptr,
                        1 if exclusive_q_rows else 0, _ptr(dn_fix), _ptr(dls_part), st)
            if dls_part is not None:         # the workgroups' partials of d(logit_scale), folded in index order
                lib.call("mhr_det_sum_into", dls_part.data_ptr(), dls_part.numel(), logit_scale.data_ptr(), 1, d_logit_scale.data_ptr(), st)
            _timed_call("mhr_nce_shared_bwd_targets", sv.qn.data_ptr(), sv.row_cap, sv.tok2row.data_ptr(), tos.data_ptr(),
                        sv.n_tok_dev.data_ptr(), G, tos.shape[1], cap, int(L_), int(P_), sv.pn.data_ptr(), sv.p_inv.data_ptr(), D,
                        logit_scale.data_ptr(), sv.lse.data_ptr(), w.data_ptr(), sv.s_pos.data_ptr(), wb_ptr, nb, sv.n_p_rows,
                        dp_rows.data_ptr(), st)
        else:
            lw_tok = torch.empty(G, cap, dtype=torch.float32, device=dev)
            _timed_call("mhr_nce_shared_bwd_tokens", sv.qn.data_ptr(), sv.u.data_ptr(), sv.q_inv.data_ptr(), sv.row_cap,
                        sv.tok2row.data_ptr(), sv.pn.data_ptr(), D, G, sv.n_tok_dev.data_ptr(), cap, logit_scale.data_ptr(),
                        sv.lse.data_ptr(), w.data_ptr(), sv.p_inv.data_ptr(), sv.s_pos.data_ptr(), sv.q_idx.data_ptr(),
                        sv.p_idx.data_ptr(), dq_rows.data_ptr(), dp_rows.data_ptr(), d_logit_scale.data_ptr(), lw_tok.data_ptr(),
                        wb_ptr, nb, sv.negs.data_ptr(), sv.n_neg, sv.fix_words.data_ptr(), sv.n_p_rows, _ptr(sv.fix_slot),
                        sv.fix_any.data_ptr(), dn_ptr, st)
            if want_negs:
                lib.call("mhr_nce_row_lw", lw_tok.data_ptr(), sv.row_first.data_ptr(), sv.n_row_dev.data_ptr(), G, cap, sv.row_cap,
                         lw_row.data_ptr(), st)
        if want_negs:
            _timed_call("mhr_nce_bwd_negs", sv.qn.data_ptr(), sv.negs.data_ptr(), _ptr(sv.supp), sv.n_neg, D, G,
                        sv.n_row_dev.data_ptr(), sv.row_cap, logit_scale.data_ptr(), lw_row.data_ptr(), d_negs.data_ptr(),
                        _ptr(dn_fix), st)
            if dn_fix is not None:           # fixed-point accumulators (tiles + suppressed-pair corrections) -> d_negs
                lib.call("mhr_det_flush", dn_fix.data_ptr(), d_negs.data_ptr(), d_negs.numel(), st)
        return d_negs, d_logit_scale
    lw = torch.empty(G, cap, dtype=torch.float32, device=dev)     # lse log2e - log2 w: written by bwd_tokens, read by bwd_negs
    dq_fix = dp_fix = dls_part = dn_fix = None
    if DETERMINISTIC:             # order-independent accumulation (include/mhr.h: deterministic mode)
        dq_fix = torch.zeros(dq_rows.shape, dtype=torch.int64, device=dev)
        dp_fix = torch.zeros(dp_rows.shape, dtype=torch.int64, device=dev)
        dls_part = torch.zeros(G * 2048 * 4, dtype=torch.float32, device=dev)
        dn_fix = torch.zeros(d_negs.shape, dtype=torch.int64, device=dev) if want_negs else None
    _timed_call("mhr_nce_bwd_tokens", sv.qn.data_ptr(), sv.pn.data_ptr(), sv.u.data_ptr(), D,
                G, sv.n_tok_dev.data_ptr(), cap, logit_scale.data_ptr(), sv.lse.data_ptr(), w.data_ptr(), sv.q_inv.data_ptr(),
                sv.p_inv.data_ptr(), sv.s_pos.data_ptr(), sv.q_idx.data_ptr(), sv.p_idx.data_ptr(), dq_rows.data_ptr(),
                dp_rows.data_ptr(), d_logit_scale.data_ptr(), lw.data_ptr(), sv.bucket_idx.data_ptr() if bucketed else 0,
                sv.n_buckets if bucketed else 0, _ptr(dq_fix), _ptr(dp_fix), _ptr(dls_part), st)
    if dq_fix is not None:
        lib.call("mhr_det_flush", dq_fix.data_ptr(), dq_rows.data_ptr(), dq_rows.numel(), st)
        lib.call("mhr_det_flush", dp_fix.data_ptr(), dp_rows.data_ptr(), dp_rows.numel(), st)
        lib.call("mhr_det_sum_into", dls_part.data_ptr(), dls_part.numel(), logit_scale.data_ptr(), 1, d_logit_scale.data_ptr(), st)
    if want_negs:
        _timed_call("mhr_nce_bwd_negs", sv.qn.data_ptr(), sv.negs.data_ptr(), sv.supp.data_ptr(), sv.n_neg, D, G,
                    sv.n_tok_dev.data_ptr(), cap, logit_scale.data_ptr(), lw.data_ptr(), d_negs.data_ptr(), _ptr(dn_fix), st)
        if dn_fix is not None:
            lib.call("mhr_det_flush", dn_fix.data_ptr(), d_negs.data_ptr(), d_negs.numel(), st)
    return d_negs, d_logit_scale


# ------------------------------------------------------------------------------------------------
# catalog scoring / top-k / merge
# ------------------------------------------------------------------------------------------------
def catalog_emit(users, H, items, tag_bits, row_bits, tau, hist_ptr, hist_items, cap, item_begin=0, item_stride=1,
                 cand=None, n_items=None):
    n_rows, D = users.shape
    dev = users.device
    if cand is None:
        cand = (torch.empty(n_rows, cap, dtype=torch.float32, device=dev),
                torch.empty(n_rows, cap, dtype=torch.int32, device=dev),
                torch.zeros(n_rows, dtype=torch.int32, device=dev))
    else:
        cand[2].zero_()
    _timed_call("mhr_catalog_score_emit", users.data_ptr(), n_rows, H, items.data_ptr(),
                items.shape[0] if n_items is None else n_items, D, item_begin,
             item_stride, _ptr(tag_bits), row_bits.data_ptr(), tau.data_ptr(), _ptr(hist_ptr), _ptr(hist_items),
             cand[0].data_ptr(), cand[1].data_ptr(), cand[2].data_ptr(), cap, _stream())
    return cand


def topk_select(cand, cap, k):
    val, idx, cnt = cand
    n_rows = cnt.shape[0]
    dev = cnt.device
    out_val = torch.empty(n_rows, k, dtype=torch.float32, device=dev)
    out_idx = torch.empty(n_rows, k, dtype=torch.int64, device=dev)
    kth = torch.empty(n_rows, dtype=torch.float32, device=dev)
    status = torch.empty(n_rows, dtype=torch.int32, device=dev)
    _timed_call("mhr_topk_select", val.data_ptr(), idx.data_ptr(), cnt.data_ptr(), cap, n_rows, k, out_val.data_ptr(),
             out_idx.data_ptr(), kth.data_ptr(), status.data_ptr(), _stream())
    return out_val, out_idx, kth, status


def _slices_for(n_rows, n_tiles):
    """item slices of the sliced emit: about 512 workgroups (2 per CU), a multiple of 8 (one per XCD), <= n_tiles - decided by
    the library (mhr_catalog_emit_slices; its workspace query sizes the candidate lists)."""
    return int(lib.load().mhr_catalog_emit_slices(int(n_rows), int(n_tiles) * 32))


def catalog_emit_sliced(users, items, n_items, tag_bits, row_bits, tau, cap_s, item_begin=0, item_stride=1):
    """-> (cand_val, cand_idx [n_rows, n_lists, cap_s], cand_cnt [n_rows, n_lists], n_lists = 2 x item slices).
    No history filter (topk_select_sliced applies it)."""
    n_rows, D = users.shape
    dev = users.device
    n_sel = (n_items - item_begin + item_stride - 1) // item_stride
    n_slices = _slices_for(n_rows, (n_sel + 31) // 32)
    val = torch.empty(n_rows, 2 * n_slices, cap_s, dtype=torch.float32, device=dev)      # 2 lists per slice (one per wave half)
    idx = torch.empty(n_rows, 2 * n_slices, cap_s, dtype=torch.int32, device=dev)
    cnt = torch.empty(n_rows, 2 * n_slices, dtype=torch.int32, device=dev)
    _timed_call("mhr_catalog_score_emit_sliced", users.data_ptr(), n_rows, items.data_ptr(), n_items, items.shape[0], D,
                item_begin, item_stride, _ptr(tag_bits), row_bits.data_ptr(), tau.data_ptr(), val.data_ptr(), idx.data_ptr(),
                cnt.data_ptr(), n_slices, cap_s, _stream())
    return val, idx, cnt, 2 * n_slices


def pack_tiles(x, n_sel=None, row_begin=0, row_stride=1, tiles_per_block=8):
    """Rows {row_begin + j row_stride} of x [n, D] bf16 -> the wide scorer's packed tile images (uint8 tensor)."""
    _chk(x, "x", torch.bfloat16)
    n, D = x.shape
    if n_sel is None:
        n_sel = (n - row_begin + row_stride - 1) // row_stride
    nbytes = lib.load().mhr_pack_tiles_bytes(n_sel, D, tiles_per_block)
    out = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    lib.call("mhr_pack_tiles", x.data_ptr(), n, D, row_begin, row_stride, n_sel, tiles_per_block, out.data_ptr(), _stream())
    return out


def pack_tiles_t(x, n_sel=None, tiles_per_block=8):
    """Columns [0, n_sel) of x [n_k, ld] bf16 as packed rows (the contraction index runs down x's rows): the transposed operand
    of mhr_wide_gemm_nt.  -> (uint8 tensor, padded contraction length)."""
    _chk(x, "x", torch.bfloat16)
    n_k, ld = x.shape
    n_sel = ld if n_sel is None else int(n_sel)
    k_pad = -(-n_k // 64) * 64
    out = torch.empty(lib.load().mhr_pack_tiles_bytes(n_sel, k_pad, tiles_per_block), dtype=torch.uint8, device=x.device)
    lib.call("mhr_pack_tiles_t", x.data_ptr(), n_k, ld, n_sel, tiles_per_block, out.data_ptr(), _stream())
    return out, k_pad


def catalog_emit_wide(users_p, n_rows, D, items_p, n_items, tag_bits, row_bits, tau, cap_s, item_begin=0, item_stride=1):
    """Feature dims beyond 256 (a multiple of 64): the LDS-tiled MFMA scorer with the fused threshold emit
    (csrc/catalog_wide.hip) on PACKED operands (pack_tiles: users with 4 tiles per block, the selected items with 8).
    -> (cand_val, cand_idx [n_rows, n_lists, cap_s], cand_cnt [n_rows, n_lists], n_lists = 4 x slices)."""
    dev = users_p.device
    n_slices = lib.load().mhr_catalog_wide_slices(n_rows)        # fixed by the kernel's XCD layout (32 ... 256)
    val = torch.empty(n_rows, 4 * n_slices, cap_s, dtype=torch.float32, device=dev)
    idx = torch.empty(n_rows, 4 * n_slices, cap_s, dtype=torch.int32, device=dev)
    cnt = torch.empty(n_rows, 4 * n_slices, dtype=torch.int32, device=dev)
    _timed_call("mhr_catalog_score_emit_wide", users_p.data_ptr(), n_rows, items_p.data_ptr(), n_items, D, item_begin, item_stride,
                _ptr(tag_bits), row_bits.data_ptr(), tau.data_ptr(), val.data_ptr(), idx.data_ptr(), cnt.data_ptr(), n_slices,
                cap_s, _stream())
    return val, idx, cnt, 4 * n_slices


def topk_select_sliced(cand, H, hist_ptr, hist_items, k):
    """-> (values [n_rows,k], indices [n_rows,k], kth value, valid-candidate count, overflow status)."""
    val, idx, cnt, n_slices = cand
    n_rows, _, cap_s = val.shape
    dev = val.device
    out_val = torch.empty(n_rows, k, dtype=torch.float32, device=dev)
    out_idx = torch.empty(n_rows, k, dtype=torch.int64, device=dev)
    kth = torch.empty(n_rows, dtype=torch.float32, device=dev)
    count = torch.empty(n_rows, dtype=torch.int32, device=dev)
    status = torch.empty(n_rows, dtype=torch.int32, device=dev)
    _timed_call("mhr_topk_select_sliced", val.data_ptr(), idx.data_ptr(), cnt.data_ptr(), n_slices, cap_s, n_rows, H,
                _ptr(hist_ptr), _ptr(hist_items), k, out_val.data_ptr(), out_idx.data_ptr(), kth.data_ptr(), count.data_ptr(),
                status.data_ptr(), _stream())
    return out_val, out_idx, kth, count, status


def sub_history(hist_ptr, hist_items, users_f):
    """CSR history (ptr [B+1] int32, items sorted per user) restricted to the users `users_f` [F] (ascending)."""
    if hist_ptr is None:
        return None, None
    dev = hist_ptr.device
    lens = (hist_ptr[1:] - hist_ptr[:-1])[users_f].long()
    sub_ptr = torch.zeros(users_f.numel() + 1, dtype=torch.int32, device=dev)
    sub_ptr[1:] = torch.cumsum(lens, 0).int()
    starts = hist_ptr[:-1][users_f].long()
    off = torch.arange(int(lens.sum()), device=dev) - torch.repeat_interleave(sub_ptr[:-1].long(), lens)
    return sub_ptr, hist_items[torch.repeat_interleave(starts, lens) + off].contiguous()


def catalog_topk(users, H, items, tag_bits, row_bits, hist_ptr, hist_items, k, cap=4096, target=None, stats=None, n_items=None,
                 k_min=None, tau_out=None, margin=None, defer_check=None):
    """Exact per-row top-k over the whole catalog (value desc, index asc), rows = (user, head) pairs.

    k_min (default k): rows with fewer than k_min candidates are re-run exactly; with k_min < k a row may return fewer than
    k finite entries (its list then holds EVERY item scoring >= its threshold).  tau_out (dict, optional): receives 'tau'
    [rows] f32, the emit threshold each row's candidates were collected with (-inf: every admissible item was a candidate).

    users [B*H, D] bf16 normalised, items [>= N, D] bf16 normalised (rows beyond n_items = N are padding: give the
    table round_up(N, 32) rows and the item tiles stream unclamped).  Returns (values [B*H,k] f32, indices [B*H,k] i64).
    Thresholds come from two strided sample passes; the full pass emits the few scores above them into per-(row, item
    slice) lists and an exact select picks the top k.  Exactness is verified (enough candidates, no list overflow, per
    row) and rows that fail are re-run with tau = -inf, so the sampling only affects speed.
    defer_check (int32 device tensor [>= 1], optional): instead of reading the verification flag here (a host sync) and repairing,
    write "some row failed" into defer_check[0] and return; the caller reads it together with its own flags and, when set,
    calls again without defer_check (catalog_topk_exact does).
    """
    n_rows, D = users.shape
    N = items.shape[0] if n_items is None else int(n_items)
    dev = users.device
    if D not in STREAM_DIMS:
        from . import wide
        return wide.catalog_topk_wide(users, H, items, N, tag_bits, row_bits, hist_ptr, hist_items, k, target=target, stats=stats,
                                      k_min=k_min, tau_out=tau_out, margin=margin)
    ninf = torch.full((n_rows,), float("-inf"), dtype=torch.float32, device=dev)
    k_min = k if k_min is None else min(k, k_min)
    if N <= cap:
        cand = catalog_emit(users, H, items, tag_bits, row_bits, ninf, hist_ptr, hist_items, N, n_items=N)
        ov, oi, _, _ = topk_select(cand, N, k)
        if tau_out is not None:
            tau_out["tau"] = ninf
        return ov, oi
    if target is None:
        # candidates aimed at per row.  The threshold is the (target/s2)-th largest of a 1/s2 sample: its rank estimate
        # scatters by about 1/sqrt(target/s2), so 2.5 k leaves > 5 sigma before a row would come up short (and such a
        # row is only re-run, never wrong); every candidate above what is needed costs a divergent slow-path visit.
        target = max(512, int(2.5 * k_min))
    s1 = max(1, -(-N // 2048))
    s2 = max(1, min(-(-N // 32768), target // 48))       # threshold = the ~50th largest of the second sample (see wide.py)
    t1 = min(1024, max(8, -(-3 * target // s1)))           # first threshold: about 3x looser than the rank aimed at (a tighter one
                                                   # starves the second sample and, as the fallback threshold, the candidates)
    t2 = min(1024, max(k_min // s2 + 1, target // s2))
    # pass 1: every s1-th item, all scores -> the t1-th largest bounds the top ~0.4 %
    nt1 = -(-(-(-N // s1)) // 32)                                      # tiles of the first sample
    c1 = catalog_emit_sliced(users, items, N, tag_bits, row_bits, ninf, 32 * -(-nt1 // _slices_for(n_rows, nt1)), 0, s1)
    _, _, kth1, _, st1 = topk_select_sliced(c1, H, hist_ptr, hist_items, t1)
    # pass 2: every s2-th item above kth1 -> the t2-th largest estimates the score of rank ~target
    c2 = catalog_emit_sliced(users, items, N, tag_bits, row_bits, kth1, 32, 0, s2)
    _, _, kth2, _, st2 = topk_select_sliced(c2, H, hist_ptr, hist_items, t2)
    tau = torch.empty(n_rows, dtype=torch.float32, device=dev)
    lib.call("mhr_topk_pick_tau", kth1.data_ptr(), kth2.data_ptr(), st1.data_ptr(), st2.data_ptr(), n_rows, tau.data_ptr(), _stream())
    n_sl = _slices_for(n_rows, -(-N // 32))
    cap_s = max(32, 4 * -(-target // n_sl) + 16)
    cand = catalog_emit_sliced(users, items, N, tag_bits, row_bits, tau, cap_s)
    ov, oi, _, cnt, st = topk_select_sliced(cand, H, hist_ptr, hist_items, k)
    flagged = torch.empty(n_rows, dtype=torch.bool, device=dev)
    any_flag = defer_check if defer_check is not None else torch.empty(1, dtype=torch.int32, device=dev)
    lib.call("mhr_topk_flag", st.data_ptr(), cnt.data_ptr(), row_bits.data_ptr(), tau.data_ptr(), int(k_min), n_rows,
             flagged.data_ptr(), any_flag.data_ptr(), _stream())
    if stats is not None:
        stats["mean_candidates"] = float(cnt.float().mean())
        stats["flagged_rows"] = int(flagged.sum())
    if defer_check is not None:                           # the caller reads the flag (with its own) and comes back if it is set
        if tau_out is not None:
            tau_out["tau"] = tau
        return ov, oi
    if bool(any_flag.item()):                             # one host sync per batch; results go to the host anyway
        users_f = torch.nonzero(flagged.view(-1, H).any(dim=1)).flatten()
        rows_f = (users_f[:, None] * H + torch.arange(H, device=dev)[None, :]).flatten()
        sub_ptr, sub_items = sub_history(hist_ptr, hist_items, users_f)
        sub_users = users[rows_f].contiguous()
        sub_bits = row_bits[rows_f].contiguous()
        sub_tau = torch.full((rows_f.numel(),), float("-inf"), dtype=torch.float32, device=dev)
        c3 = catalog_emit(sub_users, H, items, tag_bits, sub_bits, sub_tau, sub_ptr, sub_items, N, n_items=N)
        fv, fi, _, _ = topk_select(c3, N, k)
        ov[rows_f] = fv
        oi[rows_f] = fi
        tau = tau.clone()
        tau[rows_f] = float("-inf")
    if tau_out is not None:
        tau_out["tau"] = tau
    return ov, oi


DENSE_ROWS_CHUNK = 128     # rows scored densely per launch (128 x 454 k x 8 B = 465 MB of scratch at cfg1)


def dense_rows_topk(users, H, items, n_items, tag_bits, row_bits, hist_ptr, hist_items, rows, k):
    """Exact top-k (value desc, index asc) of the rows `rows` (int32 [F], indices into users [B*H, D]) with EVERY score
    kept: `mhr_catalog_score_rows_dense` (fp32 accumulation of fp32 or bf16 operands, tag / pad / history masks in place)
    + the exact select over the whole row.  hist_ptr / hist_items: the CSR history of ALL users (row // H picks the user).
    The rare path of the decode (rows the threshold scorers cannot certify); reference hstu.py:965-1015, trainer.py:724-726,
    collector.py:245.  -> (values [F, k] f32, indices [F, k] i64)."""
    N = int(n_items)
    dev = users.device
    if users.dtype != items.dtype or users.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"dense_rows_topk: users / items must both be fp32 or both bf16 (got {users.dtype}, {items.dtype})")
    if not (users.is_cuda and users.is_contiguous() and items.is_contiguous()):
        raise ValueError("dense_rows_topk: contiguous GPU tensors only (no CPU path)")
    dt = lib.F32 if users.dtype == torch.float32 else lib.BF16
    rows = rows.to(torch.int32).contiguous()
    F_ = rows.numel()
    ov = torch.empty(F_, k, dtype=torch.float32, device=dev)
    oi = torch.empty(F_, k, dtype=torch.int64, device=dev)
    for r0 in range(0, F_, DENSE_ROWS_CHUNK):
        r1 = min(F_, r0 + DENSE_ROWS_CHUNK)
        val = torch.empty(r1 - r0, N, dtype=torch.float32, device=dev)
        idx = torch.empty(r1 - r0, N, dtype=torch.int32, device=dev)
        cnt = torch.empty(r1 - r0, dtype=torch.int32, device=dev)
        _timed_call("mhr_catalog_score_rows_dense", users.data_ptr(), items.data_ptr(), dt, users.shape[1], N, rows[r0:r1].data_ptr(),
                    r1 - r0, H, _ptr(tag_bits), row_bits.data_ptr(), _ptr(hist_ptr), _ptr(hist_items), val.data_ptr(),
                    idx.data_ptr(), cnt.data_ptr(), _stream())
        fv, fi, _, _ = topk_select((val, idx, cnt), N, k)
        ov[r0:r1], oi[r0:r1] = fv, fi
    return ov, oi


BF16_SCORE_ERR = 2.0 ** -8       # |u_bf16 . i_bf16 - u . i| for unit vectors u, i (each component rounded to 8 bits)


def catalog_topk_exact(users_f32, H, items_bf, items_f32, tag_bits, row_bits, hist_ptr, hist_items, k, n_items=None, stats=None,
                       target=None, defer=False):
    """Per-row top-k ranked on FP32 scores of fp32 operands - the reference's score path (hstu.py:965-979: fp32 normalise,
    fp32 matmul; collector.py:245 torch.topk) - without a [B, H, N] tensor.  The bf16 scorer finds every candidate whose bf16
    score lies within 2^-7 of the k-th bf16 score (a superset of the fp32 top-k: the two scores differ by at most 2^-8),
    `mhr_rescore_f32` re-scores those few hundred candidates per row from the fp32 rows, and the exact select picks k by
    (fp32 value desc, index asc).  Rows whose margin set cannot be certified (more than 1024 near-ties) fall back to dense
    fp32 scoring of that row.  users_f32 [B*H, D] fp32 normalised; items_bf [>= N, D] bf16 / items_f32 [N, D] fp32 normalised.
    target: the bf16 pass's candidate budget per row (catalog_topk; tests shrink it to force its repair path).
    defer=True: enqueue everything WITHOUT the host read and return (values, indices, finish): finish() reads the two flags, repairs
    the rare rows in place and returns the final (values, indices) - the split the replayed evaluation step needs (everything in
    front of finish() is capturable: REC/trainer/trainer.py:_EvalGraph)."""
    n_rows, D = users_f32.shape
    N = items_f32.shape[0] if n_items is None else int(n_items)
    dev = users_f32.device
    users_bf = users_f32.to(torch.bfloat16).contiguous()
    # candidates kept per row: the margin set must fit.  Wide feature dims concentrate the cosines of (near-)random embeddings
    # around 0 (std 1 / sqrt(D)), so a 2^-7 margin below the k-th score holds several hundred items there: keep the maximum
    k2 = min(N - 1, 1024, max(2 * k + 64, k + 256) if D <= 256 else 1024)
    if k2 <= k:
        k2 = k
    # the bf16 scorer's candidate lists (everything above a per-row threshold tau near the rank-2.5k score), the best k2 of them
    # sorted; a row is CERTIFIED when its margin [kth - 2^-7, ...) lies above tau (nothing that could enter the fp32 top-k was
    # left below the threshold) and does not fill all k2 slots
    # ONE host read per call: the bf16 pass leaves its "some row failed" flag on the device (defer_check) next to this function's
    # "some row is uncertified" flag; only when the first is set - thresholds that came up short, never seen on trained or random
    # embeddings - the pass is repeated with its own check-and-repair
    deferred = D in STREAM_DIMS and N > 4096 and stats is None
    flags = torch.empty(2, dtype=torch.int32, device=dev)
    kk = min(k, k2)
    _chk(users_f32, "users_f32", torch.float32)
    _chk(items_f32, "items_f32", torch.float32)

    def attempt(defer):
        tinfo = {}
        bv, bi = catalog_topk(users_bf, H, items_bf, tag_bits, row_bits, hist_ptr, hist_items, k2, stats=stats, n_items=N, k_min=k,
                              tau_out=tinfo, margin=2 * BF16_SCORE_ERR,      # (margin: used by the wide scorer's threshold, see wide.py)
                              defer_check=flags[0:1] if defer else None, target=target)
        tau = tinfo.get("tau")
        if tau is None:               # a scorer that does not report its threshold certifies nothing: +inf flags every row
            tau = torch.full((n_rows,), float("inf"), dtype=torch.float32, device=dev)   # (exact scorers report tau = -inf)
        bv = bv.contiguous()
        cnt = torch.empty(n_rows, dtype=torch.int32, device=dev)          # the margin set: a prefix of the sorted list
        lib.call("mhr_topk_margin_count", bv.data_ptr(), n_rows, k2, kk, 2 * BF16_SCORE_ERR, cnt.data_ptr(), _stream())
        rv = torch.empty(n_rows, k2, dtype=torch.float32, device=dev)
        ri = torch.empty(n_rows, k2, dtype=torch.int32, device=dev)
        lib.call("mhr_rescore_f32", users_f32.data_ptr(), items_f32.data_ptr(), D, N, bi.data_ptr(), n_rows, k2, cnt.data_ptr(),
                 rv.data_ptr(), ri.data_ptr(), _stream())
        ov, oi, _, st = topk_select((rv, ri, cnt), k2, k)
        # uncertified rows: the margin reaches below the emit threshold, or fills the candidate list (the (k2+1)-th bf16 score
        # might be inside it too)
        full = torch.empty(n_rows, dtype=torch.bool, device=dev)
        lib.call("mhr_topk_uncertified", cnt.data_ptr(), bv.data_ptr(), k2, kk, tau.contiguous().data_ptr(), 1 if k2 < N - 1 else 0,
                 2 * BF16_SCORE_ERR, n_rows, full.data_ptr(), flags[1:2].data_ptr(), _stream())
        return ov, oi, cnt, full

    if defer and not deferred:
        raise ValueError("catalog_topk_exact(defer=True) needs the streaming scorer's deferred check (feature dim <= 256, N > 4096, no stats)")
    ov, oi, cnt, full = attempt(deferred)

    def finish():
        nonlocal ov, oi, cnt, full
        got = flags.tolist()                                              # the one host sync
        if deferred and got[0]:
            ov, oi, cnt, full = attempt(False)
            got = flags.tolist()
        if stats is not None:
            stats["margin_mean"] = float(cnt.float().mean())
            stats["uncertified_rows"] = int(full.sum())
        if got[1]:                                                        # a handful of rows at most: every fp32 score kept
            rows = torch.nonzero(full).flatten().int()
            fv, fi = dense_rows_topk(users_f32, H, items_f32, N, tag_bits, row_bits, hist_ptr, hist_items, rows, k)
            ov[rows.long()], oi[rows.long()] = fv, fi
        return ov, oi

    if defer:
        return ov, oi, finish
    return finish()


def multihead_merge_dedup(vals, idx, B, H, k):
    dev = vals.device
    out_idx = torch.empty(B, k, dtype=torch.int64, device=dev)
    out_val = torch.empty(B, k, dtype=torch.float32, device=dev)
    out_src = torch.empty(B, k, dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    lib.call("mhr_multihead_merge_dedup", vals.data_ptr(), idx.data_ptr(), B, H, k, out_idx.data_ptr(), out_val.data_ptr(),
             out_src.data_ptr(), status.data_ptr(), _stream())
    return out_idx, out_val, out_src, status


def hit_matrix(topk_idx, positives, n_pos):
    B, k = topk_idx.shape
    hit = torch.empty(B, k, dtype=torch.uint8, device=topk_idx.device)
    lib.call("mhr_hit_matrix", topk_idx.data_ptr(), B, k, positives.data_ptr(), positives.stride(0), n_pos, hit.data_ptr(),
             _stream())
    return hit

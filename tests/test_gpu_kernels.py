"""Kernel-level parity on a real MI355X: every C-ABI entry point against the CPU oracle
(oracle/) evaluated on the SAME bf16-rounded operands.

Tolerances (stated per test):
  * integer / index outputs: bit-exact;
  * logits / scores / losses from bf16 operands with fp32 accumulation: 1e-4 relative (north star);
  * tensors stored in bf16 by the kernel: 2^-8 relative to the row scale (one bf16 rounding);
  * gradients that pass a bf16-rounded probability / gate tile through a second MFMA: 2e-2 of the max-abs.
"""
import math

import numpy as np
import pytest
import torch

from oracle import decode_oracle as DO
from oracle import hstu_oracle as HO
from oracle import optim_oracle as OO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mhr_amd  # noqa: F401
    from mhr_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda().contiguous()


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------
def test_gather_and_pos(ops):
    g = torch.Generator().manual_seed(0)
    N, D, B, L, P = 1000, 256, 7, 20, 3
    table = torch.randn(N, D, generator=g)
    pos = torch.randn(L + 1, D, generator=g)
    ids = torch.randint(0, N, (B, L + P), generator=g)
    rows, x = ops.embedding_gather(dev(table), dev(ids), torch.bfloat16, dev(pos), L, torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(rows.cpu(), bf(table[ids]))                               # pure data movement + one rounding
    assert torch.equal(x.cpu(), table[ids[:, :L]] + pos[:L][None])               # fp32 add: bit-exact
    rows32, _ = ops.embedding_gather(dev(table), dev(ids.flatten()), torch.float32)
    assert torch.equal(rows32.cpu(), table[ids.flatten()])
    # small dim (D=16) and ragged row count
    t2 = torch.randn(50, 16, generator=g)
    i2 = torch.randint(0, 50, (13,), generator=g)
    r2, _ = ops.embedding_gather(dev(t2), dev(i2), torch.float32)
    assert torch.equal(r2.cpu(), t2[i2])



@pytest.mark.parametrize("D,B,L,P,n_neg", [(256, 7, 20, 3, 1000), (64, 5, 50, 1, 333), (1024, 2, 6, 2, 61), (16, 3, 8, 1, 9)])
def test_gather_step_is_gather_plus_l2norm_bitwise(ops, D, B, L, P, n_neg):
    """mhr_embedding_gather_step - item windows (fp32 rows + position-added x) and negative pools (gathered AND L2-normalised
    to bf16, no fp32 copy) in ONE launch (reference hstu.py:637-643, 670-672, 752-754) - gives bit for bit what the two-step
    path gives (gather, then mhr_l2norm_rows), and the indexed backward what mhr_l2norm_rows_bwd gives on the gathered rows;
    out-of-range ids are clamped and counted on both halves."""
    g = torch.Generator().manual_seed(D + n_neg)
    N = 3000
    table = torch.randn(N, D, generator=g) * 0.02
    pos = torch.randn(L + 1, D, generator=g)
    items = torch.randint(0, N, (B, L + P), generator=g)
    negs = torch.randint(0, N, (n_neg,), generator=g)
    ids_all = torch.cat([items.flatten(), negs])
    n_item = items.numel()
    rows, x, yn, norms = ops.embedding_gather_step(dev(table), dev(pos), dev(ids_all), n_item, L, L + P)
    torch.cuda.synchronize()
    assert torch.equal(rows.cpu(), table[items.flatten()])
    assert torch.equal(x.cpu(), table[items[:, :L]] + pos[:L][None])
    rows_n, _ = ops.embedding_gather(dev(table), dev(negs), torch.float32)
    y2, n2 = ops.l2norm_rows(rows_n, torch.bfloat16, want_norms=True)
    assert torch.equal(yn, y2) and torch.equal(norms, n2)
    ref = table[negs].double()
    assert rel_err(yn.float().cpu().double(), ref / ref.norm(dim=1, keepdim=True)) < 2 ** -8
    dy = dev(torch.randn(n_neg, D, generator=g))
    dx = ops.l2norm_rows_indexed_bwd(dy, dev(table), dev(negs), norms)
    dx2 = ops.l2norm_rows_bwd(dy, rows_n, n2)
    assert torch.equal(dx, dx2)
    # ids outside the table: clamped and counted, items and negatives alike
    ops.bad_id_count()
    bad = ids_all.clone()
    bad[1], bad[n_item + 2] = N + 5, -3
    rows_b, _, yn_b, _ = ops.embedding_gather_step(dev(table), dev(pos), dev(bad), n_item, L, L + P)
    torch.cuda.synchronize()
    assert ops.bad_id_count() == 2
    assert torch.equal(rows_b.cpu()[1], table[N - 1]) and torch.equal(yn_b[n_item - n_item + 2], ops.l2norm_rows(dev(table[0:1].contiguous()), torch.bfloat16)[0])


def test_gather_counts_ids_outside_the_table(ops):
    """nn.Embedding raises on an id outside [0, N) (reference hstu.py:413, 637); the gather kernel clamps, counts, and the
    host asks where it synchronises anyway (Trainer._check_nan raises IndexError)."""
    g = torch.Generator().manual_seed(3)
    table = torch.randn(50, 64, generator=g)
    ops.bad_id_count()                                              # clear
    ids = torch.tensor([0, 49, 7, 50, -1, 3, 1000], dtype=torch.int64)
    rows, _ = ops.embedding_gather(dev(table), dev(ids), torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(rows.cpu()[[0, 1, 2, 5]], table[[0, 49, 7, 3]])
    assert ops.bad_id_count() == 3 and ops.bad_id_count() == 0      # counted once each, then reset

def test_scatter_add_dense(ops):
    g = torch.Generator().manual_seed(1)
    N, D, R = 300, 64, 2000
    ids = torch.randint(0, N, (R,), generator=g)
    ids[:500] = 7                                         # a hot row
    gr = torch.randn(R, D, generator=g)
    gt = torch.zeros(N, D)
    out = ops.embedding_scatter_add(dev(bf(gr)), dev(ids), dev(gt))
    ref = OO.segment_sum_rows(ids, bf(gr), N)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-3)     # float atomics: order varies


def test_segment_sum_and_adam_rows(ops):
    g = torch.Generator().manual_seed(2)
    N, D, B, L, P, nneg = 400, 256, 6, 10, 2, 50
    W = L + P
    ids_a = torch.randint(1, N, (B * W,), generator=g)
    ids_a[:50] = 5                                        # a hot row spanning several 32-row chunks
    ids_b = torch.randint(1, N, (nneg,), generator=g)
    ids_b[:40] = 9                                        # a second one, fed from the other gradient buffer
    ga = bf(torch.randn(B * W, D, generator=g))
    gb = torch.randn(nneg, D, generator=g)
    xg = torch.randn(B, L, D, generator=g)
    ids = torch.cat([ids_a, ids_b])
    sorted_ids, perm = torch.sort(ids, stable=True)
    out_rows = torch.zeros(ids.numel(), D).cuda()
    row_slot = torch.full((N,), -1, dtype=torch.int32).cuda()
    ops.sparse_rows_segment_sum(dev(sorted_ids), dev(perm), dev(ga), dev(gb), dev(xg), L, W, out_rows, row_slot)
    torch.cuda.synchronize()
    # dense reference gradient
    full = ga.float().clone().view(B, W, D)
    full[:, :L] += xg
    dense = OO.segment_sum_rows(ids_a, full.view(-1, D), N) + OO.segment_sum_rows(ids_b, gb, N)
    slot = row_slot.cpu()
    touched = torch.unique(ids)
    assert set(torch.nonzero(slot >= 0).flatten().tolist()) == set(touched.tolist())
    got = torch.zeros(N, D)
    got[touched] = out_rows.cpu()[slot[touched].long()]
    np.testing.assert_allclose(got.numpy(), dense.numpy(), rtol=1e-5, atol=2e-5)
    # no atomics anywhere: EVERY row - the chunk-spanning hot ones included - is bitwise reproducible, and the sum of a run
    # has one fixed order (list order inside a chunk, chunks in order); non-head rows are zero on return
    out2 = torch.zeros_like(out_rows)
    slot2 = torch.full((N,), -1, dtype=torch.int32).cuda()
    ops.sparse_rows_segment_sum(dev(sorted_ids), dev(perm), dev(ga), dev(gb), dev(xg), L, W, out2, slot2)
    assert torch.equal(out2.cpu(), out_rows.cpu()) and torch.equal(slot2.cpu(), slot)
    head = torch.ones_like(sorted_ids, dtype=torch.bool)
    head[1:] = sorted_ids[1:] != sorted_ids[:-1]
    assert float(out_rows.cpu()[~head].abs().max()) == 0.0
    src = torch.cat([full.view(-1, D), gb])                      # the fixed order, restated: chunk partials, then chunk order
    for hot in (5, 9):
        pos = torch.nonzero(sorted_ids == hot).flatten()
        acc = None
        for c0 in range(int(pos[0]) // 32 * 32, int(pos[-1]) + 1, 32):
            part = torch.zeros(D)
            for q in pos[(pos >= c0) & (pos < c0 + 32)].tolist():
                r = int(perm[q])
                part = part + (ga[r].float() if r < B * W else gb[r - B * W])
                if r < B * W and r % W < L:
                    part = part + xg[r // W, r % W]
            acc = part if acc is None else acc + part
        assert torch.equal(out_rows.cpu()[int(slot[hot])], acc), f"hot row {hot}: order of the sum is not the documented one"
    # AdamW over the table, two steps, untouched rows move too (dense semantics)
    w = torch.randn(N, D, generator=g)
    m = torch.zeros(N, D)
    v = torch.zeros(N, D)
    wd, md, vd = dev(w.clone()), dev(m.clone()), dev(v.clone())
    ops.adam_rows(wd, md, vd, out_rows, row_slot, 1, 1e-2, 0.5, weight_decay=0.01)
    OO.adamw_step(w, dense * 0.5, m, v, 1, 1e-2, weight_decay=0.01)
    torch.cuda.synchronize()
    assert int((row_slot.cpu() >= 0).sum()) == 0                                  # slots self-cleaned
    ops.adam_rows(wd, md, vd, out_rows, row_slot, 2, 1e-2, 0.5, weight_decay=0.01)     # all-zero gradient step
    OO.adamw_step(w, torch.zeros_like(w), m, v, 2, 1e-2, weight_decay=0.01)
    torch.cuda.synchronize()
    np.testing.assert_allclose(wd.cpu().numpy(), w.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(vd.cpu().numpy(), v.numpy(), rtol=2e-5, atol=1e-9)


def test_adam_flat(ops):
    g = torch.Generator().manual_seed(3)
    n = 10007
    w, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    wd, md, vd = dev(w.clone()), dev(m.clone()), dev(v.clone())
    for step in (1, 2, 3):
        ops.adam_flat(wd, dev(gr), md, vd, step, 1e-3)
        OO.adamw_step(w, gr, m, v, step, 1e-3)
    torch.cuda.synchronize()
    np.testing.assert_allclose(wd.cpu().numpy(), w.numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("D", [16, 64, 256, 1024])
def test_layernorm_fwd_bwd(ops, D):
    g = torch.Generator().manual_seed(4)
    rows = 37
    x = torch.randn(rows, D, generator=g) * 3 + 1
    dy = torch.randn(rows, D, generator=g)
    y, mean, rstd = ops.layernorm_fwd(dev(x), torch.float32)
    ref = HO.layer_norm(x)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    yb, _, _ = ops.layernorm_fwd(dev(x), torch.bfloat16)
    assert rel_err(yb.float().cpu(), ref) < 2 ** -7
    xr = x.clone().requires_grad_(True)
    HO.layer_norm(xr).backward(dy)
    base = torch.randn(rows, D, generator=g)
    dx = ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dx=dev(base.clone()), accumulate=True)
    np.testing.assert_allclose(dx.cpu().numpy(), (base + xr.grad).numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,p", [(16, 0.0), (256, 0.0), (256, 0.25)])
def test_ln_gate(ops, D, p):
    g = torch.Generator().manual_seed(5)
    rows = 29
    h = bf(torch.randn(rows, 4 * D, generator=g))
    a = bf(torch.randn(rows, D, generator=g) * 0.1)
    d_o = bf(torch.randn(rows, D, generator=g))
    o, mean, rstd = ops.ln_gate_fwd(dev(h), dev(a), D, torch.float32, dropout_p=p, seed=123)
    u = h[:, :D].float().clone().requires_grad_(True)
    ar = a.float().clone().requires_grad_(True)
    ref = HO.silu(u) * HO.layer_norm(ar)
    oc = o.cpu()
    if p > 0:
        keep = oc != 0
        frac = float(keep.float().mean())
        assert abs(frac - (1 - p)) < 0.05
        ref = torch.where(keep, ref / (1 - p), torch.zeros_like(ref))
    np.testing.assert_allclose(oc.numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    ref.backward(d_o.float())
    dh = torch.zeros(rows, 4 * D, dtype=torch.bfloat16).cuda()
    da = ops.ln_gate_bwd(dev(d_o), dev(h), dev(a), mean, rstd, dh, D, dropout_p=p, seed=123)
    assert rel_err(dh[:, :D].float().cpu(), u.grad) < 2 ** -7          # stored in bf16
    assert rel_err(da.float().cpu(), ar.grad) < 2 ** -7
    assert float(dh[:, D:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,p", [(29, 0.0), (29, 0.25), (1024, 0.25), (3, 0.5)])
def test_ln_gate_bf16_rows_of_256(ops, rows, p):
    """dim 256 with bf16 in and out takes the half-wave-per-row kernels (four rows per wave in flight): same values as the
    one-wave-per-row form (fp32 output rounded) up to the order of the row sums, the SAME dropout mask, ragged row counts."""
    D = 256
    g = torch.Generator().manual_seed(rows)
    h = bf(torch.randn(rows, 4 * D, generator=g))
    a = bf(torch.randn(rows, D, generator=g) * 0.1)
    d_o = bf(torch.randn(rows, D, generator=g))
    o32, mean32, rstd32 = ops.ln_gate_fwd(dev(h), dev(a), D, torch.float32, dropout_p=p, seed=77)
    o16, mean, rstd = ops.ln_gate_fwd(dev(h), dev(a), D, torch.bfloat16, dropout_p=p, seed=77)
    assert o16.dtype == torch.bfloat16
    assert torch.equal(o16 == 0, o32 == 0)                                   # the same mask
    assert rel_err(o16.float().cpu(), o32.cpu()) < 2 ** -8
    np.testing.assert_allclose(mean.cpu().numpy(), mean32.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), rstd32.cpu().numpy(), rtol=1e-5)
    u = h[:, :D].float().clone().requires_grad_(True)
    ar = a.float().clone().requires_grad_(True)
    ref = HO.silu(u) * HO.layer_norm(ar)
    if p > 0:
        ref = torch.where((o32 != 0).cpu(), ref / (1 - p), torch.zeros_like(ref))
    ref.backward(d_o.float())
    dh = torch.zeros(rows, 4 * D, dtype=torch.bfloat16).cuda()
    da = ops.ln_gate_bwd(dev(d_o), dev(h), dev(a), mean, rstd, dh, D, dropout_p=p, seed=77)
    assert rel_err(dh[:, :D].float().cpu(), u.grad) < 2 ** -7
    assert rel_err(da.float().cpu(), ar.grad) < 2 ** -7
    assert float(dh[:, D:].abs().max()) == 0.0


def test_add_cast(ops):
    """x + y (fp32 + bf16) and the bf16 copy of the sum in one pass: exactly torch's fp32 add and round-to-nearest-even cast."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1031, 256, generator=g) * 3
    y = bf(torch.randn(1031, 256, generator=g))
    out, out16 = ops.add_cast(dev(x), dev(y))
    want = x + y.float()
    assert torch.equal(out.cpu(), want) and torch.equal(out16.cpu(), want.to(torch.bfloat16))


def test_l2norm(ops):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(33, 256, generator=g) * 5
    y, n = ops.l2norm_rows(dev(x), torch.float32, want_norms=True)
    np.testing.assert_allclose(y.cpu().numpy(), HO.l2n(x).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(n.cpu().numpy(), x.norm(dim=-1).numpy(), rtol=1e-5)
    yb = ops.l2norm_rows(dev(x), torch.bfloat16)
    assert torch.equal(yb.cpu(), bf(y.cpu())) or rel_err(yb.float().cpu(), HO.l2n(x)) < 2 ** -8


# ------------------------------------------------------------------------------------------------
def _attn_case(B, L, Hh, hd, seed):
    g = torch.Generator().manual_seed(seed)
    D = Hh * hd
    h = bf(torch.randn(B * L, 4 * D, generator=g))
    valid = torch.rand(B, L, generator=g) > 0.2
    valid[0, : L // 3] = False
    d_out = bf(torch.randn(B * L, D, generator=g) * 0.5)
    return h, valid, d_out


from kernel_oracles import attn_oracle as _attn_oracle      # the pinned HO.hstu_attention on bf16-rounded operands


@pytest.mark.parametrize("B,L,Hh,hd", [(3, 12, 2, 8), (2, 40, 4, 16), (2, 33, 2, 32), (2, 200, 8, 32), (1, 70, 2, 64),
                                       (1, 50, 1, 128), (1, 300, 2, 64), (1, 512, 2, 64), (2, 260, 3, 32), (1, 257, 1, 16),
                                       (1, 290, 1, 24)])   # 512 x 64 = cfg2's shape
def test_hstu_attention_fwd_bwd(ops, B, L, Hh, hd):
    h, valid, d_out = _attn_case(B, L, Hh, hd, 7 + L)
    D = Hh * hd
    kv = dev(valid.to(torch.uint8))
    out, act = ops.hstu_attn_fwd(dev(h), kv, B, L, Hh, hd)
    ref, gref = _attn_oracle(h, valid, B, L, Hh, hd, d_out)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((out.float().cpu() - ref).abs().max()) < 2 ** -7 * scale          # output stored in bf16
    act_ref = bf(HO.silu(h.float()))
    assert torch.equal(act.cpu()[:, :D], act_ref[:, 2 * D:3 * D])                    # q
    assert torch.equal(act.cpu()[:, D:2 * D], act_ref[:, 3 * D:])                    # k
    assert torch.equal(act.cpu()[:, 2 * D:], act_ref[:, D:2 * D])                    # v
    dh = torch.zeros(B * L, 4 * D, dtype=torch.bfloat16).cuda()
    ops.hstu_attn_bwd(dev(h), act, kv, dev(d_out), dh, B, L, Hh, hd)
    torch.cuda.synchronize()
    got = dh.float().cpu()
    assert float(got[:, :D].abs().max()) == 0.0                                      # u block untouched
    for name, sl in (("dv", slice(D, 2 * D)), ("dq", slice(2 * D, 3 * D)), ("dk", slice(3 * D, 4 * D))):
        gs = float(gref[:, sl].abs().max())
        err = float((got[:, sl] - gref[:, sl]).abs().max())
        assert err < 2e-2 * gs, (name, err, gs)
    # without saved activations the backward recomputes silu(q), silu(k), silu(v) itself: bit-identical gradients
    dh2 = torch.zeros(B * L, 4 * D, dtype=torch.bfloat16).cuda()
    ops.hstu_attn_bwd(dev(h), None, kv, dev(d_out), dh2, B, L, Hh, hd)
    torch.cuda.synchronize()
    assert torch.equal(dh2, dh)


@pytest.mark.parametrize("B,L,Hh,hd", [(5, 200, 8, 32), (4, 70, 2, 64), (3, 33, 2, 16), (6, 300, 2, 64), (2, 512, 2, 64)])
def test_hstu_attention_sequence_layout_is_bitwise_neutral(ops, B, L, Hh, hd):
    """Front-padded batches (the loaders' layout, trainset.py:111-137): with the per-batch layout (first live block per sequence,
    longest-first order) the kernels skip the dead blocks and reorder the launch - outputs and gradients keep every bit, the
    dead rows are written as zeros.  Sequences: full, empty, one valid key, a block boundary, a hole after the first key."""
    D = Hh * hd
    g = torch.Generator().manual_seed(91 + L)
    h = dev(bf(torch.randn(B * L, 4 * D, generator=g)))
    d_out = dev(bf(torch.randn(B * L, D, generator=g) * 0.5))
    lens = [L, 0, 1, min(L, 32), L // 2, L - 1][:B]
    valid = torch.zeros(B, L, dtype=torch.bool)
    for b, n in enumerate(lens):
        valid[b, L - n:] = True
    if B > 4:
        valid[4, L - lens[4] + 1: L - lens[4] + 40] = False                          # a hole right after the first valid key
    kv = dev(valid.to(torch.uint8))
    first, order, first_row = ops.attn_seq_layout(kv, B, L)
    assert first_row.cpu().tolist() == [L - n for n in lens]
    nb = (L + 31) // 32
    want_first = [((L - n) // 32 if n else nb) for n in lens]
    assert first.cpu().tolist() == want_first
    assert sorted(order.cpu().tolist()) == list(range(B))
    fo = [want_first[i] for i in order.cpu().tolist()]
    assert fo == sorted(fo)                                                          # most live blocks first
    res = []
    for layout in (None, (first, None), (first, order)):
        out, _ = ops.hstu_attn_fwd(h, kv, B, L, Hh, hd, save_act=False, layout=layout)
        dh = torch.full((B * L, 4 * D), 7.0, dtype=torch.bfloat16, device="cuda")    # (the kernels must WRITE the dead rows)
        ops.hstu_attn_bwd(h, None, kv, d_out, dh, B, L, Hh, hd, layout=layout)
        res.append((out, dh))
    torch.cuda.synchronize()
    for out, dh in res[1:]:
        assert torch.equal(out, res[0][0])
        assert torch.equal(dh, res[0][1])
    o3 = res[2][0].view(B, L, D)
    for b, n in enumerate(lens):
        assert float(o3[b, : L - n].float().abs().max() if n < L else 0.0) == 0.0


@pytest.mark.parametrize("M,N,K,bias,lda_pad,kn", [(25600, 1024, 256, False, 0, False), (800, 256, 256, True, 0, False),
                                                   (77, 256, 256, True, 0, False), (1000, 72, 64, True, 0, False),
                                                   (64, 256, 128, False, 0, False), (333, 1024, 256, False, 256, False),
                                                   (32, 8, 64, True, 0, False), (4097, 520, 256, True, 0, False),
                                                   (25600, 1024, 256, False, 0, True), (999, 256, 256, True, 0, True),
                                                   (640, 256, 64, False, 0, True), (96, 512, 128, True, 128, True)])
def test_rows_gemm_against_the_fp32_product(ops, M, N, K, bias, lda_pad, kn):
    """ops.rows_gemm (hstu.py:236-239 / 281-288 under bf16 autocast: bf16 operands, fp32 accumulation, bf16 result) against
    the fp32 product of the same bf16 operands: within one bf16 rounding of it (the summation order differs from the
    reference's library GEMM, as two library GEMMs differ from each other).  Ragged M / N, strided rows (a column block of a
    wider buffer), all three supported K, both weight layouts ([N, K] and [K, N])."""
    g = torch.Generator().manual_seed(5 + M + N)
    a_full = bf(torch.randn(M, K + lda_pad, generator=g) * 0.5)
    a = a_full[:, :K]
    w = bf(torch.randn(N, K, generator=g) * K ** -0.5)
    b = bf(torch.randn(N, generator=g) * 0.1) if bias else None
    out = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device="cuda")             # one guard row behind the result
    w_dev = dev(w.t().contiguous()) if kn else dev(w)
    ops.rows_gemm(dev(a_full)[:, :K], w_dev, dev(b) if bias else None, out=out[:M], w_is_kn=kn)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + (b.float() if bias else 0.0)
    got = out[:M].float().cpu()
    tol = 2 ** -8 * ref.abs().clamp_min(1e-3) + 1e-6                                     # half an ulp of bf16 + accumulation slack
    assert bool(((got - ref).abs() <= 2 * tol).all()), float(((got - ref).abs() / tol).max())
    assert float(out[M].float().abs().max()) == 7.0 and float(out[M].float().abs().min()) == 7.0


@pytest.mark.parametrize("B,L,D", [(5, 200, 256), (4, 50, 64), (3, 33, 512)])
def test_row_wise_encoder_kernels_do_not_load_dead_rows(ops, B, L, D):
    """`dead=(first_row, L)` on add_layernorm_fwd/bwd and ln_gate_fwd/bwd: rows in front of a sequence's first valid key (except
    its last row) are not loaded - they read as zeros.  So the result must be, bit for bit, the plain kernel's result on inputs
    whose dead rows were zeroed beforehand; live rows keep every bit of the plain result on the original inputs."""
    g = torch.Generator().manual_seed(17 + L)
    lens = [L, 0, 1, L // 2, L - 1][:B]
    first_row = torch.tensor([L - n for n in lens], dtype=torch.int32, device="cuda")
    dead_mask = torch.zeros(B, L, dtype=torch.bool)
    for b, n in enumerate(lens):
        dead_mask[b, : min(L - n, L - 1)] = True                                   # (the last row always counts as live)
    dm = dead_mask.view(-1).cuda()
    rows = B * L
    dead = (first_row, L)

    def z(t):                                                                      # the same tensor with its dead rows zeroed
        t = t.clone()
        t[dm] = 0
        return t

    x = dev(torch.randn(rows, D, generator=g))
    y = dev(bf(torch.randn(rows, D, generator=g)))
    got = ops.add_layernorm_fwd(x, y, 1e-6, dead=dead)
    want = ops.add_layernorm_fwd(z(x), z(y), 1e-6)
    plain = ops.add_layernorm_fwd(x, y, 1e-6)
    for a_, b_, c_ in zip(got, want, plain):
        assert torch.equal(a_, b_)
        assert torch.equal(a_[~dm], c_[~dm])
    assert float(got[0][dm].abs().max() if bool(dm.any()) else 0.0) == 0.0        # zeros written

    x_out, _, mean, rstd = plain
    d_xn = dev(bf(torch.randn(rows, D, generator=g)))
    d_xout = dev(torch.randn(rows, D, generator=g))
    got = ops.add_layernorm_bwd(d_xn, x_out, mean, rstd, d_xout, dead=dead)
    want = ops.add_layernorm_bwd(z(d_xn), z(x_out), z(mean), z(rstd), z(d_xout))
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)

    h = dev(bf(torch.randn(rows, 4 * D, generator=g)))
    a = dev(bf(torch.randn(rows, D, generator=g)))
    for p_drop in (0.0, 0.2):
        if D == 256:
            # the half-wave form of the gate kernels (dim 256, bf16) ignores the descriptor: one-shot, latency-bound waves gain
            # nothing from loads they skip.  In the encoder its dead rows see the attention's zeros / zero gradients anyway.
            o, m2, r2 = ops.ln_gate_fwd(h, a, D, torch.bfloat16, 1e-6, p_drop, 11, dead=dead)
            op, mp, rp = ops.ln_gate_fwd(h, a, D, torch.bfloat16, 1e-6, p_drop, 11)
            assert torch.equal(o, op) and torch.equal(m2, mp) and torch.equal(r2, rp)
            oz, _, _ = ops.ln_gate_fwd(h, z(a), D, torch.bfloat16, 1e-6, p_drop, 11)
            assert float(oz[dm].float().abs().max() if bool(dm.any()) else 0.0) == 0.0    # zero attention rows -> zero gate rows
            continue
        o, m2, r2 = ops.ln_gate_fwd(h, a, D, torch.bfloat16, 1e-6, p_drop, 11, dead=dead)
        ow, mw, rw = ops.ln_gate_fwd(z(h), z(a), D, torch.bfloat16, 1e-6, p_drop, 11)
        op, mp, rp = ops.ln_gate_fwd(h, a, D, torch.bfloat16, 1e-6, p_drop, 11)
        assert torch.equal(o, ow) and torch.equal(m2, mw) and torch.equal(r2, rw)
        assert torch.equal(o[~dm], op[~dm]) and torch.equal(m2[~dm], mp[~dm])
        d_o = dev(bf(torch.randn(rows, D, generator=g)))
        dh1, dh2 = torch.zeros_like(h), torch.zeros_like(h)
        da1 = ops.ln_gate_bwd(d_o, h, a, mp, rp, dh1, D, p_drop, 11, dead=dead)
        da2 = ops.ln_gate_bwd(z(d_o), z(h), z(a), z(mp), z(rp), dh2, D, p_drop, 11)
        assert torch.equal(da1, da2) and torch.equal(dh1, dh2)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,L,Hh,hd", [(5, 200, 8, 32), (4, 70, 2, 64), (3, 33, 2, 16)])
def test_hstu_attention_packed_sequences(ops, B, L, Hh, hd):
    """`cu_rows`: the valid positions of the windows back to back in one row buffer (no padding rows).  The packed launch
    must give, for every valid position, what the window launch gives (same products, another 32-row blocking: bf16 rounding
    apart), and write zeros into the rows behind the last sequence."""
    D = Hh * hd
    g = torch.Generator().manual_seed(7 + L)
    lens = [L, 0, 1, L // 2, L - 1][:B]
    valid = torch.zeros(B, L, dtype=torch.bool)
    for b, n in enumerate(lens):
        valid[b, L - n:] = True
    h = bf(torch.randn(B * L, 4 * D, generator=g))
    d_out = bf(torch.randn(B * L, D, generator=g) * 0.5)
    kv = dev(valid.to(torch.uint8))
    out_w, _ = ops.hstu_attn_fwd(dev(h), kv, B, L, Hh, hd, save_act=False)
    dh_w = torch.zeros(B * L, 4 * D, dtype=torch.bfloat16, device="cuda")
    ops.hstu_attn_bwd(dev(h), None, kv, dev(d_out), dh_w, B, L, Hh, hd)
    # packed operands
    sel = valid.view(-1)
    n_live = int(sel.sum())
    cap = n_live + 37                                                                 # some rows behind the last sequence
    hp = torch.zeros(cap, 4 * D, dtype=torch.bfloat16)
    hp[:n_live] = h[sel]
    dop = torch.zeros(cap, D, dtype=torch.bfloat16)
    dop[:n_live] = d_out[sel]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    ones = torch.ones(cap, dtype=torch.uint8, device="cuda")
    order = torch.tensor(sorted(range(B), key=lambda b: -lens[b]), dtype=torch.int32, device="cuda")
    for lay in ((None, None, None, cu), (None, order, None, cu)):
        hp_d = dev(hp)
        out_p, _ = ops.hstu_attn_fwd(hp_d, ones, B, L, Hh, hd, save_act=False, layout=lay)
        dh_p = torch.full((cap, 4 * D), 5.0, dtype=torch.bfloat16, device="cuda")
        dh_p[:, :D] = 0
        ops.hstu_attn_bwd(hp_d, None, ones, dev(dop), dh_p, B, L, Hh, hd, layout=lay)
        torch.cuda.synchronize()
        assert out_p.shape[0] == cap
        ref, got = out_w.float().cpu()[sel], out_p.float().cpu()[:n_live]
        assert float((got - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max()) + 1e-6
        assert float(out_p[n_live:].float().abs().max()) == 0.0                          # tail rows: zeros written
        gr, gg = dh_w.float().cpu()[sel][:, D:], dh_p.float().cpu()[:n_live, D:]
        assert float((gg - gr).abs().max()) <= 2e-2 * float(gr.abs().max()) + 1e-6
        assert float(dh_p[n_live:, D:].float().abs().max()) == 0.0


def test_seq_pack_maps_and_masked_gather(ops):
    """mhr_seq_pack_maps numbers the valid positions sequence by sequence in window order; mhr_rows_gather_masked moves rows
    both ways.  Masks with holes, an empty and a full sequence; a capacity that is too small drops the rows past it and says so."""
    B, L, D = 6, 45, 64
    g = torch.Generator().manual_seed(3)
    valid = torch.rand(B, L, generator=g) > 0.4
    valid[1] = False
    valid[2] = True
    kv = dev(valid.to(torch.uint8))
    n_valid = int(valid.sum())
    cap = n_valid + 9
    cu, src_of, row_of, overflow = ops.seq_pack_maps(kv, B, L, cap)
    lens = valid.sum(1).tolist()
    assert cu.cpu().tolist() == [0] + list(np.cumsum(lens)) and int(overflow) == 0
    want_src = torch.nonzero(valid.view(-1)).flatten().tolist()
    assert src_of.cpu().tolist() == want_src + [-1] * 9
    ro = row_of.cpu()
    assert ro[~valid.view(-1)].eq(-1).all() and ro[valid.view(-1)].tolist() == list(range(n_valid))
    x = dev(torch.randn(B * L, D, generator=g))
    xp = ops.rows_gather_masked(x, src_of)
    assert torch.equal(xp[:n_valid], x[valid.view(-1).cuda()]) and float(xp[n_valid:].abs().max()) == 0.0
    back = ops.rows_gather_masked(xp, row_of)
    assert torch.equal(back[valid.view(-1).cuda()], x[valid.view(-1).cuda()]) and float(back[~valid.view(-1).cuda()].abs().max()) == 0.0
    xb = dev(bf(torch.randn(B * L, D, generator=g)))
    assert torch.equal(ops.rows_gather_masked(xb, src_of)[:n_valid], xb[valid.view(-1).cuda()])
    small = n_valid - 7
    cu2, src2, row2, ov2 = ops.seq_pack_maps(kv, B, L, small)
    assert int(ov2) == n_valid and int(cu2[-1]) == small
    assert int((row2 >= 0).sum()) == small and int(row2.max()) == small - 1


def test_hstu_attention_golden(ops):
    """The reference's own attention outputs (tests/golden/attention_unit.npz), bf16 tolerance."""
    from conftest import load_golden
    gold = load_golden("attention_unit")
    for tag in "abc":
        q, k, v = (torch.from_numpy(gold[f"{tag}/{n}"]) for n in "qkv")
        seq = torch.from_numpy(gold[f"{tag}/seq"])
        Hh = int(gold[f"{tag}/n_heads"])
        B, L, D = q.shape
        h = torch.zeros(B * L, 4 * D)
        h[:, D:2 * D], h[:, 2 * D:3 * D], h[:, 3 * D:] = v.view(-1, D), q.view(-1, D), k.view(-1, D)
        out, _ = ops.hstu_attn_fwd(dev(bf(h)), dev((seq != 0).to(torch.uint8)), B, L, Hh, D // Hh, apply_silu=False,
                                   save_act=False)
        ref = torch.from_numpy(gold[f"{tag}/out"]).view(B * L, D)
        assert float((out.float().cpu() - ref).abs().max()) < 3e-2 * float(ref.abs().max())   # bf16 q,k,v inputs


# ------------------------------------------------------------------------------------------------
from kernel_oracles import nce_oracle as _nce_oracle        # the pinned HO.nce_logits / HO.token_ce on bf16-rounded operands


@pytest.mark.parametrize("D,n_tok,n_neg,dtype", [(16, 37, 30, torch.float32), (64, 200, 96, torch.bfloat16),
                                                 (256, 300, 512, torch.bfloat16), (256, 129, 8192, torch.bfloat16),
                                                 (32, 90, 128, torch.float32), (128, 140, 320, torch.bfloat16),   # every tile-step shape
                                                 (512, 150, 96, torch.float32),      # beyond 256: the LDS-tiled MFMA path (nce_wide.hip)
                                                 (320, 200, 300, torch.bfloat16), (1024, 70, 61, torch.bfloat16)])   # ragged: 5 chunks / 61 negatives
def test_nce_fwd_bwd(ops, D, n_tok, n_neg, dtype):
    g = torch.Generator().manual_seed(8 + D)
    n_src = 2 * n_tok
    q_rows = (torch.randn(n_src, D, generator=g) * 2).to(dtype)
    p_rows = torch.randn(n_src, D, generator=g).to(dtype)
    q_idx = torch.randint(0, n_src, (n_tok,), generator=g).int()
    p_idx = torch.randint(0, n_src, (n_tok,), generator=g).int()
    negs = bf(HO.l2n(torch.randn(n_neg, D, generator=g)))
    # plant false negatives: some negatives equal the token's positive
    for t in range(0, n_tok, 5):
        negs[(t * 7) % n_neg] = bf(HO.l2n(p_rows[p_idx[t]].float()[None]))[0]
    ls = torch.tensor(math.log(20.0))
    cap = n_tok + 11
    qi = torch.zeros(cap, dtype=torch.int32)
    pi = torch.zeros(cap, dtype=torch.int32)
    qi[:n_tok], pi[:n_tok] = q_idx, p_idx
    ntd = torch.tensor([n_tok], dtype=torch.int32).cuda()
    lsd = ls.reshape(1).cuda()
    sv = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, lsd, 0.99, want_logs=True)
    torch.cuda.synchronize()
    q = q_rows.float()[q_idx.long()].clone().requires_grad_(True)
    p = p_rows.float()[p_idx.long()].clone().requires_grad_(True)
    nn_ = negs.float().clone().requires_grad_(True)
    lsr = ls.clone().requires_grad_(True)
    loss, logits, keep, neg, pos = _nce_oracle(q, p, nn_, lsr, 0.99)
    # 1e-4 relative on logits-derived quantities (north star tolerance)
    np.testing.assert_allclose(sv.loss.cpu().numpy()[0, :n_tok], loss.detach().numpy(), rtol=1e-4, atol=1e-4)
    assert float(sv.loss[0, n_tok:].abs().max()) == 0.0
    assert keep.sum() < keep.numel()                                               # suppression exercised
    np.testing.assert_array_equal(sv.n_valid.cpu().numpy()[0, :n_tok], (keep.sum(-1) + 1).numpy())
    rank_ref = (keep & (neg > pos)).sum(-1)
    assert int((sv.rank.cpu()[0, :n_tok] - rank_ref).abs().max()) <= 1            # ties at fp32 rounding level
    w = torch.rand(cap, generator=g)
    w[n_tok:] = 0
    (loss * w[:n_tok]).sum().backward()
    dq_rows = torch.zeros(n_src, D).cuda()
    dp_rows = torch.zeros(n_src, D).cuda()
    dn, dls = ops.nce_bwd(sv, dev(w), lsd, dev(qi), dev(pi), dq_rows, dp_rows)
    torch.cuda.synchronize()
    # several tokens may share a source row: compare in the row space
    dq_ref = torch.zeros(n_src, D).index_add_(0, q_idx.long(), q.grad)
    dp_ref = torch.zeros(n_src, D).index_add_(0, p_idx.long(), p.grad)
    for name, got, ref in (("dq", dq_rows.cpu(), dq_ref), ("dp", dp_rows.cpu(), dp_ref), ("dneg", dn.cpu()[0], nn_.grad)):
        gs = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err < 2e-2 * gs, (name, err, gs)
    assert abs(float(dls.cpu()) - float(lsr.grad)) < 2e-2 * abs(float(lsr.grad)) + 1e-4
    if D > 256 and dtype == torch.bfloat16:
        # the backward's two plain products on the library's own LDS-tiled core (mhr_wide_gemm_nt + mhr_pack_tiles_t, MHR_NCE_WIDE_GEMM=own)
        # instead of library GEMMs: same operands, fp32 accumulation - the gradients agree to accumulation order
        from mhr_amd import wide as wide_mod
        dq2, dp2 = torch.zeros(n_src, D).cuda(), torch.zeros(n_src, D).cuda()
        try:
            wide_mod.OWN_GEMM = True
            dn2, dls2 = ops.nce_bwd(sv, dev(w), lsd, dev(qi), dev(pi), dq2, dp2)
        finally:
            wide_mod.OWN_GEMM = False
        torch.cuda.synchronize()
        for a_, b_ in ((dq2, dq_rows), (dp2, dp_rows), (dn2, dn), (dls2, dls)):
            assert float((a_ - b_).abs().max()) <= 1e-4 * float(b_.abs().max()) + 1e-9


@pytest.mark.parametrize("D,n_pos,P,n_neg,dtype", [(64, 40, 4, 96, torch.float32), (256, 70, 8, 512, torch.bfloat16),
                                                   (256, 33, 8, 8192, torch.float32), (32, 50, 3, 70, torch.bfloat16),
                                                   (16, 30, 2, 64, torch.float32), (32, 45, 3, 96, torch.bfloat16),      # plain forms (whole
                                                   (128, 60, 4, 256, torch.bfloat16), (64, 50, 5, 160, torch.float32)])  #  32-negative tiles)
def test_nce_shared_query_rows_match_per_token_oracle(ops, D, n_pos, P, n_neg, dtype):
    """Query-row sharing (nce_shared.hip): runs of tokens with the same query row (the offsets of one position) through
    the row-level streaming kernels + per-token corrections vs the per-token oracle: loss 1e-4, log counters, gradients.
    Planted false negatives hit single tokens of a run, first and later ones, so every correction path is exercised."""
    g = torch.Generator().manual_seed(80 + D + P)
    n_src = n_pos + 5
    q_rows = (torch.randn(n_src, D, generator=g) * 2).to(dtype)
    p_rows = torch.randn(n_pos * P + 7, D, generator=g).to(dtype)
    q_l, p_l, o_l = [], [], []
    for pos in range(n_pos):                        # each position keeps a random subset of its P offsets (category membership)
        for off in range(P):
            if torch.rand(1, generator=g).item() < 0.6:
                q_l.append(pos + 2); p_l.append(pos * P + off); o_l.append(off)
    n_tok = len(q_l)
    q_idx, p_idx, o_idx = torch.tensor(q_l).int(), torch.tensor(p_l).int(), torch.tensor(o_l).int()
    negs = bf(HO.l2n(torch.randn(n_neg, D, generator=g)))
    for t in range(0, n_tok, 3):                    # false negatives: some negatives equal a token's positive
        negs[(t * 11) % n_neg] = bf(HO.l2n(p_rows[p_idx[t]].float()[None]))[0]
    ls = torch.tensor(math.log(20.0))
    cap = (n_tok + 11 + 31) // 32 * 32
    qi, pi, oi = (torch.zeros(cap, dtype=torch.int32) for _ in range(3))
    qi[:n_tok], pi[:n_tok], oi[:n_tok] = q_idx, p_idx, o_idx
    qi[n_tok:] = 12345                               # garbage beyond n_tok must not matter
    ntd = torch.tensor([n_tok], dtype=torch.int32).cuda()
    lsd = ls.reshape(1).cuda()
    sv = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, lsd, 0.99, want_logs=True, share_rows=True)
    assert sv.shared and int(sv.n_row_dev[0]) == len(set(q_l)) < n_tok
    torch.cuda.synchronize()
    q = q_rows.float()[q_idx.long()].clone().requires_grad_(True)
    p = p_rows.float()[p_idx.long()].clone().requires_grad_(True)
    nn_ = negs.float().clone().requires_grad_(True)
    lsr = ls.clone().requires_grad_(True)
    loss, logits, keep, neg, pos = _nce_oracle(q, p, nn_, lsr, 0.99)
    np.testing.assert_allclose(sv.loss.cpu().numpy()[0, :n_tok], loss.detach().numpy(), rtol=1e-4, atol=1e-4)
    assert float(sv.loss[0, n_tok:].abs().max()) == 0.0
    assert keep.sum() < keep.numel()
    first = torch.ones(n_tok, dtype=torch.bool)
    first[1:] = q_idx[1:] != q_idx[:-1]              # log counters are defined for the first token of a row (offset 0 in the model)
    np.testing.assert_array_equal(sv.n_valid.cpu().numpy()[0, :n_tok][first.numpy()], (keep.sum(-1) + 1).numpy()[first.numpy()])
    rank_ref = (keep & (neg > pos)).sum(-1)
    assert int((sv.rank.cpu()[0, :n_tok] - rank_ref)[first].abs().max()) <= 1
    w = torch.rand(cap, generator=g)
    w[n_tok:] = 0
    w[5] = 0                                         # a zero-weight token inside a run
    (loss * w[:n_tok]).sum().backward()
    dq_rows = torch.zeros(n_src, D).cuda()
    dp_rows = torch.zeros(p_rows.shape[0], D).cuda()
    dn, dls = ops.nce_bwd(sv, dev(w), lsd, dev(qi), dev(pi), dq_rows, dp_rows)
    torch.cuda.synchronize()
    dq_ref = torch.zeros(n_src, D).index_add_(0, q_idx.long(), q.grad)
    dp_ref = torch.zeros(p_rows.shape[0], D).index_add_(0, p_idx.long(), p.grad)
    for name, got, ref in (("dq", dq_rows.cpu(), dq_ref), ("dp", dp_rows.cpu(), dp_ref), ("dneg", dn.cpu()[0], nn_.grad)):
        gs = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err < 2e-2 * gs, (name, err, gs)
    assert abs(float(dls.cpu()) - float(lsr.grad)) < 2e-2 * abs(float(lsr.grad)) + 1e-4
    # and the same answers as the per-token kernels
    sv2 = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, lsd, 0.99, want_logs=True)
    assert not sv2.shared
    np.testing.assert_allclose(sv.loss.cpu().numpy(), sv2.loss.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("D,B,L,P,G,n_neg", [(64, 3, 9, 4, 2, 96), (256, 2, 13, 8, 3, 256)])
def test_nce_window_backward_without_token_atomics(ops, D, B, L, P, G, n_neg):
    """Window-structured lists ((b, l, p) slots, target row b (L + P) + l + 1 + p): the row-wise / target-wise backward
    (mhr_nce_shared_bwd_rows + _targets, driven by token_compact's inverse slot map) against the per-token oracle, with
    per-(group, offset) bucket weights, planted false negatives and two groups pointing at the same target rows."""
    g = torch.Generator().manual_seed(5 + D + P)
    H = G
    head_rows = (torch.randn(B * H * L, D, generator=g) * 2)
    e_rows = torch.randn(B * (L + P), D, generator=g)
    valid = torch.rand(G, B, L, P, generator=g) < 0.5
    b_ = torch.arange(B)[None, :, None, None]
    l_ = torch.arange(L)[None, None, :, None]
    p_ = torch.arange(P)[None, None, None, :]
    h_ = torch.arange(G)[:, None, None, None]
    q_all = ((b_ * H + h_) * L + l_).expand(G, B, L, P).reshape(G, -1).int().contiguous()
    p_all = (b_ * (L + P) + l_ + 1 + p_).expand(1, B, L, P).reshape(-1).int().contiguous()
    o_all = p_.expand(1, B, L, P).reshape(-1).int().contiguous()
    n_slots = B * L * P
    q_idx, p_idx, o_idx, n_tok, tos = ops.token_compact(dev(valid.reshape(G, n_slots)), dev(q_all), dev(p_all), dev(o_all), slot_map=True)
    cap = q_idx.shape[1]
    for gi in range(G):                                   # the inverse map is consistent with the lists
        live = valid[gi].reshape(-1)
        t = tos[gi].cpu()
        assert bool((t[~live] == -1).all()) and t[live].tolist() == list(range(int(live.sum())))
    negs = bf(HO.l2n(torch.randn(G, n_neg, D, generator=g)))
    for gi in range(G):
        for k in range(0, int(n_tok[gi]), 4):
            negs[gi, (k * 5) % n_neg] = bf(HO.l2n(e_rows[p_idx[gi, k].cpu().long()][None]))[0]
    ls = torch.tensor(math.log(15.0))
    lsd = ls.reshape(1).cuda()
    sv = ops.nce_fwd(dev(head_rows), q_idx, dev(e_rows), p_idx, dev(negs), n_tok, cap, lsd, 0.99, bucket_idx=o_idx, n_buckets=P,
                     share_rows=True, window=(tos, L, P))
    assert sv.shared and sv.window is not None
    w_gp = torch.rand(G, P, generator=g)
    w_gp[0, 1] = 0.0
    dq = torch.zeros(B * H * L, D).cuda()
    dp = torch.zeros(B * (L + P), D).cuda()
    dn, dls = ops.nce_bwd(sv, dev(w_gp), lsd, q_idx, p_idx, dq, dp)
    torch.cuda.synchronize()
    # oracle: every token on its own
    hq = head_rows.clone().requires_grad_(True)
    ep = e_rows.clone().requires_grad_(True)
    nn_ = negs.float().clone().requires_grad_(True)
    lsr = ls.clone().requires_grad_(True)
    total = 0.0
    for gi in range(G):
        n = int(n_tok[gi])
        qi, pi, oi = q_idx[gi, :n].cpu().long(), p_idx[gi, :n].cpu().long(), o_idx[gi, :n].cpu().long()
        loss, *_ = _nce_oracle(hq[qi], ep[pi], nn_[gi], lsr, 0.99)
        np.testing.assert_allclose(sv.loss.cpu().numpy()[gi, :n], loss.detach().numpy(), rtol=1e-4, atol=1e-4)
        total = total + (loss * w_gp[gi][oi]).sum()
    total.backward()
    for name, got, ref in (("dq", dq.cpu(), hq.grad), ("dp", dp.cpu(), ep.grad), ("dneg", dn.cpu(), nn_.grad)):
        gs = float(ref.abs().max())
        assert float((got - ref).abs().max()) < 2e-2 * gs, (name, float((got - ref).abs().max()), gs)
    assert abs(float(dls.cpu()) - float(lsr.grad)) < 2e-2 * abs(float(lsr.grad)) + 1e-4
    # same numbers from the generic (per-token atomics) form of the shared path
    sv2 = ops.nce_fwd(dev(head_rows), q_idx, dev(e_rows), p_idx, dev(negs), n_tok, cap, lsd, 0.99, bucket_idx=o_idx, n_buckets=P,
                      share_rows=True)
    dq2, dp2 = torch.zeros_like(dq), torch.zeros_like(dp)
    dn2, dls2 = ops.nce_bwd(sv2, dev(w_gp), lsd, q_idx, p_idx, dq2, dp2)
    for a, b2, tol in ((dq, dq2, 1e-4), (dp, dp2, 1e-4), (dn, dn2, 1e-2)):      # dn: the row weights feed a bf16 tile (2^-9 steps)
        assert float((a - b2).abs().max()) <= tol * float(b2.abs().max()) + 1e-7
    # the target-row gather has no atomics: bitwise reproducible
    dq3, dp3 = torch.zeros_like(dq), torch.zeros_like(dp)
    ops.nce_bwd(sv, dev(w_gp), lsd, q_idx, p_idx, dq3, dp3)
    assert torch.equal(dp3, dp)


@pytest.mark.parametrize("rows,cols", [(16, 1024 * 256), (25600, 256), (3, 64), (1, 8), (4097, 1032)])
def test_sum_rows_into(ops, rows, cols):
    """Split-K partial / bias-gradient reduction into an fp32 accumulator: exact fp32 sums of the bf16 inputs up to the
    summation order (1e-5 relative to the column scale), on top of what the accumulator already holds."""
    g = torch.Generator().manual_seed(rows + cols)
    x = bf(torch.randn(rows, cols, generator=g))
    base = torch.randn(cols, generator=g)
    out = dev(base.clone())
    ops.sum_rows_into(dev(x), out)
    torch.cuda.synchronize()
    ref = base.double() + x.double().sum(0)
    scale = float(x.float().abs().sum(0).max()) + 1.0
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("G,P,S", [(4, 8, 1), (5, 8, 2), (1, 1, 1), (33, 12, 4)])
def test_loss_reduce(ops, G, P, S):
    """Mean per (group, offset), horizon / head weighting, total and the logged partial sums in one launch (reference
    hstu.py:697-723, 836-870) against the same arithmetic in float64; the backward's per-bucket token weight likewise."""
    g = torch.Generator().manual_seed(G * 100 + P)
    bsum = torch.rand(G, P, generator=g) * 50
    bcnt = torch.randint(0, 40, (G, P), generator=g).float()
    w = torch.rand(G, P, generator=g)
    total, red = ops.loss_reduce(dev(bsum), dev(bcnt), dev(w), S)
    d_total = torch.tensor(0.37)
    wt = ops.loss_reduce_bwd(dev(d_total), dev(bcnt), dev(w))
    torch.cuda.synchronize()
    per = bsum.double() / bcnt.double().clamp_min(1.0) * w.double()
    seg = per.view(G, S, -1).sum(2)
    want = torch.cat([per.reshape(-1), seg.reshape(-1), seg.sum(1), seg.sum(0)])
    assert float((red.cpu().double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert abs(float(total) - float(per.sum())) <= 1e-5 * float(per.sum())
    want_w = 0.37 * w.double() / bcnt.double().clamp_min(1.0)
    assert float((wt.cpu().double() - want_w).abs().max()) <= 1e-6 * float(want_w.abs().max())


@pytest.mark.parametrize("n,rows,cols", [(8, 25600, 256), (3, 5, 64), (2, 4097, 1032), (16, 1, 8)])
def test_sum_rows_many(ops, n, rows, cols):
    """The one-launch form for n equally shaped matrices (the encoder's bias gradients): each destination gets exactly what
    `sum_rows_into` gives it, on top of what it already held; the matrices do not mix."""
    g = torch.Generator().manual_seed(n * rows + cols)
    xs = [dev(bf(torch.randn(rows, cols, generator=g) * (i + 1))) for i in range(n)]
    bases = [torch.randn(cols, generator=g) for _ in range(n)]
    outs = [dev(b.clone()) for b in bases]
    table = torch.tensor([x.data_ptr() for x in xs] + [o.data_ptr() for o in outs], dtype=torch.int64, device="cuda")
    ops.sum_rows_many(table, n, rows, cols, keep=(xs, outs))
    torch.cuda.synchronize()
    for x, b, o in zip(xs, bases, outs):
        ref = b.double() + x.cpu().double().sum(0)
        scale = float(x.float().abs().sum(0).max()) + 1.0
        assert float((o.cpu().double() - ref).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("rows,D", [(7, 64), (300, 256), (33, 1024)])
def test_add_layernorm_fwd_bwd(ops, rows, D):
    """Fused residual add + LayerNorm vs torch fp32 on the same bf16-rounded branch: x_out exact (fp32 add), xn and
    the bf16 gradient within one bf16 rounding of the row scale, the fp32 gradient 1e-5 relative."""
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g)
    y = bf(torch.randn(rows, D, generator=g))
    x_out, xn, mean, rstd = ops.add_layernorm_fwd(dev(x), dev(y), 1e-6)
    xr = (x + y.float()).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), eps=1e-6)
    np.testing.assert_array_equal(x_out.cpu().numpy(), xr.detach().numpy())
    assert float((xn.float().cpu() - ref.detach()).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    d_xn = bf(torch.randn(rows, D, generator=g))
    d_xout = torch.randn(rows, D, generator=g)
    dx, dy = ops.add_layernorm_bwd(dev(d_xn), x_out, mean, rstd, dev(d_xout))
    ref.backward(d_xn.float())
    want = d_xout + xr.grad
    assert rel_err(dx.cpu(), want) < 1e-5
    assert float((dy.float().cpu() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())


@pytest.mark.parametrize("G,n_slots,density", [(1, 37, 0.5), (3, 4096, 0.3), (4, 9000, 0.33), (2, 5000, 0.0), (2, 300, 1.0)])
def test_token_compact_matches_boolean_mask_indexing(ops, G, n_slots, density):
    """mhr_token_compact vs the reference's `x[mask]` (hstu.py:688-690): same values, same (ascending) order; bit-exact."""
    g = torch.Generator().manual_seed(n_slots + G)
    mask = torch.rand(G, n_slots, generator=g) < density
    q_all = torch.randint(0, 1 << 20, (G, n_slots), generator=g).int()
    p_all = torch.randint(0, 1 << 20, (n_slots,), generator=g).int()
    o_all = torch.randint(0, 8, (n_slots,), generator=g).int()
    q_idx, p_idx, o_idx, n_tok = ops.token_compact(dev(mask), dev(q_all), dev(p_all), dev(o_all))
    assert q_idx.shape[1] % 32 == 0 and q_idx.shape[1] >= n_slots
    for gi in range(G):
        n = int(mask[gi].sum())
        assert int(n_tok[gi]) == n
        np.testing.assert_array_equal(q_idx[gi, :n].cpu().numpy(), q_all[gi][mask[gi]].numpy())
        np.testing.assert_array_equal(p_idx[gi, :n].cpu().numpy(), p_all[mask[gi]].numpy())
        np.testing.assert_array_equal(o_idx[gi, :n].cpu().numpy(), o_all[mask[gi]].numpy())


def test_nce_bucket_means_equal_per_token_path(ops):
    """Per-(group, offset) mean losses from the finalize kernel, and the backward driven by per-bucket weights, equal
    the per-token path with the same weights expanded by hand (losses 1e-5 relative; gradients: same kernels, 1e-5)."""
    g = torch.Generator().manual_seed(7)
    D, n_src, n_neg, cap, G, P = 64, 200, 96, 160, 2, 5
    q_rows, p_rows = torch.randn(n_src, D, generator=g), torch.randn(n_src, D, generator=g)
    n_toks = [150, 33]
    qi = torch.randint(0, n_src, (G, cap), generator=g).int()
    pi = torch.randint(0, n_src, (G, cap), generator=g).int()
    oi = torch.randint(0, P, (G, cap), generator=g).int()
    negs = bf(HO.l2n(torch.randn(G, n_neg, D, generator=g)))
    ls = torch.tensor([math.log(20.0)]).cuda()
    ntd = torch.tensor(n_toks, dtype=torch.int32).cuda()
    sv = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, ls, 0.99, bucket_idx=dev(oi), n_buckets=P)
    loss = sv.loss.cpu()
    w_gp = torch.rand(G, P, generator=g)
    w_tok = torch.zeros(G, cap)
    for gi in range(G):
        n = n_toks[gi]
        for o in range(P):
            sel = oi[gi, :n] == o
            assert float(sv.bucket_cnt[gi, o]) == float(sel.sum())
            np.testing.assert_allclose(float(sv.bucket_sum[gi, o]), float(loss[gi, :n][sel].sum()), rtol=1e-5, atol=1e-5)
        w_tok[gi, :n] = w_gp[gi][oi[gi, :n].long()]
    dq1, dp1 = torch.zeros(n_src, D).cuda(), torch.zeros(n_src, D).cuda()
    dn1, dls1 = ops.nce_bwd(sv, dev(w_gp), ls, dev(qi), dev(pi), dq1, dp1)
    sv2 = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, ls, 0.99)
    dq2, dp2 = torch.zeros(n_src, D).cuda(), torch.zeros(n_src, D).cuda()
    dn2, dls2 = ops.nce_bwd(sv2, dev(w_tok), ls, dev(qi), dev(pi), dq2, dp2)
    for a, b in ((dq1, dq2), (dp1, dp2), (dn1, dn2), (dls1, dls2)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7


def test_nce_grouped_launch_equals_per_group(ops):
    """Three groups (different token lists, negative pools and live counts, one of them EMPTY) in one launch."""
    g = torch.Generator().manual_seed(99)
    D, n_src, n_neg, cap, G = 64, 300, 160, 256, 3
    q_rows = torch.randn(n_src, D, generator=g)
    p_rows = torch.randn(n_src, D, generator=g)
    n_toks = [200, 0, 77]
    qi = torch.randint(0, n_src, (G, cap), generator=g).int()
    pi = torch.randint(0, n_src, (G, cap), generator=g).int()
    negs = bf(HO.l2n(torch.randn(G, n_neg, D, generator=g)))
    ls = torch.tensor([math.log(20.0)]).cuda()
    ntd = torch.tensor(n_toks, dtype=torch.int32).cuda()
    sv = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, ls, 0.99, want_logs=True)
    w = torch.rand(G, cap, generator=g)
    dq, dp = torch.zeros(n_src, D).cuda(), torch.zeros(n_src, D).cuda()
    dn, dls = ops.nce_bwd(sv, dev(w), ls, dev(qi), dev(pi), dq, dp)
    torch.cuda.synchronize()
    dq_ref, dp_ref = torch.zeros(n_src, D), torch.zeros(n_src, D)
    dls_ref = 0.0
    for gi in range(G):
        n = n_toks[gi]
        assert float(sv.loss[gi, n:].abs().max()) == 0.0
        if n == 0:
            assert float(dn[gi].abs().max()) == 0.0
            continue
        q = q_rows[qi[gi, :n].long()].clone().requires_grad_(True)
        p = p_rows[pi[gi, :n].long()].clone().requires_grad_(True)
        nn_ = negs[gi].float().clone().requires_grad_(True)
        lsr = torch.tensor(math.log(20.0), requires_grad=True)
        loss, *_ = _nce_oracle(q, p, nn_, lsr, 0.99)
        np.testing.assert_allclose(sv.loss.cpu().numpy()[gi, :n], loss.detach().numpy(), rtol=1e-4, atol=1e-4)
        (loss * w[gi, :n]).sum().backward()
        dq_ref.index_add_(0, qi[gi, :n].long(), q.grad)
        dp_ref.index_add_(0, pi[gi, :n].long(), p.grad)
        dls_ref += float(lsr.grad)
        assert float((dn[gi].cpu() - nn_.grad).abs().max()) < 2e-2 * float(nn_.grad.abs().max())
    assert float((dq.cpu() - dq_ref).abs().max()) < 2e-2 * float(dq_ref.abs().max())
    assert float((dp.cpu() - dp_ref).abs().max()) < 2e-2 * float(dp_ref.abs().max())
    assert abs(float(dls.cpu()) - dls_ref) < 2e-2 * abs(dls_ref) + 1e-4
    # the hoisted false-negative test restricted to the target rows each group can reference (a superset: the referenced
    # rows plus a few others; the empty group references none): identical losses, counters and saved suppression words
    mask = torch.zeros(G, n_src, dtype=torch.bool)
    for gi in range(G):
        mask[gi, pi[gi, :n_toks[gi]].long()] = True
        mask[gi, torch.randint(0, n_src, (7,), generator=g)] = True
    mask[1] = False
    sv2 = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, ls, 0.99, want_logs=True, p_row_mask=dev(mask))
    torch.cuda.synchronize()
    assert torch.equal(sv2.loss, sv.loss) and torch.equal(sv2.n_valid, sv.n_valid) and torch.equal(sv2.rank, sv.rank)
    for gi in (0, 2):
        assert torch.equal(sv2.supp[gi, :, :n_toks[gi]], sv.supp[gi, :, :n_toks[gi]])


# ------------------------------------------------------------------------------------------------
def _catalog_case(B, H, C, N, D, seed, with_hist=True, disabled_row=None):
    g = torch.Generator().manual_seed(seed)
    users = bf(HO.l2n(torch.randn(B * H, D, generator=g)))
    items = bf(HO.l2n(torch.randn(N, D, generator=g)))
    tags = torch.rand(N, C, generator=g) < 0.4
    tags[torch.arange(N), torch.randint(0, C, (N,), generator=g)] = True
    tag_bits = (tags.long() * (1 << torch.arange(C))).sum(1) | (1 << 31)
    row_bits = torch.tensor([1 << (r % C) for r in range(B * H)], dtype=torch.int64)
    if disabled_row is not None:
        row_bits[disabled_row] = 0
    hist = [torch.unique(torch.randint(1, N, (int(torch.randint(3, 40, (1,), generator=g)),), generator=g)) for _ in range(B)]
    hist_ptr = torch.zeros(B + 1, dtype=torch.int32)
    hist_ptr[1:] = torch.cumsum(torch.tensor([len(x) for x in hist]), 0).int()
    hist_items = torch.cat(hist)
    # oracle scores on the same bf16 operands
    scores = (users.float() @ items.float().T).view(B, H, N).numpy().copy()
    for r in range(B * H):
        b, hh = divmod(r, H)
        ok = ((tag_bits & int(row_bits[r])) != 0).numpy()
        scores[b, hh, ~ok] = -np.inf
    hu = np.concatenate([np.full(len(x), b) for b, x in enumerate(hist)])
    DO.suppress(scores, hu if with_hist else None, hist_items.numpy() if with_hist else None)
    def to_i32(t):  # uint32 bit pattern stored in an int32 tensor
        t = t & 0xFFFFFFFF
        return torch.where(t >= (1 << 31), t - (1 << 32), t).int()
    return users, items, to_i32(tag_bits), to_i32(row_bits), hist_ptr, hist_items, scores


def _check_topk(ov, oi, scores, k):
    B, H, N = scores.shape
    ref_v, ref_i = DO.per_head_topk(scores, k)
    ov, oi = ov.cpu().numpy().reshape(B, H, k), oi.cpu().numpy().reshape(B, H, k)
    fin = np.isfinite(ref_v)
    # the oracle gaps must make the comparison well posed (no near-ties at fp32 accumulation-order level)
    np.testing.assert_allclose(ov[fin], ref_v[fin], rtol=1e-4, atol=1e-6)           # 1e-4 rel on bf16-operand logits
    mism = (oi != ref_i) & fin
    if mism.any():                                                                   # only allowed at numerical ties
        bad = np.argwhere(mism)
        for b, h, j in bad:
            assert abs(scores[b, h, oi[b, h, j]] - ref_v[b, h, j]) < 2e-6, (b, h, j)
    assert np.array_equal(np.isfinite(ov), fin)
    assert np.array_equal(oi[~fin], ref_i[~fin])                                     # -inf fill: ascending free ids
    return ov, oi


@pytest.mark.parametrize("B,H,C,N,D,k", [(3, 4, 4, 300, 16, 20), (5, 4, 4, 3000, 64, 200), (2, 6, 3, 700, 256, 50),
                                         (3, 4, 4, 900, 512, 40)])                   # 512: the generic-width path
def test_catalog_topk_small(ops, B, H, C, N, D, k):
    users, items, tag_bits, row_bits, hp, hi, scores = _catalog_case(B, H, C, N, D, 40 + N, disabled_row=1)
    ov, oi = ops.catalog_topk(dev(users), H, dev(items), dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k, cap=4096)
    torch.cuda.synchronize()
    _check_topk(ov, oi, scores, k)


def test_catalog_topk_sampled_thresholds(ops):
    """N large enough for the two sample passes + threshold pass; exactness must not depend on sampling."""
    B, H, C, N, D, k = 8, 4, 4, 60000, 256, 200
    users, items, tag_bits, row_bits, hp, hi, scores = _catalog_case(B, H, C, N, D, 77)
    stats = {}
    ov, oi = ops.catalog_topk(dev(users), H, dev(items), dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k, cap=4096,
                              stats=stats)
    torch.cuda.synchronize()
    _check_topk(ov, oi, scores, k)
    assert stats["mean_candidates"] < 4096
    # force the fallback: a candidate budget of barely k leaves many rows short of k candidates - they are flagged and re-run
    # exactly, and the results must not change
    stats2 = {}
    ov2, oi2 = ops.catalog_topk(dev(users), H, dev(items), dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k, cap=256,
                                target=k + 8, stats=stats2)
    assert stats2["flagged_rows"] > 0
    _check_topk(ov2, oi2, scores, k)


def test_catalog_topk_wide_threshold_emit(ops):
    """Feature dim beyond the register-stationary kernels (HLLM twin / HSTU size-4): the LDS-tiled MFMA scorer with the fused
    threshold emit (csrc/catalog_wide.hip) + the shared exact select; exactness must not depend on the sampled threshold."""
    from mhr_amd import wide
    B, H, C, N, D, k = 6, 4, 4, 40000, 512, 100
    users, items, tag_bits, row_bits, hp, hi, scores = _catalog_case(B, H, C, N, D, 91, disabled_row=2)
    stats = {}
    ov, oi = wide.catalog_topk_wide(dev(users), H, dev(items), N, dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k, chunk=8192,
                                    stats=stats)
    torch.cuda.synchronize()
    _check_topk(ov, oi, scores, k)
    assert 0 < stats["mean_candidates"] < 4096
    # a hopeless candidate budget overflows the lists: every row is re-run exactly, the result must not change
    stats2 = {}
    ov2, oi2 = wide.catalog_topk_wide(dev(users), H, dev(items), N, dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k, chunk=8192,
                                      target=20000, stats=stats2)
    assert stats2["flagged_rows"] > 0
    _check_topk(ov2, oi2, scores, k)
    # through the public entry point (dispatch on the feature dim)
    ov3, oi3 = ops.catalog_topk(dev(users), H, dev(items), dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k)
    _check_topk(ov3, oi3, scores, k)


def test_merge_dedup_and_hits_golden(ops):
    """Cross-head merge + hit matrix against the reference collector's own outputs (golden fixtures)."""
    from conftest import load_golden
    for name in ("collector_combine", "collector_smallcat", "collector_additive", "collector_k200"):
        gold = load_golden(name)
        K = int(gold["cfg/K"])
        v = torch.from_numpy(np.array(gold["out/values_by_head"]))
        i = torch.from_numpy(np.array(gold["out/idx_by_head"])).long()
        B, H, _ = v.shape
        oi, ov, osrc, st = ops.multihead_merge_dedup(dev(v), dev(i), B, H, K)
        torch.cuda.synchronize()
        fin = np.isfinite(gold["out/values"])
        assert np.array_equal(oi.cpu().numpy()[fin], gold["out/idx"][fin])                    # bit-exact indices
        assert np.array_equal(osrc.cpu().numpy()[fin], gold["out/head_source"][fin])
        assert np.array_equal(ov.cpu().numpy()[fin], gold["out/values"][fin])
        assert int(st.min()) >= K
        for b in range(B):
            assert len(set(oi[b].tolist())) == K
        rows = fin.all(1)
        pos = torch.from_numpy(np.array(gold["in/positive_i"]))
        hit_prev = None
        for p in [int(x) for x in gold["cfg/pred_len_list"]]:
            hit = ops.hit_matrix(oi, dev(pos), p + 1).cpu().numpy()
            assert np.array_equal(hit[rows], gold[f"out/topk_{p}"][:, :K][rows].astype(np.uint8))


def test_end_to_end_decode_matches_oracle(ops):
    """scores -> per-head top-k -> merge on device == oracle decode of the oracle scores (indices bit-exact)."""
    B, H, C, N, D, k = 6, 4, 4, 5000, 128, 100
    users, items, tag_bits, row_bits, hp, hi, scores = _catalog_case(B, H, C, N, D, 91)
    ov, oi = ops.catalog_topk(dev(users), H, dev(items), dev(tag_bits), dev(row_bits), dev(hp), dev(hi), k)
    mi, mv, ms, st = ops.multihead_merge_dedup(ov, oi, B, H, k)
    torch.cuda.synchronize()
    rv, ri = DO.per_head_topk(scores, k)
    items_ref, vals_ref, src_ref = DO.merge_dedup(rv, ri, k)
    assert np.array_equal(mi.cpu().numpy(), items_ref)
    assert np.array_equal(ms.cpu().numpy(), src_ref)


@pytest.mark.parametrize("B,L,H,D", [(3, 7, 4, 64), (2, 200, 4, 256), (5, 1, 16, 128)])
def test_heads_residual_fwd_bwd(ops, B, L, H, D):
    """out[b, h, l] = x[b, l] + silu(z[b, l, h]) in the [B, H, L, D] layout (llm_heads.py:5-40 + hstu.py:665-667) and its backward
    vs torch on the same bf16 z: forward to fp32 rounding, dz within one bf16 rounding, dx exact up to summation order."""
    g = torch.Generator().manual_seed(B * 100 + L)
    x = torch.randn(B * L, D, generator=g)
    z = bf(torch.randn(B * L, H * D, generator=g) * 2)
    out = ops.heads_residual_fwd(dev(x), dev(z), B, L, H)
    zr = z.float().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ref = (xr.view(B, L, 1, D) + torch.nn.functional.silu(zr.view(B, L, H, D))).permute(0, 2, 1, 3)
    assert rel_err(out.cpu(), ref.detach()) < 1e-6
    d_out = torch.randn(B, H, L, D, generator=g)
    dz, dx = ops.heads_residual_bwd(dev(d_out), dev(z), B, L, H)
    ref.backward(d_out)
    assert float((dz.float().cpu() - zr.grad).abs().max()) <= 2 ** -7 * float(zr.grad.abs().max())
    assert rel_err(dx.cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("D,n_tok,n_neg,beta", [(64, 150, 96, 1.0), (256, 70, 513, 0.5)])
def test_ihn_loss_fwd_bwd(ops, D, n_tok, n_neg, beta):
    """REMI's interest-aware hard-negative loss (remi.py:203-288) through ops.nce_fwd(ihn_beta=...) - dense logit chunks + the
    ihn_dense epilogue kernels - against the oracle restatement on the same bf16-rounded operands: loss 1e-4, counters,
    gradients of queries / targets / negatives / temperature; false negatives planted."""
    from oracle import remi_oracle as RO
    g = torch.Generator().manual_seed(31 + D)
    n_src = 2 * n_tok
    q_rows = torch.randn(n_src, D, generator=g) * 2
    p_rows = torch.randn(n_src, D, generator=g)
    q_idx = torch.randint(0, n_src, (n_tok,), generator=g).int()
    p_idx = torch.randint(0, n_src, (n_tok,), generator=g).int()
    negs = bf(HO.l2n(torch.randn(n_neg, D, generator=g)))
    for t in range(0, n_tok, 6):
        negs[(t * 7) % n_neg] = bf(HO.l2n(p_rows[p_idx[t]].float()[None]))[0]
    ls = torch.tensor(math.log(20.0))
    cap = (n_tok + 9 + 31) // 32 * 32
    qi, pi = torch.zeros(cap, dtype=torch.int32), torch.zeros(cap, dtype=torch.int32)
    qi[:n_tok], pi[:n_tok] = q_idx, p_idx
    ntd = torch.tensor([n_tok], dtype=torch.int32).cuda()
    lsd = ls.reshape(1).cuda()
    sv = ops.nce_fwd(dev(q_rows), dev(qi), dev(p_rows), dev(pi), dev(negs), ntd, cap, lsd, 0.99, want_logs=True, ihn_beta=beta)
    assert sv.wide and sv.ihn_beta == beta

    def r(t):
        return t + (bf(t).float() - t).detach()
    q = q_rows[q_idx.long()].clone().requires_grad_(True)
    p = p_rows[p_idx.long()].clone().requires_grad_(True)
    nn_ = negs.float().clone().requires_grad_(True)
    lsr = ls.clone().requires_grad_(True)
    loss, logits = RO.ihn_loss(r(HO.l2n(q)), r(HO.l2n(p)), nn_, lsr, 0.99, beta, normalized=True)   # the kernel's operand precision
    np.testing.assert_allclose(sv.loss.cpu().numpy()[0, :n_tok], loss.detach().numpy(), rtol=1e-4, atol=1e-4)
    keep = logits[:, 1:] > torch.finfo(torch.float32).min / 100
    assert keep.sum() < keep.numel()
    np.testing.assert_array_equal(sv.n_valid.cpu().numpy()[0, :n_tok], (keep.sum(-1) + 1).numpy())
    w = torch.rand(cap, generator=g)
    w[n_tok:] = 0
    (loss * w[:n_tok]).sum().backward()
    dq_rows, dp_rows = torch.zeros(n_src, D).cuda(), torch.zeros(n_src, D).cuda()
    dn, dls = ops.nce_bwd(sv, dev(w), lsd, dev(qi), dev(pi), dq_rows, dp_rows)
    torch.cuda.synchronize()
    dq_ref = torch.zeros(n_src, D).index_add_(0, q_idx.long(), q.grad)
    dp_ref = torch.zeros(n_src, D).index_add_(0, p_idx.long(), p.grad)
    for name, got, ref in (("dq", dq_rows.cpu(), dq_ref), ("dp", dp_rows.cpu(), dp_ref), ("dneg", dn.cpu()[0], nn_.grad)):
        gs = float(ref.abs().max())
        assert float((got - ref).abs().max()) < 2e-2 * gs, (name, float((got - ref).abs().max()), gs)
    assert abs(float(dls.cpu()) - float(lsr.grad)) < 2e-2 * abs(float(lsr.grad)) + 1e-4


# ------------------------------------------------------------------------------------------------
# top-k ranked on fp32 scores (ops.catalog_topk_exact): the reference's score path, hstu.py:965-979 + collector.py:245
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["predict_mult", "predict_additive", "predict_switch", "predict_switch_master"])
def test_exact_topk_matches_reference_scores(ops, name):
    """The reference's OWN fp32 scores (golden `out/scores_given*`, produced by its predict()) define the expected indices:
    torch.topk of every (user, head) row.  Fed with the reference's fp32 user heads (the pinned oracle reproduces them) and
    its fp32 item table, the fused decode must return those indices bit for bit wherever the reference's scores are untied."""
    from conftest import load_golden
    from test_oracle_golden import cfg_of, weights_of
    g = load_golden(name)
    w, ocfg = weights_of(g), cfg_of(g)
    seq = torch.from_numpy(g["in/item_seq"])
    heads = HO.user_head_embeddings(w, ocfg, seq).float()                       # [B, H, D] fp32, L2-normalised (pinned by the goldens)
    feat = torch.from_numpy(g["out/item_feature"]).float()
    ref_scores = g["out/scores_given0"] if np.isfinite(g["out/scores_given0"]).any() else g["out/scores_given1"]
    if not np.isfinite(ref_scores).any():
        pytest.skip("the fixture's heads are all switched off")
    B, H, N = ref_scores.shape
    items_f32 = HO.l2n(feat)
    assert np.abs((heads.reshape(B * H, -1) @ items_f32.T).numpy().reshape(B, H, N)[np.isfinite(ref_scores)] -
                  ref_scores[np.isfinite(ref_scores)]).max() < 2e-5            # the oracle's heads ARE the reference's
    C = ocfg["num_prior_head"]
    tags = torch.from_numpy(g["in/item_tags"])                                   # [N, C]
    tag_bits = ((tags.long() * (1 << torch.arange(C))).sum(1) | (1 << 31))
    tag_bits = torch.where(tag_bits >= (1 << 31), tag_bits - (1 << 32), tag_bits).int()
    # row bits from the reference's own -inf pattern: a head is constrained to category c iff its row masks exactly the items outside c
    row_bits = []
    for b in range(B):
        for h in range(H):
            fin = np.isfinite(ref_scores[b, h])
            bit = -(1 << 31)
            if not fin.all():
                cs = [c for c in range(C) if np.array_equal(fin, tags[:, c].numpy().astype(bool))]
                bit = (1 << cs[0]) if cs else 0
            row_bits.append(bit)
    row_bits = torch.tensor(row_bits, dtype=torch.int32)
    k = 20
    items_bf = torch.cat([bf(items_f32), torch.zeros((32 - N % 32) % 32, items_f32.shape[1], dtype=torch.bfloat16)])
    ov, oi = ops.catalog_topk_exact(dev(heads.reshape(B * H, -1)), H, dev(items_bf), dev(items_f32), dev(tag_bits), dev(row_bits),
                                    None, None, k, n_items=N)
    torch.cuda.synchronize()
    sc = ref_scores.copy()
    sc[:, :, 0] = -np.inf                                                        # pad id (trainer.py:724)
    for b in range(B):
        for h in range(H):
            if row_bits[b * H + h] == 0:
                continue
            row = sc[b, h]
            order = np.lexsort((np.arange(N), -row.astype(np.float64)))
            got = oi[b * H + h].cpu().numpy()
            for j in range(k):
                if not np.isfinite(row[order[j]]):
                    break
                untied = (j == 0 or row[order[j - 1]] - row[order[j]] > 1e-6) and (row[order[j]] - row[order[j + 1]] > 1e-6)
                if untied:
                    assert got[j] == order[j], (name, b, h, j)
                assert abs(float(ov[b * H + h, j]) - row[got[j]]) <= 2e-6        # values: the reference's fp32 scores


def test_exact_topk_large_catalog_fp32_order(ops):
    """60 000 items, D = 256: fp32 ranking vs numpy on the fp32 operands - indices exact at untied positions; and the bf16
    ranking of the same operands differs somewhere (otherwise this test would not test anything)."""
    g = torch.Generator().manual_seed(123)
    B, H, N, D, k = 8, 4, 60000, 256, 200
    users = HO.l2n(torch.randn(B * H, D, generator=g))
    items = HO.l2n(torch.randn(N, D, generator=g) + 0.3 * torch.randn(1, D, generator=g))        # a common component: crowded top
    row_bits = torch.full((B * H,), -(1 << 31), dtype=torch.int32)
    items_bf = bf(items)
    stats = {}
    ov, oi = ops.catalog_topk_exact(dev(users), H, dev(items_bf), dev(items), None, dev(row_bits), None, None, k, n_items=N, stats=stats)
    bv, bi = ops.catalog_topk(dev(bf(users)), H, dev(items_bf), None, dev(row_bits), None, None, k, n_items=N)
    torch.cuda.synchronize()
    sc = (users.double() @ items.double().T).numpy()
    sc[:, 0] = -np.inf
    n_diff_bf16 = 0
    for r in range(B * H):
        order = np.lexsort((np.arange(N), -sc[r]))[:k + 1]
        got = oi[r].cpu().numpy()
        for j in range(k):
            gap_prev = sc[r, order[j - 1]] - sc[r, order[j]] if j else 1.0
            gap_next = sc[r, order[j]] - sc[r, order[j + 1]]
            if gap_prev > 2e-6 and gap_next > 2e-6:
                assert got[j] == order[j], (r, j)
        n_diff_bf16 += int((bi[r].cpu().numpy() != order[:k]).sum())
        np.testing.assert_allclose(ov[r].cpu().numpy(), sc[r, got], rtol=0, atol=2e-6)
    assert n_diff_bf16 > 0 and stats["uncertified_rows"] == 0 and k < stats["margin_mean"] <= 1024


def test_exact_topk_one_host_read_and_its_repair_path(ops):
    """Without `stats` the exact top-k reads its two verification flags in ONE host read (the bf16 pass leaves its flag on the
    device).  With a starved candidate budget the bf16 pass fails rows; the deferred attempt then notices the flag and repeats the
    pass with its own check-and-repair: same result as the path that checks at every stage."""
    g = torch.Generator().manual_seed(321)
    B, H, N, D, k = 8, 4, 60000, 256, 200
    users = HO.l2n(torch.randn(B * H, D, generator=g))
    items = HO.l2n(torch.randn(N, D, generator=g) + 0.3 * torch.randn(1, D, generator=g))
    row_bits = torch.full((B * H,), -(1 << 31), dtype=torch.int32)
    items_bf = bf(items)
    args = (dev(users), H, dev(items_bf), dev(items), None, dev(row_bits), None, None, k)
    stats = {}
    ov_s, oi_s = ops.catalog_topk_exact(*args, n_items=N, stats=stats)                 # checks at every stage
    ov_d, oi_d = ops.catalog_topk_exact(*args, n_items=N)                              # one deferred read
    assert torch.equal(oi_s, oi_d) and torch.equal(ov_s, ov_d)                       # same kernels on the same candidates
    stats2 = {}
    ops.catalog_topk_exact(*args, n_items=N, stats=stats2, target=k + 8)               # starved: the bf16 pass flags rows ...
    assert stats2["flagged_rows"] > 0
    ov_r, oi_r = ops.catalog_topk_exact(*args, n_items=N, target=k + 8)                # ... and the deferred attempt comes back for them
    assert torch.equal(oi_s, oi_r) and float((ov_s - ov_r).abs().max()) <= 2e-6     # (repaired rows: candidates in another order)


def test_topk_bookkeeping_kernels(ops):
    """pick_tau / flag / margin_count / uncertified against the elementwise expressions they replaced."""
    from mhr_amd import lib
    g = torch.Generator().manual_seed(5)
    n, k2, kk, margin = 1000, 64, 20, 2.0 ** -7
    kth1, kth2 = torch.randn(n, generator=g), torch.randn(n, generator=g)
    kth2[::7] = float("-inf")
    st1, st2 = (torch.rand(n, generator=g) < 0.05).int(), (torch.rand(n, generator=g) < 0.05).int()
    tau = torch.empty(n, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    d = {k_: dev(v) for k_, v in dict(kth1=kth1, kth2=kth2, st1=st1, st2=st2).items()}
    lib.call("mhr_topk_pick_tau", d["kth1"].data_ptr(), d["kth2"].data_ptr(), d["st1"].data_ptr(), d["st2"].data_ptr(), n, tau.data_ptr(), st)
    ok = (st1 == 0) & (st2 == 0)
    want = torch.where(torch.isfinite(kth2) & ok, kth2, torch.where(ok, kth1, torch.full((n,), float("-inf"))))
    assert torch.equal(tau.cpu(), want)
    cnt = torch.randint(0, 40, (n,), generator=g).int()
    bits = torch.where(torch.rand(n, generator=g) < 0.1, torch.zeros(n, dtype=torch.int32), torch.ones(n, dtype=torch.int32))
    stt = (torch.rand(n, generator=g) < 0.02).int()
    flagged, anyf = torch.empty(n, dtype=torch.bool, device="cuda"), torch.empty(2, dtype=torch.int32, device="cuda")
    dc, db, ds = dev(cnt), dev(bits), dev(stt)
    lib.call("mhr_topk_flag", ds.data_ptr(), dc.data_ptr(), db.data_ptr(), tau.data_ptr(), 20, n, flagged.data_ptr(), anyf.data_ptr(), st)
    want_f = (stt != 0) | ((cnt < 20) & (bits != 0) & torch.isfinite(want))
    assert torch.equal(flagged.cpu(), want_f) and int(anyf[0]) == int(want_f.any())
    zeros = torch.zeros(n, dtype=torch.int32, device="cuda")
    lib.call("mhr_topk_flag", zeros.data_ptr(), (dc + 100).data_ptr(), db.data_ptr(), tau.data_ptr(), 20, n, flagged.data_ptr(),
             anyf.data_ptr(), st)
    assert not bool(flagged.any()) and int(anyf[0]) == 0                              # written, not or-ed
    bv = torch.sort(torch.randn(n, k2, generator=g) * 0.01, dim=1, descending=True).values
    bv[::9, 40:] = float("-inf")
    dbv, mc = dev(bv), torch.empty(n, dtype=torch.int32, device="cuda")
    lib.call("mhr_topk_margin_count", dbv.data_ptr(), n, k2, kk, margin, mc.data_ptr(), st)
    want_c = ((bv >= bv[:, kk - 1:kk] - margin) & torch.isfinite(bv)).sum(1).int()
    assert torch.equal(mc.cpu(), want_c)
    full = torch.empty(n, dtype=torch.bool, device="cuda")
    lib.call("mhr_topk_uncertified", mc.data_ptr(), dbv.data_ptr(), k2, kk, tau.data_ptr(), 1, margin, n, full.data_ptr(), anyf[1:2].data_ptr(), st)
    kth = bv[:, kk - 1]
    want_u = (want_c >= k2) | (torch.isfinite(kth) & torch.isfinite(want) & (kth - margin < want))
    assert torch.equal(full.cpu(), want_u) and int(anyf[1]) == int(want_u.any())

"""The streamed form of the HSTU attention (csrc/attention_stream.hip: K / V or Q / dO tiles through a two-slot LDS ring, any
sequence length) against the pinned oracle (`kernel_oracles.attn_oracle` = oracle/hstu_oracle.py:hstu_attention on the
kernels' operand precision; reference model/IDNet/hstu.py:137-160) - at lengths the resident form cannot hold (L = 1024,
2048) and, forced through MHR_ATTN_STREAM=1, on the ragged small shapes of test_gpu_kernels.py."""
import os
import sys

import pytest
import torch

from conftest import ROOT
from kernel_oracles import attn_oracle
from oracle import hstu_oracle as HO

pytestmark = pytest.mark.gpu
CODE = os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if CODE not in sys.path:
        sys.path.insert(0, CODE)
    import mhr_amd  # noqa: F401
    from mhr_amd import ops as _ops
    return _ops


def bf(t):
    return t.to(torch.bfloat16).float()


def _case(B, L, Hh, hd, seed, pad_front=True):
    g = torch.Generator().manual_seed(seed)
    D = Hh * hd
    h = bf(torch.randn(B * L, 4 * D, generator=g))
    valid = torch.rand(B, L, generator=g) > 0.2
    if pad_front:
        valid[0, : L // 3] = False                       # whole blocks of padding keys: skipped by the ring
    if B > 1:
        valid[1, 40:200] = False                         # a hole in the middle
    d_out = bf(torch.randn(B * L, D, generator=g) * 0.5)
    return h, valid, d_out


def _check(ops, B, L, Hh, hd, seed, save_act):
    h, valid, d_out = _case(B, L, Hh, hd, seed)
    D = Hh * hd
    kv = valid.to(torch.uint8).cuda()
    hd_, dd = h.bfloat16().cuda(), d_out.bfloat16().cuda()
    out, act = ops.hstu_attn_fwd(hd_, kv, B, L, Hh, hd, save_act=save_act)
    ref, gref = attn_oracle(h, valid, B, L, Hh, hd, d_out)
    torch.cuda.synchronize()
    assert float((out.float().cpu() - ref).abs().max()) < 2 ** -7 * float(ref.abs().max())
    if save_act:
        act_ref = bf(HO.silu(h.float()))
        assert torch.equal(act.float().cpu()[:, :D], act_ref[:, 2 * D:3 * D])          # q
        assert torch.equal(act.float().cpu()[:, D:2 * D], act_ref[:, 3 * D:])          # k
        assert torch.equal(act.float().cpu()[:, 2 * D:], act_ref[:, D:2 * D])          # v
    dh = torch.zeros(B * L, 4 * D, dtype=torch.bfloat16).cuda()
    ops.hstu_attn_bwd(hd_, act if save_act else None, kv, dd, dh, B, L, Hh, hd)
    dh2 = torch.zeros_like(dh)
    ops.hstu_attn_bwd(hd_, None, kv, dd, dh2, B, L, Hh, hd)
    torch.cuda.synchronize()
    assert torch.equal(dh, dh2)                                                        # recomputed activations: same bits; no atomics
    got = dh.float().cpu()
    assert float(got[:, :D].abs().max()) == 0.0                                        # u block untouched
    for name, sl in (("dv", slice(D, 2 * D)), ("dq", slice(2 * D, 3 * D)), ("dk", slice(3 * D, 4 * D))):
        gs = float(gref[:, sl].abs().max())
        err = float((got[:, sl] - gref[:, sl]).abs().max())
        assert err < 2e-2 * gs, (name, err, gs)
    dead = ~valid.reshape(-1)
    assert float(got[dead][:, D:2 * D].abs().max()) == 0.0 and float(got[dead][:, 3 * D:].abs().max()) == 0.0
    return out, dh


@pytest.mark.parametrize("B,L,Hh,hd", [(2, 1024, 2, 64), (1, 2048, 2, 32), (1, 1024, 1, 128), (2, 1000, 3, 32), (1, 1500, 1, 16)])
def test_long_sequences_only_the_streamed_form_can_hold(ops, B, L, Hh, hd):
    _check(ops, B, L, Hh, hd, 100 + L + hd, save_act=(hd == 64))


@pytest.mark.parametrize("B,L,Hh,hd", [(3, 12, 2, 8), (2, 40, 4, 16), (2, 33, 2, 32), (2, 200, 8, 32), (1, 70, 2, 64), (1, 50, 1, 128),
                                       (2, 300, 2, 64), (2, 512, 2, 64), (2, 260, 3, 32), (1, 257, 1, 16), (1, 290, 1, 24),
                                       (2, 129, 1, 32)])
def test_streamed_form_on_the_resident_forms_shapes(ops, monkeypatch, B, L, Hh, hd):
    monkeypatch.setenv("MHR_ATTN_STREAM", "1")
    out_s, dh_s = _check(ops, B, L, Hh, hd, 7 + L, save_act=(L % 2 == 0))
    monkeypatch.setenv("MHR_ATTN_STREAM", "0")
    out_r, dh_r = _check(ops, B, L, Hh, hd, 7 + L, save_act=False)
    # same tile order, same arithmetic: the two forms agree to the last bit of the forward and to bf16 rounding in the backward
    assert torch.equal(out_s, out_r)
    assert float((dh_s.float() - dh_r.float()).abs().max()) <= 2 ** -6 * float(dh_r.float().abs().max())

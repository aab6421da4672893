"""The train step replayed from a hipGraph (REC/trainer/trainer.py:_StepGraph) against the same step issued launch by launch.

What a replayed launch cannot take as a kernel argument comes from device memory: the dropout step counter
(`mhr_ln_gate_*`'s step_seed), the optimizer step and its constants (`mhr_adam_flat` / `mhr_adam_rows_lazy`'s step_dev +
the constants' history).  These tests pin (a) that the device-side forms give the bits of the host-side forms, and
(b) that a trainer whose steps are replayed follows the trajectory of one whose steps are host-issued - same dropout
masks, same lr schedule, same lazy-table bookkeeping - on the reference's step semantics (trainer.py:494-536)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
CODE = os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code")


@pytest.fixture(scope="module")
def rec():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if CODE not in sys.path:
        sys.path.insert(0, CODE)
    import REC  # noqa: F401
    return REC


def test_device_side_seed_and_step_give_the_host_side_bits(rec):
    from mhr_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    rows, D = 300, 256
    h = torch.randn(rows, 4 * D, generator=g, device=dev).bfloat16()
    a = torch.randn(rows, D, generator=g, device=dev).bfloat16()
    d_o = torch.randn(rows, D, generator=g, device=dev).bfloat16()
    step, layer_part = 4321, 3 * 7919 + 2 * 104729
    host_seed = (step * 1000003 + layer_part) & 0x7FFFFFFFFFFFFFFF
    ctr = torch.tensor([step], dtype=torch.int64, device=dev)
    o1, m1, r1 = ops.ln_gate_fwd(h, a, D, torch.bfloat16, 1e-6, 0.25, host_seed)
    o2, m2, r2 = ops.ln_gate_fwd(h, a, D, torch.bfloat16, 1e-6, 0.25, layer_part, ctr)
    assert torch.equal(o1, o2) and 0.15 < float((o1 == 0).float().mean()) < 0.35
    dh1, dh2 = torch.zeros_like(h), torch.zeros_like(h)
    da1 = ops.ln_gate_bwd(d_o, h, a, m1, r1, dh1, D, 0.25, host_seed)
    da2 = ops.ln_gate_bwd(d_o, h, a, m2, r2, dh2, D, 0.25, layer_part, ctr)
    assert torch.equal(da1, da2) and torch.equal(dh1, dh2)
    # flat Adam: constants from the device-side history == constants derived from the arguments
    from mhr_amd import lib
    n = 10007
    w = torch.randn(n, generator=g, device=dev)
    gr = torch.randn(n, generator=g, device=dev)
    wa, ma, va = w.clone(), torch.zeros_like(w), torch.zeros_like(w)
    wb, mb, vb = w.clone(), torch.zeros_like(w), torch.zeros_like(w)
    hist_h = torch.zeros(64, 4)
    hist = torch.zeros(64, 4, device=dev)
    sd = torch.zeros(1, dtype=torch.int64, device=dev)
    for s in (1, 2, 70):
        lr = 1e-3 / s
        ops.adam_flat(wa, gr, ma, va, s, lr, 0.5, weight_decay=0.01)
        lib.call("mhr_adam_consts", lr, 0.9, 0.999, 1e-8, 0.01, s, hist_h[s % 64].data_ptr())
        hist.copy_(hist_h)
        sd.fill_(s)
        ops.adam_flat(wb, gr, mb, vb, 999, 123.0, 0.5, weight_decay=0.01, hist=hist, step_dev=sd)     # arguments ignored
    assert torch.equal(wa, wb) and torch.equal(ma, mb) and torch.equal(va, vb)


def _trainer(rec, graph, dev, **over):
    import mhr_amd.synth as synth
    from REC.config.configurator import Config, apply_run_fixups
    from REC.trainer import Trainer
    from REC.utils import get_model
    kw = dict(MAX_ITEM_LIST_LENGTH=24, pred_len=4, eval_pred_len=4, n_layers=2, n_heads=2, item_embedding_size=64,
              hstu_embedding_size=64, loss='prior', num_prior_head=3, medusa_num_layers=1, eval_num_cats=3, num_negatives=256,
              device=dev, total_iters=200, eval_interval=0, checkpoint_dir=None, save_model_note="t",
              scheduler_args={'type': 'cosine', 'warmup': 0.05}, optim_args={'learning_rate': 2e-3, 'weight_decay': 0.01},
              hidden_dropout_prob=0.2, hip_graph=graph)
    kw.update(over)
    cfg = apply_run_fixups(Config(config_dict=synth.base_config(**kw)))
    data = synth.SyntheticData(cfg, 2000, dev, seed=7)
    cfg["int_to_category"] = data.int_to_category
    torch.manual_seed(11)
    model = get_model("HSTU")(cfg, data).to(dev)
    tr = Trainer(cfg)
    tr.setup_model(model)
    return tr, model, data


@pytest.mark.parametrize("loss", ["prior", "nce"])
def test_replayed_steps_follow_the_host_issued_trajectory(rec, loss):
    dev = torch.device("cuda", 0)
    over = dict(loss=loss) if loss == "prior" else dict(loss="nce", num_prior_head=1, medusa_num_layers=0, eval_num_cats=1, pred_len=1,
                                                        eval_pred_len=1)
    tr_e, m_e, data = _trainer(rec, False, dev, **over)
    tr_g, m_g, _ = _trainer(rec, True, dev, **over)
    assert all(torch.equal(a, b) for a, b in zip(m_e.state_dict().values(), m_g.state_dict().values()))
    batches = [data.train_batch(16) for _ in range(6)]
    n_steps = 70                                   # crosses the lazy table's 64-step flush
    le, lg = [], []
    for i in range(n_steps):
        b = batches[i % len(batches)]
        le.append(float(tr_e.train_step_fn(b)["loss"]))
        lg.append(float(tr_g.train_step_fn(b)["loss"]))
    assert tr_g.graph_active and not tr_e.graph_active
    assert tr_g._step_graph.n == n_steps - 3       # three host-issued warm-up steps, everything else replayed
    assert tr_g.train_step == tr_e.train_step == n_steps and tr_g.optimizer.step_count == tr_e.optimizer.step_count == n_steps
    assert m_g._step_seed == m_e._step_seed == n_steps
    # same dropout masks, same schedule: the two runs differ by the order of float atomics in the loss backward only
    le, lg = np.array(le), np.array(lg)
    assert np.all(np.isfinite(lg))
    np.testing.assert_allclose(lg[:8], le[:8], rtol=5e-4)           # replays start at step 4: same masks, same constants
    np.testing.assert_allclose(lg[:5], le[:5], rtol=2e-5)           # (fp32 gradients end to end: atomic-order noise is no longer rounded away in bf16)
    np.testing.assert_allclose(lg[:15], le[:15], rtol=2e-3)         # (two runs of either kind differ by as much: float atomics)
    np.testing.assert_allclose(lg, le, rtol=3e-2)                   # (atomic-order noise grows along 70 steps of lr 2e-3)
    assert le[-1] < le[0]
    sd_e, sd_g = m_e.state_dict(), m_g.state_dict()                 # state_dict() flushes the lazy table
    for k in sd_e:                                                  # (Adam turns float-atomic noise into sign flips of tiny updates:
        a, b = sd_e[k].float(), sd_g[k].float()                     #  single elements may drift, the bulk must not)
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().mean()) <= 1e-2 * scale and float((a - b).abs().max()) <= 0.3 * scale, k     # (two host-issued runs: max 0.155 x scale measured)
    # a different dropout mask would show: replaying with a frozen step counter must NOT reproduce the trajectory
    assert abs(lg[10] - lg[4]) > 0 or loss == "nce"


def test_packed_rows_replayed_steps_are_the_host_issued_steps_bit_for_bit(rec):
    """Batches that carry the loader's row capacity (`_mhr_rows_cap`): the encoder runs on packed rows and the trainer keeps one
    captured step graph per capacity.  In deterministic mode the replayed run - alternating between its per-capacity graphs - and
    the host-issued run agree bit for bit over 40 steps (losses, weights), and each differs from the window-row run only by
    rounding (another 32-row blocking in the attention, another M in the library GEMMs)."""
    import mhr_amd.synth as synth
    from mhr_amd import ops
    dev = torch.device("cuda", 0)
    ops.set_deterministic(True)
    try:
        tr_e, m_e, data = _trainer(rec, False, dev, hidden_dropout_prob=0.0)    # (dropout masks are keyed by the row index, which
        tr_g, m_g, _ = _trainer(rec, True, dev, hidden_dropout_prob=0.0)       #  packing changes: off for the window comparison)
        tr_w, m_w, _ = _trainer(rec, False, dev, hidden_dropout_prob=0.0)
        batches = [data.train_batch(16) for _ in range(6)]
        hinted = []
        for bt in batches:
            m2 = bt[2].clone()
            m2._mhr_rows_cap = synth.rows_capacity(int(bt[2][:, :24].sum()), bucket=32)
            hinted.append((bt[0], bt[1], m2, bt[3]))
        caps = {b[2]._mhr_rows_cap for b in hinted}
        assert len(caps) >= 2                                   # more than one capacity: the replayed run switches graphs
        le, lg, lw = [], [], []
        for i in range(40):
            le.append(float(tr_e.train_step_fn(hinted[i % 6])["loss"]))
            lg.append(float(tr_g.train_step_fn(hinted[i % 6])["loss"]))
            lw.append(float(tr_w.train_step_fn(batches[i % 6])["loss"]))
        assert tr_g.graph_active and len(tr_g._step_graphs) == len(caps)
        assert le == lg
        sd_e, sd_g = m_e.state_dict(), m_g.state_dict()
        for k in sd_e:
            assert torch.equal(sd_e[k], sd_g[k]), k
        np.testing.assert_allclose(np.array(le[:6]), np.array(lw[:6]), rtol=2e-3)
        assert le[-1] < le[0]
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("loss", ["prior", "nce"])
def test_deterministic_mode_makes_replayed_and_host_issued_runs_bitwise_equal(rec, loss):
    """`ops.set_deterministic(True)` (include/mhr.h: deterministic mode): the float-atomic reductions of the loss backward - the
    negative-side gradient tiles, the suppressed-pair corrections, d(logit_scale), the per-offset loss sums, the bias column sums -
    take their order-independent forms.  Two host-issued runs then agree bit for bit, and so do a replayed and a host-issued run
    over all 70 steps (across the lazy table's flush): a stale replayed constant, a wrong dropout counter or a missed weight
    refresh cannot hide behind atomic noise any more."""
    from mhr_amd import ops
    dev = torch.device("cuda", 0)
    over = dict(loss=loss) if loss == "prior" else dict(loss="nce", num_prior_head=1, medusa_num_layers=0, eval_num_cats=1, pred_len=1,
                                                        eval_pred_len=1)
    ops.set_deterministic(True)
    try:
        tr_a, m_a, data = _trainer(rec, False, dev, **over)
        tr_b, m_b, _ = _trainer(rec, False, dev, **over)
        tr_g, m_g, _ = _trainer(rec, True, dev, **over)
        batches = [data.train_batch(16) for _ in range(6)]
        n_steps = 70
        la, lb, lg = [], [], []
        for i in range(n_steps):
            b = batches[i % len(batches)]
            la.append(float(tr_a.train_step_fn(b)["loss"]))
            lb.append(float(tr_b.train_step_fn(b)["loss"]))
            lg.append(float(tr_g.train_step_fn(b)["loss"]))
        assert tr_g.graph_active and tr_g._step_graph.n == n_steps - 3
        assert la == lb, [i for i in range(n_steps) if la[i] != lb[i]][:5]                 # host-issued twice: same bits
        assert la == lg, [i for i in range(n_steps) if la[i] != lg[i]][:5]                 # replayed: same bits
        sd_a, sd_b, sd_g = m_a.state_dict(), m_b.state_dict(), m_g.state_dict()
        for k in sd_a:
            assert torch.equal(sd_a[k], sd_b[k]), k
            assert torch.equal(sd_a[k], sd_g[k]), k
        assert torch.equal(tr_a.optimizer.flat_m, tr_g.optimizer.flat_m) and torch.equal(tr_a.optimizer.t_v, tr_g.optimizer.t_v)
        assert la[-1] < la[0]
    finally:
        ops.set_deterministic(False)
    # and the default mode is the fast one again
    from mhr_amd import lib
    assert lib.load().mhr_get_deterministic() == 0


def test_replayed_evaluation_step_gives_the_host_issued_results(rec):
    """Trainer.evaluate replays the evaluation step (encoder -> heads -> full-catalog decode up to its verification flags) from a
    hipGraph after two host-issued batches; histories of different lengths share a padded capacity.  Same metrics, same per-head
    lists as the launch-by-launch form (reference trainer.py:698-729, 985-990)."""
    import mhr_amd.synth as synth
    from REC.config.configurator import Config, apply_run_fixups
    from REC.trainer import Trainer
    from REC.utils import get_model
    dev = torch.device("cuda", 0)
    res, lists = {}, {}
    for mode in (True, False):
        torch.manual_seed(1)
        cfgd = synth.base_config(MAX_ITEM_LIST_LENGTH=20, pred_len=4, eval_pred_len=4, n_layers=2, n_heads=2, item_embedding_size=64,
                                 hstu_embedding_size=64, loss='prior', medusa_num_layers=1, num_prior_head=3, num_segment_head=1,
                                 eval_num_cats=3, num_negatives=64, topk=[5, 10, 20], device=dev, total_iters=4, eval_interval=0,
                                 checkpoint_dir=None, save_model_note="t", hidden_dropout_prob=0.0, hip_graph_eval=mode)
        cfg = apply_run_fixups(Config(config_dict=cfgd))
        N = 6000
        data = synth.SyntheticData(cfg, N, dev)
        cfg["int_to_category"] = data.int_to_category
        model = get_model("HSTU")(cfg, data).to(dev)
        tr = Trainer(cfg)
        tr.setup_model(model)
        batches = [data.eval_batch(8, hist_extra=10 + 7 * i) for i in range(7)]     # histories of different lengths

        class Loader(list):
            item_tags = data.item_tags
        res[mode] = tr.evaluate(Loader(batches))
        tr.compute_item_feature(data.item_tags)
        model.eval()
        lists[mode] = []
        for b in batches:                               # (a replayed step's lists live in the graph's static tensors: clone)
            f = tr._full_sort_batch_eval(b)[0]
            lists[mode].append((f.indices.clone(), f.values.clone()))
        if mode:
            gs = [g for g in tr._eval_graphs.values() if g.graph is not None]
            assert len(gs) == 1 and getattr(tr, "_eval_graph_failed", None) is None
        else:
            assert "_eval_graphs" not in tr.__dict__
    assert res[True] == res[False]
    for a, b in zip(lists[True], lists[False]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_host_issued_steps_interleave_with_replays(rec):
    """bench.py's evented pass issues steps from the host between replays: counters, lazy-table state and the constants'
    history are shared, so the run is the same run."""
    dev = torch.device("cuda", 0)
    tr_a, m_a, data = _trainer(rec, True, dev, hidden_dropout_prob=0.0)
    tr_b, m_b, _ = _trainer(rec, True, dev, hidden_dropout_prob=0.0)
    batches = [data.train_batch(16) for _ in range(4)]
    for i in range(24):
        b = batches[i % 4]
        tr_a.train_step_fn(b)
        tr_b.train_step_fn(b, graph=False if 8 <= i < 14 else None)
    assert tr_a.graph_active and tr_b.graph_active and tr_b._step_graph.n == 24 - 3 - 6
    sd_a, sd_b = m_a.state_dict(), m_b.state_dict()
    for k in sd_a:
        a, b = sd_a[k].float(), sd_b[k].float()
        assert float((a - b).abs().max()) <= 1e-2 * float(a.abs().max()) + 1e-6, k
    assert torch.equal(tr_a.optimizer.last_step, tr_b.optimizer.last_step)


def test_resume_continues_the_run(rec, tmp_path):
    """A checkpoint carries the weights, the optimizer (moments, step, lazy-table bookkeeping), the lr-schedule position and the
    model's dropout step counter: the resumed trainer's next steps are the original run's next steps (reference
    trainer.py:319-340 / 350-372)."""
    dev = torch.device("cuda", 0)
    tr_a, m_a, data = _trainer(rec, True, dev)
    batches = [data.train_batch(16) for _ in range(4)]
    for i in range(7):
        tr_a.train_step_fn(batches[i % 4])
    tr_a.checkpoint_dir, tr_a.saved_model_name = str(tmp_path), "ck.pth"
    tr_a._save_checkpoint()
    la = [float(tr_a.train_step_fn(batches[(7 + i) % 4])["loss"]) for i in range(4)]
    tr_b, m_b, _ = _trainer(rec, True, dev)
    tr_b.resume(os.path.join(str(tmp_path), "ck.pth"))
    assert tr_b.train_step == 7 and tr_b.optimizer.step_count == 7 and m_b._step_seed == 7
    lb = [float(tr_b.train_step_fn(batches[(7 + i) % 4])["loss"]) for i in range(4)]
    np.testing.assert_allclose(lb, la, rtol=2e-3)                   # same dropout masks, same constants; float-atomic noise only
    tr_c, m_c, _ = _trainer(rec, True, dev)                         # without the step counter the masks restart: a different run
    tr_c.resume(os.path.join(str(tmp_path), "ck.pth"))
    m_c._step_seed = 0
    lc = [float(tr_c.train_step_fn(batches[(7 + i) % 4])["loss"]) for i in range(4)]
    assert max(abs(a - c) / abs(a) for a, c in zip(la, lc)) > 2e-3


def test_resume_into_a_trainer_with_a_captured_graph_reads_the_loaded_weights(rec, tmp_path):
    """A captured step reads the GEMM operands from the optimizer's bf16 shadows through baked pointers; load_state_dict()
    writes the fp32 masters only.  resume() (and any in-place weight load in front of a replay) must re-cast the shadows, or the
    first replayed step runs forward and backward on the OLD weights and applies that gradient to the loaded ones."""
    dev = torch.device("cuda", 0)
    tr, m, data = _trainer(rec, True, dev, hidden_dropout_prob=0.0)
    batches = [data.train_batch(16) for _ in range(4)]
    for i in range(7):
        tr.train_step_fn(batches[i % 4])
    assert tr.graph_active
    tr.checkpoint_dir, tr.saved_model_name = str(tmp_path), "ck.pth"
    tr._save_checkpoint()
    ck = os.path.join(str(tmp_path), "ck.pth")
    want = float(tr.train_step_fn(batches[3])["loss"])                # step 8 of the run, from the checkpointed state
    for i in range(40):                                               # move the weights well away from the checkpoint
        tr.train_step_fn(batches[i % 4])
    far = float(tr.train_step_fn(batches[3], graph=False)["loss"])
    assert abs(far - want) > 2e-2 * abs(want)                         # (otherwise a stale replay could not be told apart)
    tr.resume(ck)
    opt = tr.optimizer
    assert not opt.shadows_stale() and torch.equal(opt.flat_w16, opt.flat_w.to(torch.bfloat16))
    n_before = tr._step_graph.n
    got = float(tr.train_step_fn(batches[3])["loss"])
    assert tr._step_graph.n == n_before + 1                           # a replay, not a host-issued step
    assert abs(got - want) <= 2e-3 * abs(want), (got, want, far)
    # an in-place weight load WITHOUT resume(): the replay notices the stale shadows itself
    sd = torch.load(ck, map_location=dev, weights_only=True)
    for i in range(20):
        tr.train_step_fn(batches[i % 4])
    m.load_state_dict(sd["model"])
    tr.optimizer.load_state_dict(sd["optimizer"])
    tr.train_step, m._step_seed, tr._micro_step = int(sd["iter_idx"]), int(sd["step_seed"]), int(sd["micro_step"])
    assert opt.shadows_stale()
    got2 = float(tr.train_step_fn(batches[3])["loss"])
    assert abs(got2 - want) <= 2e-3 * abs(want), (got2, want)
    # and the same through a fresh trainer issuing every launch from the host: the reference point of both
    tr_e, m_e, _ = _trainer(rec, False, dev, hidden_dropout_prob=0.0)
    tr_e.resume(ck)
    ref = float(tr_e.train_step_fn(batches[3])["loss"])
    assert abs(ref - want) <= 2e-3 * abs(want)


def test_a_failed_capture_leaves_no_half_step_behind(rec, monkeypatch):
    """A capture that raises partway through the step has run part of a step's Python and launched nothing: the optimizer's
    deferred queues and the model's pending rows point at capture-pool tensors.  The Trainer drops them and trains on with
    host-issued launches - the same run as a trainer that never tried to capture."""
    from REC.model import multihead
    dev = torch.device("cuda", 0)
    tr_g, m_g, data = _trainer(rec, True, dev, hidden_dropout_prob=0.0)
    tr_e, m_e, _ = _trainer(rec, False, dev, hidden_dropout_prob=0.0)
    batches = [data.train_batch(16) for _ in range(4)]
    orig = multihead.MultiHeadDecoding._multihead_loss
    calls = {"n": 0}

    def flaky(self, *a, **kw):
        if torch.cuda.is_current_stream_capturing():
            calls["n"] += 1
            raise RuntimeError("injected failure inside the captured step (after the encoder queued its weight gradients)")
        return orig(self, *a, **kw)
    monkeypatch.setattr(multihead.MultiHeadDecoding, "_multihead_loss", flaky)
    lg, le = [], []
    for i in range(8):
        lg.append(float(tr_g.train_step_fn(batches[i % 4])["loss"]))
        le.append(float(tr_e.train_step_fn(batches[i % 4])["loss"]))
    assert calls["n"] == 1 and tr_g._graph_failed and not tr_g.graph_active and "injected" in tr_g.graph_failure
    assert not getattr(tr_g.optimizer, "_deferred", []) and not getattr(tr_g.optimizer, "_bias_q", [])
    np.testing.assert_allclose(lg, le, rtol=2e-3)
    assert np.all(np.isfinite(lg))
    # with hip_graph_required the failure is an error, not a silent change of mode
    tr_r, _, _ = _trainer(rec, True, dev, hidden_dropout_prob=0.0, hip_graph_required=True)
    with pytest.raises(RuntimeError, match="hip_graph_required"):
        for i in range(5):
            tr_r.train_step_fn(batches[i % 4])


@pytest.mark.parametrize("stacked", [True, False])
def test_deferred_weight_gradients_give_the_per_layer_reduction(rec, stacked):
    """With one backward per step the encoder's weight gradients are deferred to the optimizer step: as ONE batched split-K
    GEMM per projection over all layers (hstu_functional.WeightGradStack, stacked=True) or, without it, as per-layer split-K
    GEMMs whose partials go through ONE arena and ONE reduction launch (optim.FusedAdamW.enable_partial_arena).  A trainer with
    neither forms and sums them inside each layer's backward.  Same products, fp32 sums of bf16 partials: the trainers stay
    together (up to the partials' rounding and the float-atomic noise of the loss backward)."""
    from REC.model.IDNet import hstu as hstu_mod
    dev = torch.device("cuda", 0)
    flag = hstu_mod.WEIGHT_GRAD_STACK
    try:
        hstu_mod.WEIGHT_GRAD_STACK = stacked
        tr_a, m_a, data = _trainer(rec, False, dev, hidden_dropout_prob=0.0)
        tr_b, m_b, _ = _trainer(rec, False, dev, hidden_dropout_prob=0.0)
        tr_b.optimizer.enable_partial_arena(False)
        assert tr_a.optimizer._arena_on and not tr_b.optimizer._arena_on
        batches = [data.train_batch(64) for _ in range(4)]           # 64 x 24 tokens: the weight gradients split four ways
        for i in range(6):
            la = float(tr_a.train_step_fn(batches[i % 4])["loss"])
            lb = float(tr_b.train_step_fn(batches[i % 4])["loss"])
            assert abs(la - lb) <= 2e-3 * abs(lb)
    finally:
        hstu_mod.WEIGHT_GRAD_STACK = flag
    assert (tr_a.optimizer._arena is None) == stacked and tr_b.optimizer._arena is None
    assert not getattr(tr_a.optimizer, "_deferred", []) and not getattr(tr_a.optimizer, "_bias_q", [])
    sd_a, sd_b = m_a.state_dict(), m_b.state_dict()
    for k in sd_a:
        a, b = sd_a[k].float(), sd_b[k].float()
        assert float((a - b).abs().max()) <= 1e-2 * float(a.abs().max()) + 1e-6, k

"""CPU-side checks (no GPU): the C-ABI library builds, loads and exports every symbol of include/mhr.h; the
reference-shaped host classes keep the reference's registry surface and state_dict names; product metrics and the
LR schedule reproduce the reference's golden values; the data-parallel exchange is correct on 2 gloo ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

CODE = os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code")
if CODE not in sys.path:
    sys.path.insert(0, CODE)


@pytest.fixture(scope="session")
def built_lib():
    import __graft_entry__ as ge
    return ge.build_hip_library()


def test_library_exports_every_declared_symbol(built_lib):
    import mhr_amd
    from mhr_amd import lib
    assert os.path.exists(built_lib)
    names = lib.declared_symbols()
    assert len(names) >= 20 and "mhr_nce_fwd" in names and "mhr_catalog_score_emit" in names
    assert lib.check_exports()
    dll = lib.load()
    assert dll.mhr_abi_version() >= 1
    assert dll.mhr_last_error() is not None
    # argument validation happens on the host before any launch: a null pointer is rejected with a message
    rc = dll.mhr_layernorm_fwd(None, 0, None, 0, None, None, 4, 16, 1e-6, None)
    assert rc == -1 and b"null" in dll.mhr_last_error()
    rc = dll.mhr_catalog_score_emit(1, 4, 4, 1, 100, 48, 0, 1, None, 1, 1, None, None, 1, 1, 1, 16, None)
    assert rc == -1 and b"dim=48" in dll.mhr_last_error()


def test_workspace_queries(built_lib):
    """The C ABI's size queries are host functions (no launch): SURVEY.md 8b - the library allocates nothing, the caller sizes
    the scorers' candidate lists from these."""
    import mhr_amd.lib as L
    dll = L.load()
    # 1024 (user, head) rows over 453 938 items: 4 row tiles -> 16 slice groups x 8 XCDs = 128 slices, 2 lists per slice
    assert dll.mhr_catalog_emit_slices(1024, 453938) == 128
    assert dll.mhr_catalog_emit_slices(4, 453938) == 512 and dll.mhr_catalog_emit_slices(4, 100) == 8
    cap_s = 48
    assert dll.mhr_catalog_score_emit_sliced_workspace_bytes(1024, 453938, cap_s) == 1024 * 256 * (cap_s * 8 + 4)
    nw = dll.mhr_catalog_wide_slices(4096)
    assert dll.mhr_catalog_score_emit_wide_workspace_bytes(4096, 32) == 4096 * 4 * nw * (32 * 8 + 4)
    assert dll.mhr_catalog_score_emit_workspace_bytes(64, 4096) == 64 * (4096 * 8 + 4)
    assert dll.mhr_catalog_score_rows_dense_workspace_bytes(24, 453938) == 24 * (453938 * 8 + 4)
    assert dll.mhr_catalog_score_emit_sliced_workspace_bytes(0, 10, 10) == 0
    assert "mhr_pack_tiles_bytes" in L.RETURNS_INT64 and "mhr_catalog_score_rows_dense_workspace_bytes" in L.RETURNS_INT64


def test_product_path_has_no_cpu_fallback(built_lib):
    import mhr_amd  # noqa: F401
    from mhr_amd import ops
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.layernorm_fwd(torch.zeros(4, 16))
    src = open(os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "ops.py")).read()
    assert "oracle" not in src.replace("# oracle", "")          # the product never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd")):
        for f in files:
            if f.endswith(".py") and f != "smoke.py":
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def _cfg(name):
    g = load_golden(name)
    c = json.loads(str(g["cfg/json"]))
    c["int_to_category"] = {int(k): v for k, v in c["int_to_category"].items()}
    return g, c


class FakeData:
    def __init__(self, c):
        self.item_num, self.category_counts, self.category_to_int = c["item_num"], c["category_counts"], c["category_to_int"]


@pytest.mark.parametrize("name", ["hstu_nce_tiny", "hstu_nce_multistep", "hstu_prior_hier", "hstu_prior_mult",
                                  "hstu_prior_additive", "hstu_prior_proj", "predict_mult", "predict_additive"])
def test_registry_and_state_dict_names_match_reference(name):
    import REC  # noqa: F401
    from REC.config.configurator import Config
    from REC.utils import InputType, get_model
    g, c = _cfg(name)
    cls = get_model("HSTU")
    assert cls.__name__ == "HSTU" and cls.input_type == InputType.SEQ
    model = cls(Config(config_dict=c), FakeData(c))
    ref = {k[2:]: v.shape for k, v in g.items() if k.startswith("w/")}
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(mine) == set(ref), (set(mine) ^ set(ref))
    for k in ref:
        assert tuple(ref[k]) == mine[k], k
    model.load_state_dict({k[2:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("w/")}, strict=True)
    with pytest.raises(ValueError):
        get_model("NoSuchModel")
    with pytest.raises(RuntimeError, match="MI355X"):
        model(tuple(torch.from_numpy(g["in/" + k]) for k in ("items", "neg_items", "mask", "tags"))) if "in/items" in g else (_ for _ in ()).throw(RuntimeError("MI355X"))


def test_optimizer_state_without_layout_record_is_refused_when_the_order_differs():
    """FusedAdamW lays the decoding heads' weights (and biases) back to back (`adjacent_parameters()`), so its flat moment
    buffers are NOT in named_parameters() order for multi-head models.  A checkpoint from before the layout was recorded has
    no 'layout' key: loading it must fail loudly instead of shifting the moments onto the wrong parameters; for a model whose
    order is the plain one it loads."""
    import REC  # noqa: F401
    import mhr_amd  # noqa: F401
    from mhr_amd.optim import FusedAdamW
    from REC.config.configurator import Config
    from REC.utils import get_model
    for name, reordered in (("hstu_prior_mult", True), ("hstu_nce_tiny", False)):
        g, c = _cfg(name)
        model = get_model("HSTU")(Config(config_dict=c), FakeData(c))
        opt = FusedAdamW(model)
        plain = [n for n, p in model.named_parameters() if p is not model.item_embedding.weight]
        assert ([n for n, _ in opt.layout] != plain) == reordered, name
        sd = opt.state_dict()
        opt.load_state_dict(sd)                                       # with its layout record: fine
        old = {k: v for k, v in sd.items() if k != "layout"}
        if reordered:
            with pytest.raises(RuntimeError, match="layout"):
                opt.load_state_dict(old)
        else:
            opt.load_state_dict(old)
        bad = dict(sd, layout=[[n + "_x", k] for n, k in sd["layout"]])
        with pytest.raises(RuntimeError, match="layout"):
            opt.load_state_dict(bad)
        # a state saved when every parameter's slot was padded to 4 elements (no 'pad' record): loaded slot by slot
        opt.flat_m.copy_(torch.arange(opt.flat_m.numel(), dtype=torch.float32))
        want = [opt.flat_m[p._mhr_flat_off:p._mhr_flat_off + p.numel()].clone() for p in opt.dense]
        packed4 = torch.cat([torch.cat([w_, w_.new_zeros((-w_.numel()) % 4)]) for w_ in want])
        sd4 = {k: v for k, v in opt.state_dict().items() if k != "pad"}
        sd4["flat_m"] = packed4
        sd4["flat_v"] = packed4.clone()
        opt.flat_m.zero_()
        opt.load_state_dict(sd4)
        for p, w_ in zip(opt.dense, want):
            assert torch.equal(opt.flat_m[p._mhr_flat_off:p._mhr_flat_off + p.numel()], w_)
            assert p._mhr_flat_off % 8 == 0                           # fp32 views and bf16 shadows both 16-byte aligned


def test_dead_row_skipping_is_off_for_models_whose_switch_loss_reads_padding_positions():
    """The prior switch's category loss runs over every position of the window, padding included (reference hstu.py:757-806):
    such models must not let the encoder skip the rows in front of the first valid key."""
    import REC  # noqa: F401
    from REC.config.configurator import Config
    from REC.utils import get_model
    for name, want in (("hstu_prior_mult", True), ("hstu_nce_tiny", True), ("hstu_switch_inout_asym", False)):
        g, c = _cfg(name)
        model = get_model("HSTU")(Config(config_dict=c), FakeData(c))
        assert model._dead_rows_allowed() is want, name
        if not want:
            model.switch_last_only = True
            assert model._dead_rows_allowed() is True


def test_unknown_head_interaction_raises():
    from REC.config.configurator import Config
    from REC.utils import get_model
    g, c = _cfg("hstu_nce_tiny")
    c["head_interaction"] = "bogus"
    with pytest.raises(ValueError, match="Unknown head_interaction"):
        get_model("HSTU")(Config(config_dict=c), FakeData(c))
    c["head_interaction"] = "multiplicative"
    c["loss"] = "bpr"
    with pytest.raises(NotImplementedError):
        get_model("HSTU")(Config(config_dict=c), FakeData(c))


def test_truncated_normal_contract():
    from REC.model.IDNet.hstu import truncated_normal
    torch.manual_seed(0)
    x = truncated_normal(torch.empty(200000), 0.0, 0.02)
    assert float(x.abs().max()) < 2 * 0.02 + 1e-6 or float((x.abs() >= 0.04).float().mean()) < 1e-4   # 4 draws all outside: p ~ 4e-6
    assert abs(float(x.std()) - 0.02 * 0.88) < 0.002


def test_config_semantics_and_run_fixups():
    from REC.config.configurator import Config, apply_run_fixups
    cfg = Config(config_dict={"eval_pred_len": 8, "metrics_pred_len_list": [1], "loss": "nce", "medusa_num_layers": 0,
                              "prior_switch": "in", "dataset": "Pixel8M", "topk": 10})
    assert cfg["missing"] is None and cfg.get("missing", 3) == 3 and cfg.topk == [10]
    with pytest.raises(AttributeError):
        cfg.missing
    apply_run_fixups(cfg)
    assert cfg["metrics_pred_len_list"] == [0, 3, 7] and cfg["prior_switch"] is None


@pytest.mark.parametrize("name", ["collector_combine", "collector_additive", "collector_k200", "collector_single", "collector_average"])
def test_product_metrics_vs_reference_golden(name):
    from REC.evaluator import Evaluator
    from REC.evaluator.collector import DataStruct
    g = load_golden(name)
    topk = [int(k) for k in g["cfg/topk"]]
    C = int(g["cfg/C"])
    cfg = dict(metrics=["Recall", "NDCG"], shared_metrics=["Entropy"], topk=topk, eval_num_cats=C, eval_by_cat=False,
               eval_pred_len=4, outlier_user_metrics=None, int_to_category={i: f"cat{i}" for i in range(C)})

    class Cfg(dict):
        def __getitem__(self, k):
            return dict.get(self, k)

        def get(self, k, d=None):
            v = dict.get(self, k)
            return d if v is None else v
    ev = Evaluator(Cfg(cfg))
    for p in [int(x) for x in g["cfg/pred_len_list"]]:
        st = DataStruct()
        st.set('rec.topk', torch.from_numpy(np.array(g[f"out/topk_{p}"])))
        res = ev.evaluate(st, pred_len=p)
        for k, v in res.items():
            np.testing.assert_allclose(v, float(g[f"out/m{p}/{k}"]), rtol=1e-12, err_msg=k)
    st = DataStruct()
    st.set('rec.rec_tags', torch.from_numpy(np.array(g["out/rec_tags"])))
    for k, v in ev.evaluate(st, pred_len=-1).items():
        np.testing.assert_allclose(v, float(g[f"out/shared/{k}"]), rtol=1e-6, err_msg=k)


def test_lr_schedule_vs_reference_golden():
    from REC.utils.lr_scheduler import cosine_warmup_factor, get_cosine_schedule_with_warmup
    g = load_golden("schedule_adam")
    np.testing.assert_allclose([1e-3 * cosine_warmup_factor(s, 10, 100) for s in range(100)], g["lrs"], rtol=1e-12)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    sch = get_cosine_schedule_with_warmup(opt, 10, 100)
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)


def test_synthetic_batch_contract():
    import mhr_amd.synth as synth
    from REC.config.configurator import Config, apply_run_fixups
    spec = synth.CONFIGS["cfg1"]
    cfg = apply_run_fixups(Config(config_dict=dict(spec["cfg"], MAX_ITEM_LIST_LENGTH=20, num_negatives=256)))
    d = synth.SyntheticData(cfg, 5000, "cpu", world=2)
    items, neg, mask, tags = d.train_batch(16)
    L, P, C = 20, 8, 4
    assert items.shape == (16, L + P) and items.dtype == torch.int64 and int(items.min()) >= 1 and int(items.max()) < 5000
    assert neg.shape == (16, C + 1, 8) and mask.shape == (16, L + P) and tags.shape == (16, L + P, C)
    assert set(mask.unique().tolist()) <= {0, 1} and bool((mask[:, L:] == 1).all())
    assert bool((mask[:, 1:] >= mask[:, :-1]).all())                                   # front padding only
    for c in range(C):                                                                  # by-category pools
        assert bool(d.item_tags[neg[:, c].reshape(-1), c].all())
    assert bool(d.item_tags[1:].any(dim=1).all())
    ev = d.eval_batch(5)
    assert ev[1].shape == (5, L) and ev[2].shape == (5, 8) and ev[6].shape == (5, 8, C)
    assert bool(((ev[1] == 0).long()[:, 1:] <= (ev[1] == 0).long()[:, :-1]).all())     # zeros only in front
    assert int(ev[3][0].max()) <= 4


DP_SCRIPT = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
import mhr_amd
from mhr_amd import distributed as D
from REC.model.basemodel import all_gather_ids
g = torch.Generator().manual_seed(100 + rank)
N, Dm, n_priv, n_neg = 50, 8, 12, 5
priv_ids = torch.randint(1, N, (n_priv,), generator=g)
my_negs = torch.randint(1, N, (n_neg,), generator=g)
all_negs = all_gather_ids(my_negs).reshape(-1)                       # same on every rank, rank order
ids_all = torch.cat([priv_ids, all_negs])
rows = torch.randn(ids_all.numel(), Dm, generator=g)
dense_local = torch.zeros(N, Dm).index_add_(0, ids_all, rows)
ref = dense_local.clone(); dist.all_reduce(ref)                        # what DDP would produce (sum; /W applied later)
ids, rp, rs = D.exchange_sparse_rows(ids_all, rows.clone(), n_priv, wire_dtype=torch.float32)
dense_comb = torch.zeros(N, Dm).index_add_(0, ids, torch.cat([rp, rs]))
ids_b, rp_b, rs_b = D.exchange_sparse_rows(ids_all, rows.clone(), n_priv)           # default wire dtype: bf16 private rows
dense_bf = torch.zeros(N, Dm).index_add_(0, ids_b, torch.cat([rp_b.float(), rs_b]))
ok_bf = rp_b.dtype == D.ROWS_WIRE_DTYPE and bool(torch.equal(ids_b, ids)) and bool(torch.allclose(dense_bf, ref, atol=3e-2, rtol=2 ** -7))
flat = torch.full((7,), float(rank + 1)); D.allreduce_mean_(flat)
vec = D.allreduce_metric_sums(torch.tensor([1.0 + rank, 10.0], dtype=torch.float64))
ok = ok_bf and bool(torch.allclose(dense_comb, ref, atol=1e-5)) and bool(torch.allclose(flat, torch.full((7,), (1 + world) / 2 * 1.0))) \
     and vec.tolist() == [sum(1.0 + r for r in range(world)), 10.0 * world] and all_negs.numel() == world * n_neg
gathered = [None] * world
dist.all_gather_object(gathered, all_negs.tolist())
ok = ok and all(x == gathered[0] for x in gathered)
print(json.dumps({"rank": rank, "ok": ok}))
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_data_parallel_exchange_two_gloo_ranks(tmp_path, built_lib):
    script = tmp_path / "dp.py"
    script.write_text(DP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, CODE], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o, e[-2000:])
        assert json.loads(o.strip().splitlines()[-1])["ok"]


def test_kernel_oracles_are_the_pinned_oracle_functions():
    """The oracles the GPU kernel tests compare with (tests/kernel_oracles.py) are the fixture-pinned `HO.hstu_attention` /
    `HO.nce_logits` / `HO.token_ce` with a rounding hook for the kernels' bf16 operands: with the hook off they return the
    pinned functions' values bit for bit, and the hook moves them by no more than bf16 rounding can."""
    import math
    import torch
    from kernel_oracles import attn_oracle, bf16_round, nce_oracle
    from oracle import hstu_oracle as HO
    g = torch.Generator().manual_seed(0)
    B, L, Hh, hd = 2, 19, 2, 8
    D = Hh * hd
    h = torch.randn(B * L, 4 * D, generator=g).to(torch.bfloat16)
    valid = torch.rand(B, L, generator=g) > 0.2
    out, _ = attn_oracle(h, valid, B, L, Hh, hd, operand_round=None)
    u, v, q, k = torch.split(HO.silu(h.float()), [D, D, D, D], dim=-1)
    want = HO.hstu_attention(q.view(B, L, D), k.view(B, L, D), v.view(B, L, D), valid, Hh).reshape(B * L, D)
    assert torch.equal(out, want)
    out_r, _ = attn_oracle(h, valid, B, L, Hh, hd)
    assert 0 < float((out_r - want).abs().max()) <= 2 ** -6 * float(want.abs().max())
    T, Nn = 23, 40
    qq, pp = torch.randn(T, D, generator=g) * 2, torch.randn(T, D, generator=g)
    negs = HO.l2n(torch.randn(Nn, D, generator=g))
    negs[3] = HO.l2n(pp[5][None])[0]                                       # a false negative
    ls = torch.tensor(math.log(20.0))
    loss, logits, keep, neg, pos = nce_oracle(qq, pp, negs, ls, 0.99, operand_round=None)
    want_logits = HO.nce_logits(qq, pp, negs, ls, 0.99)
    assert torch.equal(logits, want_logits) and torch.equal(loss, HO.token_ce(want_logits))
    assert not bool(keep[5, 3]) and int((~keep).sum()) >= 1 and bool(torch.isinf(neg[5, 3]))
    loss_r, *_ = nce_oracle(qq, pp, negs, ls, 0.99)
    assert 0 < float((loss_r - loss).abs().max()) <= 0.25
    assert float((bf16_round(qq) - qq).abs().max()) <= 2 ** -8 * float(qq.abs().max())


def test_bench_launches_its_own_ranks_and_checks_the_world_size(monkeypatch):
    """`python bench.py --gpus N` is the documented contract: with N > 1 and no launcher around it, bench.py starts N ranks
    under torch.distributed.run on 127.0.0.1 (before anything touches the GPU) and relays their exit code; under a launcher
    whose WORLD_SIZE disagrees with --gpus it refuses to run (round-1 advice: --gpus used to be ignored)."""
    import importlib
    import subprocess
    import sys as _sys
    bench = importlib.import_module("bench")
    seen = {}

    class _R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                               # the launcher's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code) and "--gpus 4" in str(e.value.code)

import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
import mhr_amd.synth as synth
from mhr_amd import ops
from REC.config.configurator import Config, apply_run_fixups
from REC.trainer import Trainer
from REC.utils import get_model
dev = torch.device("cuda", 0)
spec = synth.CONFIGS["cfg1"]
cfg = apply_run_fixups(Config(config_dict=dict(spec["cfg"], device=dev, total_iters=10, eval_interval=0, checkpoint_dir=None, save_model_note="x")))
data = synth.SyntheticData(cfg, spec["item_num"], dev)
cfg["int_to_category"] = data.int_to_category
model = get_model("HSTU")(cfg, data).to(dev)
tr = Trainer(cfg); tr.setup_model(model)
tr.compute_item_feature(data.item_tags)
batches = [data.eval_batch(256) for _ in range(3)]
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    b = batches[rep]
    t0 = T(); users = model._user_heads(b[1]); t1 = T()
    fused = model.predict_topk(b[1], tr.item_feature, tr.all_item_tags, b[6], b[3], k=200); t2 = T()
    tr.eval_collector.eval_batch_collect(fused, b[4], b[2]); t3 = T()
    print(f"rep {rep}: user_heads {1e3*(t1-t0):.2f} ms  predict_topk(total incl. heads) {1e3*(t2-t1):.2f} ms  collect {1e3*(t3-t2):.2f} ms")
# finer: inside predict_topk
users = model._user_heads(batches[0][1]).to(torch.bfloat16).view(256 * 4, -1).contiguous()
_, items_bf, tag_bits = model._item_cache
row_bits = model._row_constraints(256, batches[0][6], dev)
hu, hi = batches[0][3]
t0 = T(); order = torch.argsort(hu * model.item_num + hi); hist_items = hi[order].contiguous(); hp = torch.zeros(257, dtype=torch.int32, device=dev); hp[1:] = torch.cumsum(torch.bincount(hu, minlength=256), 0).int(); t1 = T()
st = {}
ov, oi = ops.catalog_topk(users, 4, items_bf, tag_bits, row_bits, hp, hist_items, 200, stats=st); t2 = T()
print(f"hist csr {1e3*(t1-t0):.2f} ms  catalog_topk {1e3*(t2-t1):.2f} ms  stats {st}")
ops.PROFILE = {k: [] for k in ("mhr_catalog_score_emit", "mhr_topk_select")}
ov, oi = ops.catalog_topk(users, 4, items_bf, tag_bits, row_bits, hp, hist_items, 200); torch.cuda.synchronize()
for k, v in ops.PROFILE.items():
    print(k, [round(a.elapsed_time(b), 3) for a, b in v])

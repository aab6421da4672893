import os, sys, time, torch, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
import mhr_amd.synth as synth
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
batches = [data.eval_batch(256) for _ in range(4)]
def step(i):
    fused, pu, pi, tt, _ = tr._full_sort_batch_eval(batches[i % 4])
    tr.eval_collector.eval_batch_collect(fused, pu, pi)
for i in range(4): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): step(i)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(i)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

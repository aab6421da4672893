"""cProfile of the host side of the cfg1 train step (where do the ~5 ms of issue time per step go?)."""
import cProfile, os, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
import mhr_amd.synth as synth
from REC.config.configurator import Config, apply_run_fixups
from REC.trainer import Trainer
from REC.utils import get_model
dev = torch.device("cuda", 0)
spec = synth.CONFIGS["cfg1"]
cfg = apply_run_fixups(Config(config_dict=dict(spec["cfg"], device=dev, total_iters=30000, eval_interval=0, checkpoint_dir=None, save_model_note="x")))
data = synth.SyntheticData(cfg, spec["item_num"], dev)
cfg["int_to_category"] = data.int_to_category
torch.manual_seed(2020)
model = get_model("HSTU")(cfg, data).to(dev)
tr = Trainer(cfg); tr.setup_model(model); tr.train_step = 3000
EVAL = len(sys.argv) > 1 and sys.argv[1] == "eval"
if EVAL:
    batches = [data.eval_batch(256) for _ in range(4)]
    tr.compute_item_feature(data.item_tags)

    def step(b):
        with torch.no_grad():
            fused, pu, pi, tt, _ = tr._full_sort_batch_eval(b)
            tr.eval_collector.eval_batch_collect(fused, pu, pi)
    tr.train_step_fn = step
else:
    batches = [data.train_batch(128) for _ in range(4)]
    model.train()
for i in range(6):
    tr.train_step_fn(batches[i % 4])
torch.cuda.synchronize()
N = 20
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    tr.train_step_fn(batches[i % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / N * 1e3, ct / N * 1e3, nc / N, f"{os.path.basename(fn)}:{line}:{name}"))
rows.sort(reverse=True)
print("self ms/step | cum ms/step | calls/step | function")
for r in rows[:45]:
    print(f"{r[0]:8.3f} {r[1]:8.3f} {r[2]:7.1f}  {r[3][:110]}")
print("total self ms/step", sum(r[0] for r in rows))

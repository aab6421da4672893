"""Does a captured step graph replay slower when OTHER step graphs are alive in the process?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
import mhr_amd  # noqa
import mhr_amd.synth as synth
from REC.config.configurator import Config, apply_run_fixups
from REC.trainer import Trainer
from REC.utils import get_model
dev = torch.device("cuda", 0)
spec = synth.CONFIGS["cfg1"]
cfg = apply_run_fixups(Config(config_dict=dict(spec["cfg"], device=dev, total_iters=30000, eval_interval=0, checkpoint_dir=None, save_model_note="x", hip_graph_required=True)))
data = synth.SyntheticData(cfg, spec["item_num"], dev, seed=2020)
cfg["int_to_category"] = data.int_to_category
torch.manual_seed(2020)
model = get_model("HSTU")(cfg, data).to(dev)
tr = Trainer(cfg); tr.setup_model(model); tr.train_step = 3000
model.train()
B = cfg["train_batch_size"]
A = [data.train_batch(B) for _ in range(4)]
def run(batches, n=40):
    for i in range(6): tr.train_step_fn(batches[i % len(batches)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): tr.train_step_fn(batches[i % len(batches)])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("A alone            : %.3f ms" % run(A))
print("A alone again      : %.3f ms" % run(A))
Bb = [data.train_batch(B - 8) for _ in range(4)]          # another signature -> another graph
for i in range(5): tr.train_step_fn(Bb[i % 4])
print("B alone            : %.3f ms" % run(Bb))
print("A alone            : %.3f ms" % run(A))
mix = [A[0], Bb[0], A[1], Bb[1], A[2], Bb[2], A[3], Bb[3]]
print("A, B alternating   : %.3f ms" % run(mix))
mix2 = [A[0], A[1], A[2], A[3], Bb[0], Bb[1], Bb[2], Bb[3]]
print("AAAA BBBB          : %.3f ms" % run(mix2))

"""Where the torch glue of one cfg1 train step goes: torch.profiler over 3 steps, ops grouped by the module-level phase
(record_function ranges) with counts, device time and host time.  python tools/step_profile.py [train|eval]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
import mhr_amd.synth as synth
from REC.config.configurator import Config, apply_run_fixups
from REC.trainer import Trainer
from REC.utils import get_model
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
spec = synth.CONFIGS["cfg1"]
cfg = apply_run_fixups(Config(config_dict=dict(spec["cfg"], device=dev, total_iters=30000, eval_interval=0, checkpoint_dir=None, save_model_note="x")))
data = synth.SyntheticData(cfg, spec["item_num"], dev)
cfg["int_to_category"] = data.int_to_category
torch.manual_seed(2020)
model = get_model("HSTU")(cfg, data).to(dev)
tr = Trainer(cfg); tr.setup_model(model); tr.train_step = 3000
batches = [data.train_batch(128) for _ in range(4)]
model.train()
for i in range(6):
    tr.train_step_fn(batches[i % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(3):
        tr.train_step_fn(batches[i % 4])
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    rows.append((dt / 3, e.count / 3, e.self_cpu_time_total / 3, e.key, [str(e.input_shapes)[:110]]))
rows.sort(key=lambda r: -r[0])
print("device us/step | calls/step | host us/step | op | python frames")
kern = [r for r in rows if not r[3].startswith('aten::') and r[2] == 0]
print('--- device kernels'); [print(f'{r[0]:9.1f} {r[1]:6.1f}  {r[3][:110]}') for r in kern[:45]]
print('kernel total us/step', sum(r[0] for r in kern))
print('--- aten ops by input shape')
rows = [r for r in rows if r[3].startswith('aten::')]
for r in rows[:45]:
    print(f"{r[0]:9.1f} {r[1]:6.1f} {r[2]:9.1f}  {r[3][:60]:60s} {' <- '.join(x.split('/')[-1][:70] for x in r[4])}")
print("total device us/step", sum(r[0] for r in rows), " host self us/step", sum(r[2] for r in rows))

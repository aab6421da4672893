"""Probe: which hipGraph capture pattern of the train step is fragile on this stack (run on the GPU box)."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))
import torch
import mhr_amd.synth as synth
from REC.config.configurator import Config, apply_run_fixups
from REC.trainer import Trainer
from REC.utils import get_model

def make(**kw):
    dev = torch.device("cuda", 0)
    base = dict(MAX_ITEM_LIST_LENGTH=12, pred_len=2, eval_pred_len=2, n_layers=1, n_heads=2, item_embedding_size=32,
                hstu_embedding_size=32, loss='prior', num_prior_head=3, eval_num_cats=3, medusa_num_layers=1,
                num_negatives=256, train_batch_size=32, total_iters=250, eval_interval=0, hidden_dropout_prob=0.0,
                checkpoint_dir=None, save_model_note="t", device=dev, optim_args={'learning_rate': 3e-3, 'weight_decay': 0.0},
                scheduler_args=None)
    base.update(kw)
    cfg = apply_run_fixups(Config(config_dict=synth.base_config(**base)))
    data = synth.SyntheticData(cfg, 400, dev, seed=1)
    cfg["int_to_category"] = data.int_to_category
    torch.manual_seed(0)
    model = get_model("HSTU")(cfg, data).to(dev)
    tr = Trainer(cfg); tr.setup_model(model)
    return tr, data

which = sys.argv[1]
if which == "single":
    tr, data = make()
    for i in range(12):
        out = tr.train_step_fn(data.train_batch(32))
    torch.cuda.synchronize(); print("single ok", float(out["loss"]), tr.graph_active, flush=True)
elif which == "alternate":
    tr, data = make()
    for ep in range(4):
        for i in range(9):
            out = tr.train_step_fn(data.train_batch(32))
        out = tr.train_step_fn(data.train_batch(12))
        torch.cuda.synchronize(); print("epoch", ep, float(out["loss"]), tr.graph_active, flush=True)
    print("alternate ok", flush=True)

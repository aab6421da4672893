"""Registry, seeding and early stopping (reference `code/REC/utils/utils.py:38-158`)."""
import importlib
import importlib.util
import os
import random

import numpy as np
import torch


def ensure_dir(path):
    os.makedirs(path, exist_ok=True)


def get_model(model_name):
    """`REC.model.IDNet.<name.lower()>` first, then `REC.model.HLLM.<...>`; returns the class named `model_name`."""
    fname = model_name.lower()
    for pkg in ('REC.model.IDNet', 'REC.model.HLLM'):
        mod_path = f'{pkg}.{fname}'
        if importlib.util.find_spec(mod_path) is not None:
            return getattr(importlib.import_module(mod_path), model_name)
    raise ValueError(f'`model_name` [{model_name}] is not the name of an existing model.')


def init_seed(seed, reproducibility=True):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def early_stopping(value, best, cur_step, max_step, bigger=True):
    """-> (best, cur_step, stop_flag, update_flag)"""
    improved = value >= best if bigger else value <= best
    if improved:
        return value, 0, False, True
    cur_step += 1
    return best, cur_step, cur_step > max_step, False


def calculate_valid_score(valid_result, valid_metric=None):
    return valid_result[valid_metric] if valid_metric else valid_result['Recall@10']


def dict2str(result_dict):
    return '    '.join(f'{k}: {v}' for k, v in result_dict.items())

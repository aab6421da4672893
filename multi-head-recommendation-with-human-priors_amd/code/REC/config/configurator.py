"""YAML-merge configuration with the reference's access rules (`code/REC/config/configurator.py:16-180`):
ordered merge of the given YAML files, `--key value` overrides, missing keys read as None, `.get(k, d)` returns d
for missing/None, attribute access for present keys."""
import copy

import yaml


class Config(object):
    def __init__(self, config_file_list=None, config_dict=None):
        self.final_config_dict = {}
        for path in config_file_list or []:
            with open(path, 'r', encoding='utf-8') as f:
                self.final_config_dict.update(yaml.load(f.read(), Loader=yaml.SafeLoader) or {})
        if config_dict:
            self.final_config_dict.update(copy.deepcopy(config_dict))
        topk = self.final_config_dict.get('topk')
        if isinstance(topk, int):
            self.final_config_dict['topk'] = [topk]
        self.final_config_dict.setdefault('device', 'cuda')

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise TypeError("index must be a str.")
        self.final_config_dict[key] = value

    def __getattr__(self, item):
        d = self.__dict__.get('final_config_dict')
        if d is None:
            raise AttributeError("'Config' object has no attribute 'final_config_dict'")
        if item in d:
            return d[item]
        raise AttributeError(f"'Config' object has no attribute '{item}'")

    def __getitem__(self, item):
        return self.final_config_dict.get(item, None)

    def get(self, key, default=None):
        res = self[key]
        return default if res is None else res

    def __contains__(self, key):
        if not isinstance(key, str):
            raise TypeError("index must be a str.")
        return key in self.final_config_dict

    def __str__(self):
        return '\n'.join(f'{k} = {v}' for k, v in self.final_config_dict.items())

    __repr__ = __str__


def apply_run_fixups(config):
    """The derived keys `code/run.py:90-104` computes before building the model."""
    lst = list(config['metrics_pred_len_list'] or [1])
    if config['eval_pred_len'] not in lst:
        lst.append(config['eval_pred_len'])
    half = config['eval_pred_len'] // 2
    if half > 0 and half not in lst:
        lst.append(half)
    assert all(isinstance(x, int) and x >= 0 for x in lst), "metrics_pred_len_list must be non-negative integers"
    config['metrics_pred_len_list'] = sorted(x - 1 for x in lst)
    if config['loss'] not in ['prior'] or config['medusa_num_layers'] == 0:
        config['prior_switch'] = None
    if 'merrec' in str(config['dataset']):
        config['category_by'] = 'event'
    return config

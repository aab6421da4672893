"""Runs the configured metrics over a collected DataStruct (reference `code/REC/evaluator/evaluator.py:11-40`)."""
from collections import OrderedDict

from .metrics import metrics_dict


class Evaluator(object):
    def __init__(self, config):
        self.config = config
        self.metrics = [m.lower() for m in config['metrics']]
        self.shared_metrics = [m.lower() for m in (config['shared_metrics'] or [])]
        unknown = [m for m in self.metrics + self.shared_metrics if m not in metrics_dict]
        if unknown:
            raise NotImplementedError(f"metrics {unknown} are outside the hot path (Recall / NDCG / Entropy are built)")
        self.metric_class = {m: metrics_dict[m](config) for m in self.metrics + self.shared_metrics}

    def evaluate(self, dataobject, pred_len=1):
        result = OrderedDict()
        for m in (self.shared_metrics if pred_len == -1 else self.metrics):
            result.update(self.metric_class[m].calculate_metric(dataobject, pred_len=pred_len))
        return result

from .collector import Collector, DataStruct  # noqa: F401
from .evaluator import Evaluator  # noqa: F401
from .metrics import NDCG, Entropy, Recall, metrics_dict  # noqa: F401

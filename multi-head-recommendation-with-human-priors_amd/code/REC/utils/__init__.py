from .enum_type import EvaluatorType, InputType  # noqa: F401
from .utils import calculate_valid_score, dict2str, early_stopping, ensure_dir, get_model, init_seed  # noqa: F401

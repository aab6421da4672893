"""Device-side batchers with the reference's batch contract (`REC/data/dataset/{trainset,evalset,collate_fn}.py`)."""
from .batcher import SeqEvalBatcher, SeqStore, SeqTrainBatcher  # noqa: F401

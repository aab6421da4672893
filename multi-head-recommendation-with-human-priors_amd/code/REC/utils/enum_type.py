from enum import Enum


class InputType(Enum):
    SEQ = 1
    PAIR = 2
    AUGSEQ = 3


class EvaluatorType(Enum):
    RANKING = 1
    VALUE = 2

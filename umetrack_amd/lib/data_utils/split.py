"""Dataset splits; folder names as in lib/data_utils/split.py:13-23."""
import enum


class Split(enum.Enum):
    TRAIN = "training"
    TEST = "testing"

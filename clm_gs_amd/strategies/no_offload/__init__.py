from .engine import baseline_accumGrads_impl, baseline_accumGrads_micro_step
from .gaussian_model import GaussianModelNoOffload

__all__ = ["GaussianModelNoOffload", "baseline_accumGrads_impl", "baseline_accumGrads_micro_step"]

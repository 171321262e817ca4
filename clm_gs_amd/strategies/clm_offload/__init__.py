from .engine import clm_offload_eval_one_cam, clm_offload_train_one_batch
from .gaussian_model import GaussianModelCLMOffload

__all__ = ["GaussianModelCLMOffload", "clm_offload_train_one_batch", "clm_offload_eval_one_cam"]

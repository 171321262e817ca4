"""naive_offload (scope row f4): every parameter on the host, whole-model copies per batch."""
from .engine import naive_offload_eval_one_cam, naive_offload_train_one_batch, render_single_image  # noqa: F401
from .gaussian_model import GaussianModelNaiveOffload  # noqa: F401

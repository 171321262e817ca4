"""naive_offload is OUT OF SCOPE for the hot path: BASELINE.json's
configs name only no_offload and clm_offload.  Signature-compatible stubs keep
`from strategies.naive_offload import ...` (train.py:43-57) importable and fail loudly."""


def _unsupported(*_a, **_k):
    raise NotImplementedError(
        "naive_offload is not built in clm_gs_amd (scope row f4); use no_offload or clm_offload")


class GaussianModelNaiveOffload:
    def __init__(self, *a, **k):
        _unsupported()


naive_offload_train_one_batch = _unsupported
naive_offload_eval_one_cam = _unsupported

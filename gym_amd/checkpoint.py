"""RunningMeanStd checkpoints in the reference's own file format.

`VecNormalize::save/load` (wrappers/vec_env/vec_normalize.h:79-108) call `torch::save(module, path)` /
`torch::load(module, path)` on `RunningMeanStdImpl`, whose `save()` writes three NON-buffer tensors named `mean`, `m_Var`
and `m_Count` into a `torch::serialize::OutputArchive` (common/normalizers/running_mean_std.h:28-40).  That archive is a
TorchScript module file, so the same bytes are produced here by scripting a module with three parameters of those names,
and read with `torch.jit.load`.  tests/test_ref_pin.py checks both directions against the reference's own save()/load()
(live in the build container, and against the frozen files tests/golden/ref_rms_checkpoint_*.pt everywhere).
"""
from __future__ import annotations

import numpy as np
import torch


class _RunningMeanStdFile(torch.nn.Module):
    def __init__(self, mean: torch.Tensor, var: torch.Tensor, count: torch.Tensor):
        super().__init__()
        self.mean = torch.nn.Parameter(mean, requires_grad=False)
        self.m_Var = torch.nn.Parameter(var, requires_grad=False)
        self.m_Count = torch.nn.Parameter(count, requires_grad=False)

    def forward(self) -> torch.Tensor:
        return self.mean


def save_running_mean_std(path: str, mean, var, count: float) -> None:
    """mean / var: arrays of the feature shape ([] for the scalar return statistics, vec_normalize.h:57); float32."""
    m = torch.as_tensor(np.asarray(mean, dtype=np.float32)).clone()
    v = torch.as_tensor(np.asarray(var, dtype=np.float32)).clone()
    c = torch.tensor(float(count), dtype=torch.float32)
    torch.jit.script(_RunningMeanStdFile(m, v, c)).save(str(path))


def load_running_mean_std(path: str):
    """-> (mean ndarray, var ndarray, count float) from a file written by the reference or by save_running_mean_std."""
    mod = torch.jit.load(str(path), map_location="cpu")
    t = dict(mod.named_parameters())
    t.update(dict(mod.named_buffers()))
    missing = [k for k in ("mean", "m_Var", "m_Count") if k not in t]
    if missing:
        raise ValueError(f"{path}: not a RunningMeanStd checkpoint (missing {missing})")
    return (t["mean"].detach().to(torch.float32).numpy().copy(), t["m_Var"].detach().to(torch.float32).numpy().copy(),
            float(t["m_Count"].detach()))

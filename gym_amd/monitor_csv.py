"""Monitor's CSV log (wrappers/monitor.h:20-35,44-50,93-94) for batched envs.

The reference gives every env its own Monitor and, when a file name is passed, its own `<name>.csv`:

    t_start: <clock ticks>\\nid: <id>\\nt,r,l<first row>\\n<row>\\n...          row = "<t>,<r>,<l>"

Two things to know, both reproduced byte for byte (tests/test_ref_pin.py compares with a file the reference wrote):
the header line `t,r,l` is NOT followed by a newline (monitor.h:28, the first row continues it), and the three values
are floats printed by an ostream at default precision, i.e. `%g` with 6 significant digits (monitor.h:16-18,31).

One file per env does not scale to millions of envs, so `MonitorCsv` logs a chosen subset (default: the first 16) and
reads only the finished episodes' lanes from the device.
"""
from __future__ import annotations

import os
import time

import numpy as np

from .vec_env import VecEnvWrapper


def _g(x) -> str:
    return "%g" % float(np.float32(x))


class ResultWriter:
    """gym::ResultWriter (monitor.h:20-35)."""

    def __init__(self, file_name: str, t_start: int, env_id: str):
        if not file_name:
            raise ValueError("file name must not be empty")                 # assert(!fileName.empty()), monitor.h:25
        root, _ = os.path.splitext(file_name)
        self.path = root + ".csv"                                            # replace_extension(".csv"), monitor.h:26
        self.os = open(self.path, "w", newline="")
        self.os.write(f"t_start: {t_start}\nid: {env_id}\n")
        self.os.write("t,r,l")                                               # no newline: monitor.h:28

    def write(self, r: float, l: float, t: float):
        self.os.write(f"{_g(t)},{_g(r)},{_g(l)}\n")                          # monitor.h:31
        return self

    def close(self):
        self.os.close()


class MonitorCsv(VecEnvWrapper):
    """Writes `<directory>/<prefix><global env index>.csv` for the envs in `env_indices` (local indices of this shard)."""

    def __init__(self, venv, directory: str, env_indices=None, prefix: str = "env", flush_every: int = 64):
        super().__init__(venv)
        os.makedirs(directory, exist_ok=True)
        base = venv
        while hasattr(base, "venv"):
            base = base.venv
        self._base = base
        self._idx = np.array(sorted(range(min(16, venv.nEnvs())) if env_indices is None else env_indices), dtype=np.int64)
        if self._idx.size and (self._idx.min() < 0 or self._idx.max() >= venv.nEnvs()):
            raise IndexError("env index out of range")
        self._t0 = time.perf_counter()
        t_start = time.time_ns()
        off = getattr(base, "env_offset", 0)
        self.writers = {int(i): ResultWriter(os.path.join(directory, f"{prefix}{off + int(i)}"), t_start, f"{prefix}{off + int(i)}") for i in self._idx}
        self._dev_idx = None
        self._flush_every, self._since_flush = flush_every, 0

    def stepWait(self):
        import torch
        ret = self.venv.stepWait()
        eng = self._base.engine
        if self._dev_idx is None:
            self._dev_idx = torch.from_numpy(self._idx).to(eng.device)
        done = eng.done[self._dev_idx]
        if bool(done.any()):                                                 # one small D2H only when a logged env finished
            length, _, ep_ret = eng.episode_info()
            d = done.cpu().numpy() != 0
            ls, rs = length[self._dev_idx].cpu().numpy(), ep_ret[self._dev_idx].cpu().numpy()
            t = time.perf_counter() - self._t0
            for k in np.nonzero(d)[0]:
                self.writers[int(self._idx[k])].write(rs[k], ls[k], t)
        self._since_flush += 1
        if self._since_flush >= self._flush_every:
            for w in self.writers.values():
                w.os.flush()
            self._since_flush = 0
        return ret

    def close(self):
        for w in self.writers.values():
            w.close()

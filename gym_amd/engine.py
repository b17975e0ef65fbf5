"""Thin Python owner of a gymhip engine handle.

torch is plumbing here: it owns the device workspace (so results are zero-copy torch tensors on the caller's
stream) and provides torch.distributed for the one collective.  All compute runs in libgymhip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L


@dataclass
class EngineConfig:
    env_kind: int = L.ENV_CARTPOLE
    n_envs: int = 1
    precision: int = L.PRECISION_LOCKSTEP_F64
    reset_mode: int = L.RESET_TAPE
    wrappers: int = 0
    max_episode_steps: int = 0
    env_offset: int = 0
    seed: int = 0
    clip_obs: float = 10.0      # vec_normalize.h:24-27
    clip_reward: float = 10.0
    gamma: float = 0.99
    epsilon: float = 1e-8

    def to_c(self) -> L.Config:
        c = L.Config()
        c.struct_size = C.sizeof(L.Config)
        c.env_kind, c.precision, c.reset_mode = self.env_kind, self.precision, self.reset_mode
        c.wrappers, c.max_episode_steps = self.wrappers, self.max_episode_steps
        c.n_envs, c.env_offset, c.seed = self.n_envs, self.env_offset, self.seed
        c.clip_obs, c.clip_reward, c.gamma, c.epsilon = self.clip_obs, self.clip_reward, self.gamma, self.epsilon
        return c


_ACT_DTYPES = {torch.int32: L.ACT_I32, torch.int64: L.ACT_I64, torch.uint8: L.ACT_U8}


class Engine:
    """Owns one gymhip_engine on `device`; exposes its device buffers as torch tensors (views of the workspace)."""

    def __init__(self, cfg: EngineConfig, device: int | torch.device = 0):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.GymhipError("no GPU visible: the gymhip engine has no CPU fallback")
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        self.cfg = cfg
        ccfg = cfg.to_c()
        nbytes = self.lib.gymhip_workspace_bytes(C.byref(ccfg))
        if nbytes == 0:
            raise L.GymhipError("invalid engine configuration")
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        self._ws_off = (-base) % 256
        h = C.c_void_p()
        L.check(self.lib.gymhip_create(C.byref(ccfg), self.device.index, C.c_void_p(base + self._ws_off), nbytes, C.byref(h)), "gymhip_create")
        self.h = h
        b = L.Buffers()
        L.check(self.lib.gymhip_get_buffers(self.h, C.byref(b)), "gymhip_get_buffers")
        self.F = b.obs_dim
        self.moments_len = b.moments_len
        n, F = cfg.n_envs, self.F
        self._b = b
        self.obs = self._view(b.obs, torch.float32, (n, F))
        self.obs_norm = self._view(b.obs_norm, torch.float32, (n, F))
        self.reward = self._view(b.reward, torch.float32, (n,))
        self.done = self._view(b.done, torch.float32, (n,))
        self.counter = self._view(b.counter, torch.int32, (n,))   # packed words (gymhip.h); decode with counters()
        self._dec = None
        self.returns = self._view(b.returns, torch.float32, (n,))
        self.norm_stats = self._view(b.norm_stats, torch.float32, (2 * F + 4,))
        self.moments = self._view(b.moments, torch.float64, (self.moments_len,))
        self.episode_totals = self._view(b.episode_totals, torch.float64, (3,))
        # MT19937 engines: env i's std::mt19937 state array (u32 [N][624], untempered words; int32 view of the same bits)
        self.mt_state = self._view(b.mt_state, torch.int32, (n, 624))
        # LOCKSTEP: int32 residual rows of the split-double state (gymhip.h); state_f64() joins them with obs
        self.state_residual = self._view(b.state, torch.int32, (n, F)) if cfg.precision == L.PRECISION_LOCKSTEP_F64 else None

    def state_f64(self):
        """The envs' state as the reference holds it (double m_State rows [N, F]); exact in LOCKSTEP mode."""
        out = torch.empty((self.cfg.n_envs, self.F), dtype=torch.float64, device=self.device)
        L.check(self.lib.gymhip_read_state_f64(self.h, C.c_void_p(out.data_ptr()), self._stream()), "gymhip_read_state_f64")
        return out

    def _view(self, ptr, dtype, shape):
        if not ptr:
            return None
        off = ptr - self.workspace.data_ptr()
        numel = int(np.prod(shape))
        esz = torch.empty((), dtype=dtype).element_size()
        return self.workspace[off:off + numel * esz].view(dtype).view(*shape)

    def _stream(self):
        # the caller's current stream as a raw hipStream_t; the private fast path saves ~1 us per call at launch-bound sizes
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return C.c_void_p(raw(self.device.index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gymhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reset tape -----------------------------------------------------------------------------------
    def upload_reset_tape(self, tape: np.ndarray):
        tape = np.ascontiguousarray(tape, dtype=np.float32)
        assert tape.ndim == 3 and tape.shape[1:] == (self.cfg.n_envs, 4), tape.shape
        L.check(self.lib.gymhip_upload_reset_tape(self.h, tape.ctypes.data_as(C.c_void_p), tape.shape[0]), "gymhip_upload_reset_tape")

    def seed_reset_tape(self, seed: int, episodes: int):
        """VecEnv::seed for TAPE mode: the library draws the tape with std::mt19937(seed + env_offset + i) on the host."""
        L.check(self.lib.gymhip_seed_reset_tape(self.h, seed, episodes), "gymhip_seed_reset_tape")

    def seed_reset_tape_unseeded(self, episodes: int):
        L.check(self.lib.gymhip_unseeded_reset_tape(self.h, episodes), "gymhip_unseeded_reset_tape")

    def tape_cursor_max(self) -> int:
        """Rows of the reset tape consumed so far by the env that consumed most (synchronises the stream)."""
        v = C.c_int32(0)
        L.check(self.lib.gymhip_tape_cursor_max(self.h, C.byref(v), self._stream()), "gymhip_tape_cursor_max")
        return v.value

    def extend_reset_tape(self, episodes: int):
        """More rows of the same host-drawn std::mt19937 streams; cursors are kept."""
        L.check(self.lib.gymhip_extend_reset_tape(self.h, episodes), "gymhip_extend_reset_tape")

    def generate_reset_tape(self, episodes: int):
        L.check(self.lib.gymhip_generate_reset_tape(self.h, episodes, self._stream()), "gymhip_generate_reset_tape")

    def download_reset_tape(self, episodes: int) -> np.ndarray:
        out = np.empty((episodes, self.cfg.n_envs, 4), np.float32)
        L.check(self.lib.gymhip_download_reset_tape(self.h, out.ctypes.data_as(C.c_void_p), episodes), "gymhip_download_reset_tape")
        return out

    def seed(self, seed: int):
        """VecEnv::seed: MT19937 engines re-seed every env's device-resident std::mt19937 (seed + env_offset + i); COUNTER_RNG
        engines re-key their draw.  Synchronises the device."""
        L.check(self.lib.gymhip_seed(self.h, seed), "gymhip_seed")

    # ---- stepping -------------------------------------------------------------------------------------
    def _act(self, actions: torch.Tensor):
        if actions.device != self.device or actions.dtype not in _ACT_DTYPES or not actions.is_contiguous() or actions.numel() != self.cfg.n_envs:
            # TensorAdapter::decode semantics (tensor_adapter.cpp:22-23): any tensor is narrowed to int32
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous().view(-1)
            if actions.numel() != self.cfg.n_envs:
                raise ValueError(f"expected {self.cfg.n_envs} actions, got {actions.numel()}")
        return actions

    def reset(self):
        L.check(self.lib.gymhip_reset(self.h, self._stream()), "gymhip_reset")

    def step(self, actions: torch.Tensor):
        a = self._act(actions)
        L.check(self.lib.gymhip_step(self.h, C.c_void_p(a.data_ptr()), _ACT_DTYPES[a.dtype], self._stream()), "gymhip_step")

    def reset_begin(self):
        L.check(self.lib.gymhip_reset_begin(self.h, self._stream()), "gymhip_reset_begin")

    def reset_finish(self):
        L.check(self.lib.gymhip_reset_finish(self.h, self._stream()), "gymhip_reset_finish")

    def step_begin(self, actions: torch.Tensor):
        a = self._act(actions)
        self._keep = a
        L.check(self.lib.gymhip_step_begin(self.h, C.c_void_p(a.data_ptr()), _ACT_DTYPES[a.dtype], self._stream()), "gymhip_step_begin")

    def set_exchange_form(self, form: int):
        """What `moments` holds between *_begin and *_finish: L.EXCHANGE_MOMENTS (2F+6 doubles after a reduce launch) or
        L.EXCHANGE_ROWS (the per-block partial rows themselves; no reduce launch).  Re-points `self.moments`."""
        L.check(self.lib.gymhip_set_exchange_form(self.h, form), "gymhip_set_exchange_form")
        b = L.Buffers()
        L.check(self.lib.gymhip_get_buffers(self.h, C.byref(b)), "gymhip_get_buffers")
        self.moments_len = b.moments_len
        self.moments = self._view(b.moments, torch.float64, (self.moments_len,))

    def step_finish(self):
        L.check(self.lib.gymhip_step_finish(self.h, self._stream()), "gymhip_step_finish")

    def rollout(self, action_tape: torch.Tensor, rec_done=None, rec_reward=None, rec_truncated=None, rec_obs=None):
        assert action_tape.dtype == torch.uint8 and action_tape.is_contiguous() and action_tape.device == self.device
        k = action_tape.shape[0]
        assert action_tape.numel() == k * self.cfg.n_envs
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        L.check(self.lib.gymhip_rollout(self.h, ptr(action_tape), k, ptr(rec_done), ptr(rec_reward), ptr(rec_truncated), ptr(rec_obs), self._stream()), "gymhip_rollout")

    def counters(self):
        """Decode the packed counter words into device lanes: dict(elapsed i32, length i32, truncated u8, ret f32,
        cursor i32), each [N].  length/truncated/ret describe the episode the LAST step finished (valid where
        done != 0, zero elsewhere): info["episode"].{l,r} and info["TimeLimit.truncated"]."""
        n = self.cfg.n_envs
        if self._dec is None:
            mk = lambda dt: torch.empty(n, dtype=dt, device=self.device)
            self._dec = dict(elapsed=mk(torch.int32), length=mk(torch.int32), truncated=mk(torch.uint8), ret=mk(torch.float32), cursor=mk(torch.int32))
        d = self._dec
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.gymhip_decode_counters(self.h, p(d["elapsed"]), p(d["length"]), p(d["truncated"]), p(d["ret"]), p(d["cursor"]), self._stream()),
                "gymhip_decode_counters")
        return d

    def episode_info(self):
        d = self.counters()
        return d["length"], d["truncated"], d["ret"]

    # ---- statistics -----------------------------------------------------------------------------------
    def episode_stats(self) -> np.ndarray:
        out = np.zeros(3, np.float64)
        L.check(self.lib.gymhip_episode_stats(self.h, out.ctypes.data_as(C.c_void_p), self._stream()), "gymhip_episode_stats")
        return out

    def launch_info(self) -> dict:
        """The launch shape and cache policy gymhip_create chose (gymhip_launch_info): speed only, never results."""
        out = (C.c_int32 * 4)()
        L.check(self.lib.gymhip_launch_info(self.h, out), "gymhip_launch_info")
        return {"grid": out[0], "lanes": out[1], "streams_non_temporally": bool(out[2]), "kept_lanes": out[3]}

    def poll_error(self) -> int:
        v = C.c_int32(0)
        L.check(self.lib.gymhip_poll_error(self.h, C.byref(v), self._stream()), "gymhip_poll_error")
        return v.value

    def reset_index(self, index: int):
        """VecEnv::reset(int index) (+ VecNormalize::reset(int index)): results are row `index` of obs / obs_norm."""
        L.check(self.lib.gymhip_reset_index(self.h, int(index), self._stream()), "gymhip_reset_index")

    def update_norm_stats(self, obs_rows: torch.Tensor | None = None, reward_rows: torch.Tensor | None = None):
        """Updateable::update (vec_normalize.h:191-206): push rollout rows into the running statistics."""
        def prep(t, width):
            if t is None:
                return None, 0
            t = t.to(self.device, torch.float32).reshape(-1, width).contiguous()
            return t, t.shape[0]
        o, no = prep(obs_rows, self.F)
        r, nr = prep(reward_rows, 1)
        L.check(self.lib.gymhip_update_norm_stats(self.h, C.c_void_p(o.data_ptr()) if o is not None else None, no,
                                                  C.c_void_p(r.data_ptr()) if r is not None else None, nr, self._stream()),
                "gymhip_update_norm_stats")

    def get_norm_stats(self) -> np.ndarray:
        out = np.zeros(2 * self.F + 4, np.float32)
        L.check(self.lib.gymhip_get_norm_stats(self.h, out.ctypes.data_as(C.c_void_p), self._stream()), "gymhip_get_norm_stats")
        return out

    def set_norm_stats(self, stats: np.ndarray):
        stats = np.ascontiguousarray(stats, np.float32)
        assert stats.size == 2 * self.F + 4
        L.check(self.lib.gymhip_set_norm_stats(self.h, stats.ctypes.data_as(C.c_void_p), self._stream()), "gymhip_set_norm_stats")

"""ctypes binding of oracle/libgym_oracle.so — the CPU checker.  Test infrastructure only (see oracle/gym_oracle.cpp)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgym_oracle.so")

CARTPOLE, MOUNTAINCAR, SCRIPTED = 0, 1, 2
F_TIME_LIMIT, F_CLIP_REWARD, F_MONITOR = 1, 2, 4

_vp, _i64, _u64, _f = C.c_void_p, C.c_int64, C.c_uint64, C.c_float


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_oracle():
    """Compile the oracle if its .so is missing (plain g++; seconds)."""
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "gym_oracle.cpp")):
        subprocess.run(["make", "-C", ORACLE_DIR, "libgym_oracle.so"], check=True, capture_output=True)
    return ORACLE_SO


def default_threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        l = self.lib
        l.oracle_cartpole_reset_tape.argtypes = [_u64, _i64, _i64, _i64, _vp, C.c_int]
        l.oracle_mt19937_raw.argtypes = [_u64, _i64, _vp]
        l.oracle_glibc_sincos_f32.argtypes = [_vp, _i64, _vp, _vp, C.c_int]
        l.oracle_pole_sincos.argtypes = [_vp, _i64, _vp, _vp, C.c_int]
        l.oracle_sincos_mismatch_sweep.argtypes = [C.c_double, C.c_double, _i64, _i64, _i64, _vp, C.c_int]
        l.oracle_vec_create.restype = _vp
        l.oracle_vec_create.argtypes = [C.c_int, _i64, C.c_int, _i64, C.c_int, _vp, _vp, _i64]
        l.oracle_vec_destroy.argtypes = [_vp]
        l.oracle_atari_chain_create.restype = _vp
        l.oracle_atari_chain_set_max_as_written.argtypes = [_vp, C.c_int]
        l.oracle_atari_chain_create.argtypes = [_i64, _u64, _u64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
        l.oracle_atari_chain_destroy.argtypes = [_vp]
        l.oracle_atari_chain_reset.argtypes = [_vp, _vp]
        l.oracle_atari_chain_reset_index.argtypes = [_vp, _i64, _vp]
        l.oracle_atari_chain_step.argtypes = [_vp] * 9
        l.oracle_vec_obs_dim.argtypes = [_vp]
        l.oracle_vec_seed.argtypes = [_vp, _u64]
        l.oracle_vec_set_reset_tape.argtypes = [_vp, _vp, _i64]
        l.oracle_vec_reset.argtypes = [_vp, _vp, _vp]
        l.oracle_vec_set_auto_reset.argtypes = [_vp, C.c_int]
        l.oracle_vec_reset_index.argtypes = [_vp, _i64, _vp, _vp]
        l.oracle_vec_step.argtypes = [_vp] + [_vp] * 8
        l.oracle_vec_rollout.restype = C.c_double
        l.oracle_vec_rollout.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
        l.oracle_vecnorm_attach.argtypes = [_vp, _f, _f, _f, _f]
        l.oracle_vecnorm_set_flags.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
        l.oracle_vecnorm_set_stats.argtypes = [_vp, _vp]
        l.oracle_vecnorm_reset.argtypes = [_vp, _vp, _vp]
        l.oracle_vecnorm_step.argtypes = [_vp] + [_vp] * 8
        l.oracle_vecnorm_stats.argtypes = [_vp, _vp]
        l.oracle_vecnorm_returns.argtypes = [_vp, _vp]
        l.oracle_rms_init.argtypes = [C.c_int, _vp]
        l.oracle_rms_batch_moments.argtypes = [_vp, _i64, C.c_int, _vp, _vp]
        l.oracle_rms_update_from_moments.argtypes = [C.c_int, _vp, _vp, _vp, _i64]
        l.oracle_host_env_post_step.argtypes = [_i64] + [_vp] * 7 + [C.c_int] + [_vp] * 5
        l.oracle_atari_preprocess.argtypes = [_vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int]
        l.oracle_gray.argtypes = [_vp, _i64, C.c_int, _vp]
        l.oracle_area_resize.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int]
        l.oracle_area_table.restype = C.c_int
        l.oracle_area_table.argtypes = [C.c_int, C.c_int, _vp, _vp, _vp, C.c_int]
        l.oracle_framestack_reset.argtypes = [_vp, _vp, _i64, C.c_int, C.c_int]
        l.oracle_framestack_update.argtypes = [_vp, _vp, _i64, C.c_int, C.c_int]
        l.oracle_framestack_update_as_written.argtypes = [_vp, _vp, _i64, C.c_int, C.c_int]
        l.oracle_norm_permute.argtypes = [_vp, _i64, C.c_int, C.c_int, _vp]
        l.oracle_norm_permute_as_written.argtypes = [_vp, _i64, C.c_int, C.c_int, _vp]

    # ---- RNG ------------------------------------------------------------------------------------------
    def cartpole_reset_tape(self, seed, n_envs, episodes, env_offset=0, threads=None):
        out = np.empty((episodes, n_envs, 4), np.float32)
        self.lib.oracle_cartpole_reset_tape(seed, env_offset, n_envs, episodes, _p(out), threads or default_threads())
        return out

    def mt19937_raw(self, seed, count):
        out = np.empty(count, np.uint32)
        self.lib.oracle_mt19937_raw(seed, count, _p(out))
        return out

    # ---- sin/cos of the pole angle (cartpole.cpp:39-40) -------------------------------------------------
    def glibc_sincos_f32(self, theta):
        theta = np.ascontiguousarray(theta, np.float64)
        s, c = np.empty(theta.size, np.float32), np.empty(theta.size, np.float32)
        self.lib.oracle_glibc_sincos_f32(_p(theta), theta.size, _p(s), _p(c), default_threads())
        return s, c

    def pole_sincos(self, theta):
        """Host twin of the HIP kernel's pole_sincos (doubles, before the float narrowing)."""
        theta = np.ascontiguousarray(theta, np.float64)
        s, c = np.empty(theta.size, np.float64), np.empty(theta.size, np.float64)
        self.lib.oracle_pole_sincos(_p(theta), theta.size, _p(s), _p(c), default_threads())
        return s, c

    def sincos_mismatch_sweep(self, lo, hi, count, k0=0, count_here=None):
        out = np.zeros(4, np.float64)
        self.lib.oracle_sincos_mismatch_sweep(lo, hi, count, k0, count if count_here is None else count_here, _p(out), default_threads())
        return dict(sin_mismatches=int(out[0]), cos_mismatches=int(out[1]), sin_max_ulp=float(out[2]), cos_max_ulp=float(out[3]))

    def vec(self, kind, n_envs, flags=0, max_episode_steps=0, threads=None, script=None):
        return OracleVec(self, kind, n_envs, flags, max_episode_steps, threads or default_threads(), script)

    # ---- RunningMeanStd -------------------------------------------------------------------------------
    def rms_init(self, F):
        st = np.empty(2 * F + 1, np.float32)
        self.lib.oracle_rms_init(F, _p(st))
        return st

    def rms_batch_moments(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n, F = x.shape
        bm, bv = np.empty(F, np.float32), np.empty(F, np.float32)
        self.lib.oracle_rms_batch_moments(_p(x), n, F, _p(bm), _p(bv))
        return bm, bv

    def rms_update_from_moments(self, state, bm, bv, n):
        F = bm.size
        self.lib.oracle_rms_update_from_moments(F, _p(state), _p(np.ascontiguousarray(bm, np.float32)), _p(np.ascontiguousarray(bv, np.float32)), n)
        return state

    # ---- Atari ----------------------------------------------------------------------------------------
    def atari_preprocess(self, frames, dh=84, dw=84, gray15=1, threads=None):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, two, H, W, ch = frames.shape
        assert two == 2 and ch == 3
        out = np.empty((n, dh, dw), np.uint8)
        self.lib.oracle_atari_preprocess(_p(frames), n, H, W, dh, dw, gray15, _p(out), threads or default_threads())
        return out

    def gray(self, rgb, gray15=1):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        out = np.empty(rgb.shape[:-1], np.uint8)
        self.lib.oracle_gray(_p(rgb), out.size, gray15, _p(out))
        return out

    def area_resize(self, src, dh, dw):
        src = np.ascontiguousarray(src, np.uint8)
        out = np.empty((dh, dw), np.uint8)
        self.lib.oracle_area_resize(_p(src), src.shape[0], src.shape[1], _p(out), dh, dw)
        return out

    def area_table(self, ssize, dsize):
        cap = 4 * ssize
        si, di, al = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
        k = self.lib.oracle_area_table(ssize, dsize, _p(si), _p(di), _p(al), cap)
        return si[:k].copy(), di[:k].copy(), al[:k].copy()

    def framestack_reset(self, stack, frame):
        n, hw, k = stack.shape[0], frame[0].size, stack.shape[-1]
        self.lib.oracle_framestack_reset(_p(stack), _p(np.ascontiguousarray(frame)), n, hw, k)

    def framestack_update(self, stack, frame):
        n, hw, k = stack.shape[0], frame[0].size, stack.shape[-1]
        self.lib.oracle_framestack_update(_p(stack), _p(np.ascontiguousarray(frame)), n, hw, k)

    def framestack_update_as_written(self, stack, frame):
        """stacked_observation.h:126-130 as written (channels-last): roll by -1, then channels 1.. take the new frame."""
        n, hw, k = stack.shape[0], frame[0].size, stack.shape[-1]
        self.lib.oracle_framestack_update_as_written(_p(stack), _p(np.ascontiguousarray(frame)), n, hw, k)

    def atari_chain(self, n, seed, game_base, palette, frame_skip=4, noop_max=30, life_loss=True, fire=False, clip=True):
        """The reference's Atari wrapper chain (AtariWrapper::make) over the scripted emulator, n envs, SyncVecEnv auto-reset."""
        return AtariChain(self.lib, n, seed, game_base, np.ascontiguousarray(palette, np.uint32), frame_skip, noop_max, life_loss, fire, clip)

    def norm_permute(self, nhwc):
        n, h, w, k = nhwc.shape
        out = np.empty((n, k, h, w), np.float32)
        self.lib.oracle_norm_permute(_p(np.ascontiguousarray(nhwc)), n, h * w, k, _p(out))
        return out

    def norm_permute_as_written(self, nhwc):
        """vec_atari.cpp:70-81 on the reference's int8 stack: signed bytes / 255."""
        n, h, w, k = nhwc.shape
        out = np.empty((n, k, h, w), np.float32)
        self.lib.oracle_norm_permute_as_written(_p(np.ascontiguousarray(nhwc.view(np.uint8))), n, h * w, k, _p(out))
        return out

    def host_env_post_step(self, rew, done, lives, st, clip):
        n = rew.size
        out = dict(reward=np.empty(n, np.float32), done=np.empty(n, np.uint8), has_ep=np.empty(n, np.uint8),
                   ep_r=np.empty(n, np.float32), ep_l=np.empty(n, np.float32))
        self.lib.oracle_host_env_post_step(n, _p(rew), _p(done), _p(lives), _p(st["lives"]), _p(st["was_real_done"]), _p(st["ret"]),
                                           _p(st["len"]), int(clip), _p(out["reward"]), _p(out["done"]), _p(out["has_ep"]),
                                           _p(out["ep_r"]), _p(out["ep_l"]))
        return out


class OracleVec:
    """SyncVecEnv (+ optional VecNormalize) of the oracle's per-env objects."""

    def __init__(self, o: Oracle, kind, n, flags, max_steps, threads, script):
        self.o, self.lib, self.n = o, o.lib, n
        self._script = script
        if script is not None:
            rew, dn = script
            self._srew = np.ascontiguousarray(rew, np.float32)
            self._sdn = np.ascontiguousarray(dn, np.uint8)
            self.h = self.lib.oracle_vec_create(kind, n, flags, max_steps, threads, _p(self._srew), _p(self._sdn), self._srew.size)
        else:
            self.h = self.lib.oracle_vec_create(kind, n, flags, max_steps, threads, None, None, 0)
        self.F = self.lib.oracle_vec_obs_dim(self.h)
        self.has_norm = False

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_vec_destroy(self.h)
            self.h = None

    def seed(self, s):
        self.lib.oracle_vec_seed(self.h, s)

    def set_reset_tape(self, tape):
        self._tape = np.ascontiguousarray(tape, np.float32)
        self.lib.oracle_vec_set_reset_tape(self.h, _p(self._tape), self._tape.shape[0])

    def reset(self, want64=False):
        obs = np.empty((self.n, self.F), np.float32)
        o64 = np.empty((self.n, self.F), np.float64) if want64 else None
        self.lib.oracle_vec_reset(self.h, _p(obs), _p(o64))
        return (obs, o64) if want64 else obs

    def set_auto_reset(self, on):
        self.lib.oracle_vec_set_auto_reset(self.h, int(on))

    def reset_index(self, index):
        row, row64 = np.empty(self.F, np.float32), np.empty(self.F, np.float64)
        self.lib.oracle_vec_reset_index(self.h, index, _p(row), _p(row64))
        return row, row64

    def step(self, actions, want64=False):
        a = np.ascontiguousarray(actions, np.int32)
        n, F = self.n, self.F
        r = dict(obs=np.empty((n, F), np.float32), reward=np.empty(n, np.float32), done=np.empty(n, np.float32),
                 truncated=np.empty(n, np.uint8), ep_r=np.empty(n, np.float32), ep_l=np.empty(n, np.float32))
        r["obs64"] = np.empty((n, F), np.float64) if want64 else None
        self.lib.oracle_vec_step(self.h, _p(a), _p(r["obs"]), _p(r["obs64"]), _p(r["reward"]), _p(r["done"]), _p(r["truncated"]), _p(r["ep_r"]), _p(r["ep_l"]))
        return r

    def rollout(self, tape, rec_obs=False, rec_reward=True, rec_done=True, rec_trunc=True):
        tape = np.ascontiguousarray(tape, np.uint8)
        T, n, F = tape.shape[0], self.n, self.F
        r = dict(obs=np.empty((T, n, F), np.float32) if rec_obs else None, reward=np.empty((T, n), np.float32) if rec_reward else None,
                 done=np.empty((T, n), np.uint8) if rec_done else None, truncated=np.empty((T, n), np.uint8) if rec_trunc else None,
                 final_obs=np.empty((n, F), np.float32), final_obs64=np.empty((n, F), np.float64))
        r["seconds"] = self.lib.oracle_vec_rollout(self.h, _p(tape), T, _p(r["obs"]), _p(r["reward"]), _p(r["done"]), _p(r["truncated"]),
                                                   _p(r["final_obs"]), _p(r["final_obs64"]))
        return r

    # ---- VecNormalize ---------------------------------------------------------------------------------
    def attach_norm(self, clip_obs=10.0, clip_rew=10.0, gamma=0.99, eps=1e-8):
        self.lib.oracle_vecnorm_attach(self.h, clip_obs, clip_rew, gamma, eps)
        self.has_norm = True

    def norm_flags(self, training=True, norm_reward=True, norm_obs=True):
        self.lib.oracle_vecnorm_set_flags(self.h, int(training), int(norm_reward), int(norm_obs))

    def norm_set_stats(self, stats):
        self.lib.oracle_vecnorm_set_stats(self.h, _p(np.ascontiguousarray(stats, np.float32)))

    def norm_reset(self):
        obs = np.empty((self.n, self.F), np.float32)
        raw = np.empty((self.n, self.F), np.float32)
        self.lib.oracle_vecnorm_reset(self.h, _p(obs), _p(raw))
        return obs, raw

    def norm_step(self, actions):
        a = np.ascontiguousarray(actions, np.int32)
        n, F = self.n, self.F
        r = dict(obs=np.empty((n, F), np.float32), raw_obs=np.empty((n, F), np.float32), reward=np.empty(n, np.float32),
                 done=np.empty(n, np.float32), truncated=np.empty(n, np.uint8), ep_r=np.empty(n, np.float32), ep_l=np.empty(n, np.float32))
        self.lib.oracle_vecnorm_step(self.h, _p(a), _p(r["obs"]), _p(r["raw_obs"]), _p(r["reward"]), _p(r["done"]), _p(r["truncated"]), _p(r["ep_r"]), _p(r["ep_l"]))
        return r

    def norm_stats(self):
        out = np.empty(2 * self.F + 4, np.float32)
        self.lib.oracle_vecnorm_stats(self.h, _p(out))
        return out

    def norm_returns(self):
        out = np.empty(self.n, np.float32)
        self.lib.oracle_vecnorm_returns(self.h, _p(out))
        return out


def splitmix_actions(seed, T, n, n_actions):
    """Fixed action tape u8 [T][n]: splitmix64(seed + t*n + i) % n_actions."""
    idx = np.arange(T * n, dtype=np.uint64) + np.uint64(seed)
    with np.errstate(over="ignore"):
        z = idx + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(n_actions)).astype(np.uint8).reshape(T, n)


class AtariChain:
    def __init__(self, lib, n, seed, game_base, palette, frame_skip, noop_max, life_loss, fire, clip):
        self.lib, self.n = lib, n
        self.h = lib.oracle_atari_chain_create(n, seed, game_base, frame_skip, noop_max, int(life_loss), int(fire), int(clip), _p(palette))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_atari_chain_destroy(self.h)
            self.h = None

    def set_max_as_written(self, on=True):
        """MaxAndSkipEnv's buffers alias AtariEnv's one frame buffer in the reference (atari_env.cpp:11-17): observation = last frame."""
        self.lib.oracle_atari_chain_set_max_as_written(self.h, int(on))

    def reset(self):
        fr = np.empty((self.n, 84, 84), np.uint8)
        self.lib.oracle_atari_chain_reset(self.h, _p(fr))
        return fr

    def reset_index(self, index):
        """SyncVecEnv::reset(int index): env `index` alone runs its outermost reset; returns its 84x84 frame."""
        fr = np.empty((84, 84), np.uint8)
        self.lib.oracle_atari_chain_reset_index(self.h, int(index), _p(fr))
        return fr

    def step(self, actions):
        n = self.n
        out = dict(frame=np.empty((n, 84, 84), np.uint8), reward=np.empty(n, np.float32), done=np.empty(n, np.uint8), has_ep=np.empty(n, np.uint8),
                   ep_r=np.empty(n, np.float32), ep_l=np.empty(n, np.float32), lives=np.empty(n, np.int32))
        a = np.ascontiguousarray(actions, np.int32)
        self.lib.oracle_atari_chain_step(self.h, _p(a), _p(out["frame"]), _p(out["reward"]), _p(out["done"]), _p(out["has_ep"]), _p(out["ep_r"]),
                                         _p(out["ep_l"]), _p(out["lives"]))
        return out

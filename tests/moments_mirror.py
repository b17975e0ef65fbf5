"""Test infrastructure: host mirror of the normaliser's all-reduce payload — numpy, float64/float32 exactly as the kernels.  Not part of the product package: it lets the sharded reduction be checked on CPU with gloo.

The step kernel emits, per GPU, shifted power sums of the observations and discounted returns plus episode counters:

    moments[0]            n            envs on this rank
    moments[1 : 1+F]      S1_f = sum_i (obs_if - K_f)          K = the running mean BEFORE this step (replicated)
    moments[1+F : 1+2F]   S2_f = sum_i (obs_if - K_f)^2
    moments[1+2F]         S1 of the returns lane (shift = running return mean)
    moments[2+2F]         S2 of the returns lane
    moments[3+2F : 6+2F]  episodes finished, sum of their returns, sum of their lengths

All entries are plain sums over envs, so a SUM all-reduce (RCCL, 2F+6 doubles) gives every rank the global payload;
each rank then applies the identical float32 RunningMeanStd::updateFromMoment (running_mean_std.h:46-66).
`local_moments` / `apply_moments` restate in numpy what step_kernel + reduce_partials_kernel / normalize_kernel do, so
the sharded reduction can be validated on CPU (tests/test_dist_gloo.py) with gloo standing in for RCCL.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def moments_len(F: int) -> int:
    return 2 * F + 6


def local_moments(obs: np.ndarray, returns: np.ndarray | None, stats: np.ndarray, ep=(0.0, 0.0, 0.0)) -> np.ndarray:
    """obs f32 [n, F]; returns f32 [n] or None; stats f32 [2F+4] (the running statistics before the update)."""
    n, F = obs.shape
    m = np.zeros(moments_len(F), np.float64)
    m[0] = n
    d = obs.astype(np.float64) - stats[:F].astype(np.float64)
    m[1:1 + F] = d.sum(0)
    m[1 + F:1 + 2 * F] = (d * d).sum(0)
    if returns is not None:
        dr = returns.astype(np.float64) - np.float64(stats[2 * F + 1])
        m[1 + 2 * F], m[2 + 2 * F] = dr.sum(), (dr * dr).sum()
    m[3 + 2 * F:6 + 2 * F] = ep
    return m


def _rms_update(mean, var, count, bm, bv, nf):
    tot = f32(count + nf)
    delta = f32(bm - mean)
    new_mean = f32(mean + f32(f32(delta * nf) / tot))
    m2 = f32(f32(f32(var * count) + f32(bv * nf)) + f32(f32(f32(f32(delta * delta) * count) * nf) / tot))
    return new_mean, f32(m2 / tot), tot


def apply_moments(stats: np.ndarray, moments: np.ndarray, F: int, update_obs=True, update_ret=True) -> np.ndarray:
    """The float32 running-statistics update every rank performs on the all-reduced payload."""
    s = stats.astype(f32).copy()
    n = moments[0]
    nf = f32(n)
    if update_obs:
        tot = None
        for f in range(F):
            m1, m2 = moments[1 + f] / n, moments[1 + F + f] / n
            v = m2 - m1 * m1
            bm, bv = f32(np.float64(s[f]) + m1), f32(max(v, 0.0))
            s[f], s[F + f], tot = _rms_update(s[f], s[F + f], s[2 * F], bm, bv, nf)
        s[2 * F] = tot
    if update_ret:
        m1, m2 = moments[1 + 2 * F] / n, moments[2 + 2 * F] / n
        v = m2 - m1 * m1
        bm, bv = f32(np.float64(s[2 * F + 1]) + m1), f32(max(v, 0.0))
        s[2 * F + 1], s[2 * F + 2], s[2 * F + 3] = _rms_update(s[2 * F + 1], s[2 * F + 2], s[2 * F + 3], bm, bv, nf)
    return s


def initial_stats(F: int) -> np.ndarray:
    """RunningMeanStd ctor (running_mean_std.h:16-20): mean 0, var 1, count 1e-4 for obs and returns."""
    s = np.zeros(2 * F + 4, f32)
    s[F:2 * F] = 1
    s[2 * F] = 1e-4
    s[2 * F + 2] = 1
    s[2 * F + 3] = 1e-4
    return s

"""gymhip_update_norm_stats_noalloc — the one entry point that allocates nothing (include/gymhip.h): called through the C
ABI, its results must be bit-identical to gymhip_update_norm_stats' on a twin engine, on two calls in a row, the second
with a smaller input (stale partial rows of the first call must not leak into the second), and it must reject inputs
beyond its stated limit.  Collected last on purpose (file name): the round's end runs `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(kind, n, F_wrappers):
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    return Engine(EngineConfig(env_kind=kind, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                               wrappers=F_wrappers, max_episode_steps=200), 0)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("kind", ["mountaincar", "cartpole"])
def test_noalloc_update_equals_update_on_two_calls_second_smaller(kind):
    from gym_amd import _lib as L
    lib = L.load()
    env = L.ENV_MOUNTAINCAR if kind == "mountaincar" else L.ENV_CARTPOLE
    F = 2 if kind == "mountaincar" else 4
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    a, b = _engine(env, 512, w), _engine(env, 512, w)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(11)
    prev_count = a.get_norm_stats()[2 * F]
    # first call: 700,001 rows -> the full partial grid (512 rows); second: 3,001 rows -> 6 partial rows; third: one row
    for k, rows in enumerate((700_001, 3_001, 1)):
        obs = (torch.randn((rows, F), device="cuda", generator=g) * 0.3 - 0.5).contiguous()
        rew = (torch.randn((rows,), device="cuda", generator=g) * 3).contiguous() if k != 1 else None
        n_rew = rows if rew is not None else 0
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        L.check(lib.gymhip_update_norm_stats_noalloc(a.h, _ptr(obs), rows, _ptr(rew), n_rew, a._stream()), "noalloc")
        assert torch.cuda.memory_allocated() == before
        L.check(lib.gymhip_update_norm_stats(b.h, _ptr(obs), rows, _ptr(rew), n_rew, b._stream()), "update")
        torch.cuda.synchronize()
        sa, sb = a.get_norm_stats(), b.get_norm_stats()
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)), (k, sa, sb)     # bit-identical statistics
        assert torch.equal(a.norm_stats, b.norm_stats)
        assert sa[2 * F] > prev_count                                                  # the statistics moved
        prev_count = sa[2 * F]
    # the engines keep stepping identically afterwards
    act = torch.zeros(512, dtype=torch.int32, device="cuda")
    a.step(act); b.step(act)
    assert torch.equal(a.obs_norm, b.obs_norm) and torch.equal(a.reward, b.reward)
    assert a.poll_error() == 0 and b.poll_error() == 0


def test_noalloc_update_rejects_sizes_beyond_its_limit_and_bad_state():
    from gym_amd import _lib as L
    lib = L.load()
    w = L.WRAP_TIME_LIMIT | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    e = _engine(L.ENV_MOUNTAINCAR, 64, w)
    e.reset()
    st = e.get_norm_stats()
    tiny = torch.zeros((4, 2), device="cuda")
    rc = lib.gymhip_update_norm_stats_noalloc(e.h, _ptr(tiny), L.NOALLOC_MAX_ROWS + 1, None, 0, e._stream())   # rejected before any launch
    assert rc == 1 and b"at most" in lib.gymhip_last_error()
    rc = lib.gymhip_update_norm_stats_noalloc(e.h, None, 0, _ptr(tiny), L.NOALLOC_MAX_ROWS + 1, e._stream())
    assert rc == 1
    assert lib.gymhip_update_norm_stats_noalloc(e.h, _ptr(tiny), 0, None, 0, e._stream()) == 1                   # non-positive row count
    assert np.array_equal(e.get_norm_stats(), st)                                                                # nothing moved
    plain = _engine(L.ENV_MOUNTAINCAR, 64, L.WRAP_TIME_LIMIT)
    assert lib.gymhip_update_norm_stats_noalloc(plain.h, _ptr(tiny), 4, None, 0, plain._stream()) == 1           # no normaliser

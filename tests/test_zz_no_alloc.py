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
        # hipMemGetInfo sees EVERY device allocation of the process (torch.cuda.memory_allocated() only counts torch's caching
        # allocator, so a hipMalloc inside libgymhip.so would slip past it); no torch allocation happens between the two reads
        free_before = torch.cuda.mem_get_info()[0]
        L.check(lib.gymhip_update_norm_stats_noalloc(a.h, _ptr(obs), rows, _ptr(rew), n_rew, a._stream()), "noalloc")
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
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


def test_mem_get_info_check_can_see_a_library_side_allocation():
    """The allocation check above must be able to fail: a raw hipMalloc made behind torch's back (here through the engine
    constructor's own workspace path, workspace = NULL) moves hipMemGetInfo's free bytes while torch.cuda.memory_allocated() stays put."""
    from gym_amd import _lib as L
    lib = L.load()
    torch.cuda.synchronize()
    free_before, torch_before = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
    cfg = L.Config()
    cfg.struct_size = C.sizeof(L.Config)
    cfg.env_kind, cfg.precision, cfg.reset_mode = L.ENV_MOUNTAINCAR, L.PRECISION_LOCKSTEP_F64, L.RESET_COUNTER_RNG
    cfg.wrappers, cfg.max_episode_steps, cfg.n_envs = L.WRAP_TIME_LIMIT, 200, 4 << 20
    cfg.clip_obs, cfg.clip_reward, cfg.gamma, cfg.epsilon = 10.0, 10.0, 0.99, 1e-8
    h = C.c_void_p()
    L.check(lib.gymhip_create(C.byref(cfg), 0, None, 0, C.byref(h)), "gymhip_create")      # the library hipMallocs > 100 MB itself
    try:
        assert torch.cuda.memory_allocated() == torch_before                                  # torch's counter is blind to it
        assert torch.cuda.mem_get_info()[0] < free_before - (64 << 20)                        # hipMemGetInfo is not
    finally:
        lib.gymhip_destroy(h)


def test_noalloc_update_captured_in_a_hip_graph_replays_like_eager_calls():
    """The use the header names: code under stream capture cannot allocate.  The no-alloc call is recorded into a HIP graph (capture
    fails on any hipMalloc / synchronisation inside it) and replayed three times; the statistics equal three eager calls of
    gymhip_update_norm_stats on a twin engine, bit for bit, and the replays allocate nothing."""
    from gym_amd import _lib as L
    lib = L.load()
    F = 2
    w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    a, b = _engine(L.ENV_MOUNTAINCAR, 512, w), _engine(L.ENV_MOUNTAINCAR, 512, w)
    a.reset(); b.reset()
    g_ = torch.Generator(device="cuda").manual_seed(5)
    rows = 40_001
    batches = [(torch.randn((rows, F), device="cuda", generator=g_) * 0.2 - 0.4, torch.randn((rows,), device="cuda", generator=g_) * 2) for _ in range(4)]
    static_obs, static_rew = batches[0][0].clone(), batches[0][1].clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):                       # warm-up call on the capture stream, mirrored eagerly on the twin
        L.check(lib.gymhip_update_norm_stats_noalloc(a.h, _ptr(static_obs), rows, _ptr(static_rew), rows, C.c_void_p(side.cuda_stream)), "noalloc warm-up")
    side.synchronize()
    L.check(lib.gymhip_update_norm_stats(b.h, _ptr(batches[0][0]), rows, _ptr(batches[0][1]), rows, b._stream()), "update")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        L.check(lib.gymhip_update_norm_stats_noalloc(a.h, _ptr(static_obs), rows, _ptr(static_rew), rows, C.c_void_p(side.cuda_stream)), "noalloc under capture")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for k in range(1, 4):
        static_obs.copy_(batches[k][0]); static_rew.copy_(batches[k][1])
        graph.replay()
        L.check(lib.gymhip_update_norm_stats(b.h, _ptr(batches[k][0]), rows, _ptr(batches[k][1]), rows, b._stream()), "update")
        torch.cuda.synchronize()
        sa, sb = a.get_norm_stats(), b.get_norm_stats()
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)), (k, sa, sb)
    assert torch.cuda.mem_get_info()[0] == free_before
    act = torch.ones(512, dtype=torch.int32, device="cuda")
    a.step(act); b.step(act)
    assert torch.equal(a.obs_norm, b.obs_norm) and torch.equal(a.reward, b.reward)

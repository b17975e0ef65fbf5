"""GPU, two processes (both on cuda:0, gloo standing in for RCCL): the sharded HipVecEnv + VecNormalize path end to end.
Each rank owns half of the env range, steps its shard on the GPU, all-reduces the 12-double moments payload between the
step and normalise kernels (gym_amd.vec_env.allreduce_sum) and must reproduce, on its rows, what the oracle's single
VecNormalize over ALL envs produces; the replicated statistics must be identical on both ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, T, out_dir, exchange="moments"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_amd.vec_env import HipVecEnv, TrainingVecNormRO, shard_range
    from tests.oracle_lib import splitmix_actions
    env = TrainingVecNormRO(HipVecEnv("MountainCar-v0", n, max_episode_steps=200, clip_reward=True, monitor=True, device=0, rank=rank, world=world, exchange=exchange))
    lo, hi = shard_range(n, rank, world)
    assert env.nEnvs() == hi - lo
    obs = env.reset()
    rec = dict(obs=[obs.cpu().numpy().copy()], reward=[], done=[], stats=[np.concatenate([np.ravel(v) for d in env.get() for v in d.values()])])
    acts = splitmix_actions(5, T, n, 3)[:, lo:hi]
    for t in range(T):
        s = env.step(torch.from_numpy(acts[t]).cuda())
        rec["obs"].append(s.observation.cpu().numpy().copy()); rec["reward"].append(s.reward.cpu().numpy()[:, 0].copy())
        rec["done"].append(s.done.cpu().numpy()[:, 0].copy())
        rec["stats"].append(np.concatenate([np.ravel(v) for d in env.get() for v in d.values()]))
    rec["episodes"] = env.venv.episode_statistics()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: np.asarray(v) for k, v in rec.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("exchange", ["moments", "rows"])
def test_two_rank_sharded_vecnormalize_matches_single_process_oracle(tmp_path, oracle, exchange):
    from gym_amd.vec_env import shard_range
    from tests.oracle_lib import MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT, splitmix_actions
    n, T, world = 6000, 230, 2
    mp.spawn(_worker, args=(world, _free_port(), n, T, str(tmp_path), exchange), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    assert np.array_equal(r[0]["stats"], r[1]["stats"])                            # replicated statistics never drift apart
    assert np.array_equal(r[0]["episodes"], r[1]["episodes"])
    full = oracle.vec(MOUNTAINCAR, n, F_TIME_LIMIT | F_CLIP_REWARD | F_MONITOR, 200)
    full.attach_norm()
    o_obs, _ = full.norm_reset()
    acts = splitmix_actions(5, T, n, 3)
    episodes = 0
    for k in range(world):
        lo, hi = shard_range(n, k, world)
        assert np.allclose(r[k]["obs"][0], o_obs[lo:hi], rtol=1e-5, atol=1e-5)
    for t in range(T):
        o = full.norm_step(acts[t])
        st = full.norm_stats()
        got = r[0]["stats"][t + 1]                                                 # mean[2] var[2] count | mean var count
        assert (np.abs(got - st) <= 1e-5 * np.maximum(np.abs(st), 1e-6)).all(), (t, got, st)
        cond = np.maximum(1.0, np.abs(st[:2]) / np.sqrt(st[2:4] + 1e-8))
        for k in range(world):
            lo, hi = shard_range(n, k, world)
            assert np.array_equal(r[k]["done"][t], o["done"][lo:hi]), t
            assert (np.abs(r[k]["obs"][t + 1] - o["obs"][lo:hi]) <= 1e-5 * cond * np.maximum(1.0, np.abs(o["obs"][lo:hi]))).all(), t
            assert (np.abs(r[k]["reward"][t] - o["reward"][lo:hi]) <= 1e-5 * np.maximum(1.0, np.abs(o["reward"][lo:hi]))).all(), t
        episodes += int(o["done"].sum())
    assert r[0]["episodes"][0] == episodes == n                                    # every env truncates once at step 200


def _rccl_worker(rank, port, n, T, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    from gym_amd.vec_env import allreduce_sum
    from tests.oracle_lib import splitmix_actions
    w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
    mk = lambda: Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                                     wrappers=w, max_episode_steps=200), 0)
    a, b = mk(), mk()
    a.reset()
    b.reset_begin(); allreduce_sum(b.moments); b.reset_finish()
    acts = torch.from_numpy(splitmix_actions(5, T, n, 3)).cuda()
    same = True
    for t in range(T):
        a.step(acts[t])
        b.step_begin(acts[t]); allreduce_sum(b.moments); b.step_finish()      # exactly bench.py's / HipVecEnv's N>1 call sequence
        same = same and torch.equal(a.obs_norm, b.obs_norm) and torch.equal(a.reward, b.reward) and np.array_equal(a.get_norm_stats(), b.get_norm_stats())
    tot = torch.from_numpy(b.episode_stats()).cuda()
    allreduce_sum(tot)
    mx = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)                                   # bench.py's max-over-ranks
    np.savez(os.path.join(out_dir, "rccl.npz"), same=same, tot=tot.cpu().numpy(), ref=a.episode_stats(), mx=mx.cpu().numpy(),
             backend=np.array(dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_single_rank_rccl_allreduce_on_the_moments_buffer(tmp_path):
    """The N>1 code path on the real backend: a one-rank RCCL communicator (all a 1-GPU box can host — RCCL refuses two
    ranks on one device) all-reduces the f64 `moments` slice of the workspace in place between *_begin and *_finish, on
    the launch stream; results must equal the unsplit step bit for bit."""
    mp.spawn(_rccl_worker, args=(_free_port(), 50000, 60, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert str(r["backend"]) == "nccl" and bool(r["same"]) and np.array_equal(r["tot"], r["ref"]) and r["mx"][0] == 1.25

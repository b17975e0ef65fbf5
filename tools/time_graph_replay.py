"""Per-step time of gymhip_step at launch-bound sizes: eager calls against one HIP graph holding K recorded steps (the same static
action buffer for every step: a timing probe, not a rollout).  usage: python tools/time_graph_replay.py [n_envs] [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for label, cfg in (("cartpole lockstep, counter-RNG resets", dict(env_kind=L.ENV_CARTPOLE, reset_mode=L.RESET_COUNTER_RNG, wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR,
                                                                  max_episode_steps=500, seed=1)),
                   ("mountaincar + VecNormalize", dict(env_kind=L.ENV_MOUNTAINCAR, reset_mode=L.RESET_TAPE, max_episode_steps=200,
                                                       wrappers=L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING))):
    eng = Engine(EngineConfig(n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, **cfg), 0)
    eng.reset()
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(50):
            eng.step(a)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(10 * K):
            eng.step(a)
        side.synchronize()
        eager = (time.perf_counter() - t0) / (10 * K)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            eng.step(a)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (10 * K)
    print(f"{label}, {n} envs: eager {eager * 1e6:.2f} us/step, graph of {K} steps {graph * 1e6:.2f} us/step", flush=True)
    assert eng.poll_error() == 0

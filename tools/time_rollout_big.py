"""gymhip_rollout at 16.7 M envs: K steps per launch, state in registers (no per-step observation / reward / done traffic): what the
arithmetic alone allows.  usage: time_rollout_big.py [envs] [K] [lockstep|f32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mode = sys.argv[3] if len(sys.argv) > 3 else "lockstep"
eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64 if mode == "lockstep" else L.PRECISION_F32,
                          reset_mode=L.RESET_COUNTER_RNG, wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500, seed=1), 0)
eng.reset()
tape = torch.randint(0, 2, (K, n), dtype=torch.uint8, device="cuda")
for _ in range(2):
    eng.rollout(tape)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    eng.rollout(tape)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"rollout {mode} envs={n} K={K}: {ms:.3f} ms per launch, {ms / K * 1e3:.1f} us per step, {n * K / ms * 1e3:.3e} env-steps/s, error_flag={eng.poll_error()}", flush=True)

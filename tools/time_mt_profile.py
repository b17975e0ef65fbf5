"""Per-step kernel time of the MT19937 engine over a long run (regenerations cluster in time while the envs' cursors are still close):
mean kernel ms per bucket of steps.  usage: python tools/time_mt_profile.py [envs] [steps] [bucket]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 250
mode = {"mt": L.RESET_MT19937, "rng": L.RESET_COUNTER_RNG}[os.environ.get("MODE", "mt")]
eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=mode,
                          wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500), 0)
if mode == L.RESET_MT19937:
    eng.seed(0)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = [torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]
for k in range(50):
    eng.step(acts[k % 4])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps // bucket + 1)]
ev[0].record()
for k in range(steps):
    eng.step(acts[k % 4])
    if (k + 1) % bucket == 0:
        ev[(k + 1) // bucket].record()
torch.cuda.synchronize()
cur = eng.counters()["cursor"]
print("launch", eng.launch_info(), "cursor min/max", int(cur.min()), int(cur.max()), "err", eng.poll_error())
for b in range(steps // bucket):
    print(f"steps {b * bucket:6d}..{(b + 1) * bucket:6d}: {ev[b].elapsed_time(ev[b + 1]) / bucket:.4f} ms/step")

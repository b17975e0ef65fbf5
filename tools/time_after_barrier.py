"""Per-step kernel times of the headline engine right after a rank barrier (1-rank RCCL group): which steps pay for the barrier?
usage: python tools/time_after_barrier.py [nccl|none] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
torch.cuda.set_device(0)
if mode == "nccl":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 1 << 24
eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                          wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500), 0)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = [torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]
tok = torch.zeros(1, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for k in range(8):
        eng.step(acts[k % 4])
    torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize()
    if mode == "nccl":
        dist.all_reduce(tok)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    w0 = time.perf_counter()
    ev[0].record()
    for k in range(steps):
        eng.step(acts[k % 4])
        ev[k + 1].record()
    torch.cuda.synchronize()
    if mode == "nccl":
        dist.all_reduce(tok)
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
    print(f"{mode} rep {rep}: wall {1e3 * (w1 - w0):.3f} ms for {steps} steps; per-step ms: " + " ".join(f"{x:.3f}" for x in ms), flush=True)

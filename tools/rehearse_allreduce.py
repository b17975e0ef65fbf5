"""2-rank rehearsal on ONE GPU (gloo): which form of the moments all-reduce is safe?  Prints progress lines."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(45, exit=True)
import torch, torch.distributed as dist
from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig
rank = int(os.environ["RANK"]); torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("BACKEND", "gloo"))
def log(*a):
    if rank == 0: print(*a, flush=True)
t = torch.ones(12, dtype=torch.float64, device="cuda")
for _ in range(50): dist.all_reduce(t)
torch.cuda.synchronize(); log("A plain cuda f64 x50 ok", float(t[0]))
ws = torch.zeros(4096, dtype=torch.uint8, device="cuda"); v = ws[256:256 + 96].view(torch.float64); v.fill_(1.0)
for _ in range(50): dist.all_reduce(v)
torch.cuda.synchronize(); log("B view-of-uint8 x50 ok", float(v[0]))
w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
n = 1 << 20
eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, reset_mode=L.RESET_COUNTER_RNG, wrappers=w, max_episode_steps=200, env_offset=rank * n), 0)
a = torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda")
eng.reset_begin(); dist.all_reduce(eng.moments); eng.reset_finish(); torch.cuda.synchronize(); log("C reset ok", float(eng.moments[0]))
for k in range(40):
    eng.step_begin(a); dist.all_reduce(eng.moments); eng.step_finish()
    if k % 10 == 0: torch.cuda.synchronize(); log("C step", k, float(eng.moments[0]))
torch.cuda.synchronize(); log("C steps ok", eng.get_norm_stats()[:3])
dist.destroy_process_group()

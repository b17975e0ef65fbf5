"""The sharded config-3 step (BASELINE configs[2]: MountainCar + TimeLimit(200) + ClipReward + normaliser, n envs PER GPU) on ONE GPU
with a 1-rank RCCL communicator: wall time per step of
  unsplit   gymhip_step (no exchange: the single-GPU form, for reference)
  moments   step_begin [step + reduce launch] -> ncclAllReduce(2F+6 f64) -> step_finish [normalise]
  rows      step_begin [step]                 -> ncclAllReduce(partial rows, 32 KB) -> step_finish [normalise folds the rows]
each eager and replayed from one HIP graph that holds the RCCL call.  usage: python tools/time_sharded_config3.py [envs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING
g = torch.Generator(device="cuda").manual_seed(3)
acts = [torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]


def engine(form):
    e = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG, wrappers=w, max_episode_steps=200), 0)
    if form == "rows":
        e.set_exchange_form(L.EXCHANGE_ROWS)
    if form == "unsplit":
        e.reset()
    else:
        e.reset_begin(); dist.all_reduce(e.moments); e.reset_finish()
    return e


def one_step(e, form, a):
    if form == "unsplit":
        e.step(a)
    else:
        e.step_begin(a); dist.all_reduce(e.moments); e.step_finish()


def timed(fn, k):
    for i in range(100):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


out = {}
for form in ("unsplit", "moments", "rows"):
    e = engine(form)
    out[form] = timed(lambda i: one_step(e, form, acts[i % 4]), steps)
    payload = e.moments.numel() * 8 if form != "unsplit" else 0
    print(f"{form:8s} eager : {out[form]:7.2f} us/step   (payload {payload} B)", flush=True)
    if os.environ.get("NO_GRAPH"):
        continue
    try:
        static = acts[0].clone()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            one_step(e, form, static)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            one_step(e, form, static)
        us = timed(lambda i: gr.replay(), steps)
        print(f"{form:8s} graph : {us:7.2f} us/step", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"{form:8s} graph : capture failed: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
    st = e.get_norm_stats()
    print(f"         obs count {st[4]:.0f}", flush=True)
dist.destroy_process_group()

"""How far the float-state throughput mode (GYMHIP_PRECISION_F32) drifts from the lockstep mode (double-exact state), measured
on the GPU at BASELINE configs[1]'s size: CartPole + TimeLimit(500), N = 65,536 envs, T = 10,000 steps, the same host-drawn
reset tape and the same action stream for both engines.  Reported the way SURVEY.md §7 ("hard parts") reports its CPU estimate:
  * envs whose done stream differs from lockstep's at some step within T (after that the env is out of phase for good),
  * of the observation samples of envs still in phase, the fraction beyond 1e-5 relative error (absolute floor 1e-2), by episode age.
usage: python tools/measure_f32_divergence.py [n_envs] [steps] [cartpole|mountaincar]     (MountainCar: + TimeLimit(200), three actions)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd import _lib as L
from gym_amd.engine import Engine, EngineConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cart = (sys.argv[3] if len(sys.argv) > 3 else "cartpole") == "cartpole"
seed = 0
w = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR
mk = lambda prec: Engine(EngineConfig(env_kind=L.ENV_CARTPOLE if cart else L.ENV_MOUNTAINCAR, n_envs=n, precision=prec, reset_mode=L.RESET_TAPE, wrappers=w,
                                      max_episode_steps=500 if cart else 200), 0)
exact, fast = mk(L.PRECISION_LOCKSTEP_F64), mk(L.PRECISION_F32)
episodes = T // 8 + 64
for e in (exact, fast):
    if cart:
        e.seed_reset_tape(seed, episodes)
    e.reset()
assert torch.equal(exact.obs, fast.obs)
g = torch.Generator(device="cuda").manual_seed(1)
in_phase = torch.ones(n, dtype=torch.bool, device="cuda")
age = torch.zeros(n, dtype=torch.int32, device="cuda")          # steps since the episode's reset (lockstep engine)
buckets = [(0, 10), (10, 25), (25, 50), (50, 100), (100, 501)]
bad = torch.zeros(len(buckets), dtype=torch.float64, device="cuda")
tot = torch.zeros(len(buckets), dtype=torch.float64, device="cuda")
episodes_in_phase = 0.0
max_rel = 0.0
for t in range(T):
    a = torch.randint(0, 2 if cart else 3, (n,), dtype=torch.int32, device="cuda", generator=g)
    exact.step(a); fast.step(a)
    age += 1
    same_done = exact.done == fast.done
    newly_lost = in_phase & ~same_done
    in_phase &= same_done
    d = exact.done != 0
    # compare pre-reset information only where both are mid-episode: after a done the obs row is the (identical) reset state
    live = in_phase & ~d
    rel = (fast.obs - exact.obs).abs() / exact.obs.abs().clamp_min(1e-2)
    over = (rel > 1e-5).any(dim=1)
    if t % 50 == 0 and bool(live.any()):
        max_rel = max(max_rel, float(rel[live].max()))
    for k, (lo, hi) in enumerate(buckets):
        m = live & (age > lo) & (age <= hi)
        tot[k] += m.sum()
        bad[k] += (m & over).sum()
    episodes_in_phase += float((d & in_phase).sum())
    age[d] = 0
lost = int((~in_phase).sum())
print(f"{'cartpole' if cart else 'mountaincar'} envs={n} steps={T} episodes_in_phase={episodes_in_phase:.0f}")
print(f"largest relative error seen in phase: {max_rel:.3e}")
print(f"done stream lost by {lost} of {n} envs within {T} steps = {lost / max(episodes_in_phase, 1):.3e} per episode")
samples, beyond = float(tot.sum()), float(bad.sum())
print(f"in-phase mid-episode observation rows: {samples:.0f}; beyond 1e-5 relative (floor 1e-2): {beyond:.0f} = {100 * beyond / samples:.3f} %")
for (lo, hi), b_, t_ in zip(buckets, bad.tolist(), tot.tolist()):
    print(f"  episode age {lo + 1:3d}..{hi:3d}: {100 * b_ / max(t_, 1):7.3f} %  of {t_:.0f} rows")
assert exact.poll_error() == 0 and fast.poll_error() == 0

#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched CartPole VecEnv step on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[4] CartPole leg — 16,777,216 envs per GPU, TimeLimit(500)+Monitor,
fused step + in-kernel auto-reset (reset states from the in-kernel counter RNG, as SURVEY.md §8d prescribes for this
config; the host-drawn reset-tape variant is reported in `extra`), one `VecEnv::step` == one kernel launch, LOCKSTEP
(fp64-state) arithmetic, i.e. the kernel whose streams are bit-exact against the CPU oracle for both reset sources
(tests/test_gpu_classic_lockstep.py).  configs[1] (N=65,536)
is the parity-test case: at 6.5 MB per step it is launch-latency-bound and cannot show a roofline fraction.
Envs shard across GPUs with no data-path collective (weak scaling: per-GPU work fixed).

One JSON line on stdout (rank 0).  `value` counts env-steps of ALL ranks / max-over-ranks wall time, inputs resident
in HBM.  `roofline` prices the step kernel: algorithmic bytes per env-step x envs per launch / mean launch duration
(HIP events on the launch stream).  `cpu_baseline` is the CPU oracle (a port of the reference's per-env-object
SyncVecEnv loop) timed on this box's host cores on a bounded sample.

Launch forms: `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) starts N rank processes ITSELF — the
parent never touches the GPU, relays rank 0's JSON line and exits non-zero if any rank fails or the launch times out —
and `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the driver's form) runs as the rank it is told
to be.  Either way there is one process per GPU and `config.ranks_seen` reports the world size torch.distributed saw.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

# Algorithmic HBM bytes per env-step (52 B, "lockstep doubles the 32 B state term"):
# state read+write (obs rows alias the state; lockstep adds the 4 B/element residual rows), action, reward, done,
# step counter r/w
BYTES_PER_STEP = {
    ("cartpole", "lockstep"): 32 + 32 + 4 + 4 + 4 + 8,        # 84
    ("cartpole", "f32"): 16 + 16 + 4 + 4 + 4 + 8,             # 52
    ("mountaincar", "lockstep"): 16 + 16 + 4 + 4 + 4 + 8,     # 52
    ("mountaincar", "f32"): 8 + 8 + 4 + 4 + 4 + 8,            # 36
}


_BARRIER_TOKEN = {}


def rank_barrier(torch, dist):
    """A barrier over all ranks: an all-reduce of one pre-allocated element + a synchronize.  dist.barrier() does the same job but
    costs ~2 ms per call on the nccl backend (measured with a 1-rank RCCL group: it allocates, and synchronises the whole device
    twice), which is most of a 20-step timed region at 0.2 ms per step; this form costs the collective's own latency."""
    key = dist.get_backend()
    if key not in _BARRIER_TOKEN:
        _BARRIER_TOKEN[key] = torch.zeros(1, dtype=torch.float32, device="cuda" if key == "nccl" else "cpu")
    dist.all_reduce(_BARRIER_TOKEN[key])
    if key == "nccl":
        torch.cuda.synchronize()


def max_over_ranks(torch, dist, dt):
    """MAX of a host scalar over ranks (device tensor for nccl, host tensor otherwise)."""
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


PREWARM_SECONDS = float(os.environ.get("GYM_BENCH_PREWARM_S", "0.3"))   # untimed, before the caller's --warmup: clocks ramp on wall time, not on a step count (0 for counter-collection passes: tools/profile_r02.sh)


def prewarm(torch, eng, actions, max_steps=None, seconds=PREWARM_SECONDS):
    """Untimed pre-warm of fixed WALL time so that the timed region does not depend on the caller's --warmup count for
    ramped clocks: `seconds` of eng.step (at most `max_steps` of them when the engine's reset tape is finite, the rest of
    the time on a plain HBM copy loop).  Returns the number of engine steps taken."""
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds and (max_steps is None or k < max_steps):
        for _ in range(8):
            if max_steps is not None and k >= max_steps:
                break
            eng.step(actions[k % len(actions)])
            k += 1
        torch.cuda.synchronize()
    if time.perf_counter() - t0 < seconds:
        a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                b.copy_(a)
            torch.cuda.synchronize()
        del a, b
    return k


def time_engine(torch, dist, eng, actions, steps, warmup, world, prewarm_max_steps=None):
    """Pre-warm, W untimed + K timed VecEnv steps.  Returns (wall seconds max over ranks, mean kernel ms from HIP events)."""
    prewarm(torch, eng, actions, None if prewarm_max_steps is None else prewarm_max_steps // 2)   # the other half of a finite tape's allowance is for the re-warm below
    for w in range(warmup):
        eng.step(actions[w % len(actions)])
    torch.cuda.synchronize()
    if dist.is_initialized():
        # Two barriers.  The first absorbs the skew between the ranks (and, the first time, the collective library's lazy set-up): the
        # GPU idles while it waits, and on MI355X the ~25 ms after such a gap run up to 20 % slow (measured with a 1-rank RCCL group:
        # 0.25 ms per step decaying back to 0.21 over ~100 steps).  So the engine is re-warmed AFTER it, untimed; the ranks then
        # reach the second barrier together, nobody idles, and the timed region starts on a warm chip.
        rank_barrier(torch, dist)
        rewarm = 160 if prewarm_max_steps is None else max(0, min(160, prewarm_max_steps // 2))
        for w in range(rewarm):
            eng.step(actions[w % len(actions)])
        torch.cuda.synchronize()
        rank_barrier(torch, dist)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        eng.step(actions[k % len(actions)])
        ev[k][1].record()
    torch.cuda.synchronize()
    if dist.is_initialized():
        rank_barrier(torch, dist)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        dt = max_over_ranks(torch, dist, dt)
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    if os.environ.get("GYM_BENCH_DUMP_STEPS"):   # debugging aid: every step's own duration and the gaps between them
        print("per-step ms: " + " ".join(f"{a.elapsed_time(b):.3f}" for a, b in ev), file=sys.stderr, flush=True)
        print("gaps ms: " + " ".join(f"{ev[k][1].elapsed_time(ev[k + 1][0]):.3f}" for k in range(steps - 1)), file=sys.stderr, flush=True)
    return dt, kern_ms


def cpu_model():
    """The host CPU's model name (BASELINE.md's CPU-baseline plan, item 4), or "unknown"."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_target):
    """CPU oracle (port of the reference's per-env-object loop), bounded samples — BASELINE.md's CPU-baseline plan:
    (0) configs[0] exactly: ONE CartPole env, seed 0, TimeLimit(500), action tape a_t = mt19937(1)() & 1, 5e6 steps, 1 core
        (loop shape of test/classic_control.cpp:57-63), its done stream checked against the committed golden prefix;
    (1) the SyncVecEnv-shaped loop (sync_env.h:114-123) at N = 1028 (the reference's own smoke-test size,
        test/classic_control.cpp:91-97) and N = 65,536 (configs[1]) on 1 core, and OpenMP over envs (the ASyncVecEnv shape,
        async_env.h:48-86) on all cores the box gives this job."""
    import numpy as np
    from tests.oracle_lib import CARTPOLE, F_MONITOR, F_TIME_LIMIT, Oracle, default_threads, splitmix_actions
    o = Oracle()
    out = {"nproc_affinity": len(os.sched_getaffinity(0)), "os_cpu_count": os.cpu_count(), "threads_all": default_threads(), "cpu_model": cpu_model()}
    # (0) configs[0]
    T0 = 5_000_000
    tape = (o.mt19937_raw(1, T0) & 1).astype(np.uint8).reshape(T0, 1)
    ov = o.vec(CARTPOLE, 1, F_TIME_LIMIT | F_MONITOR, 500, threads=1)
    ov.seed(0)
    ov.reset()
    r = ov.rollout(tape, rec_reward=False, rec_done=True, rec_trunc=False)
    c0 = {"value": T0 / r["seconds"], "unit": "env-steps/s", "cores": 1, "steps": T0, "seconds": round(r["seconds"], 3),
          "episodes": int(r["done"].sum())}
    gpath = os.path.join(ROOT, "tests", "golden", "cartpole_config0_prefix.npz")
    if os.path.exists(gpath):
        g = np.load(gpath, allow_pickle=False)
        k = int(g["steps"])
        c0["golden_prefix_steps"] = k
        c0["golden_prefix_ok"] = bool(np.array_equal(np.packbits(r["done"][:k, 0]), g["done"]))
    out["config0_single_env_5e6_steps"] = c0
    # (1) vec loops
    per = max(0.5, seconds_target / 4)
    for n_envs in (1028, 65536):
        for label, threads in (("1core", 1), ("allcores", default_threads())):
            ov = o.vec(CARTPOLE, n_envs, F_TIME_LIMIT | F_MONITOR, 500, threads=threads)
            ov.seed(0)
            ov.reset()
            T = max(8, min(250, (1 << 22) // n_envs))
            acts = splitmix_actions(99, T, n_envs, 2)
            done_steps, secs = 0, 0.0
            while secs < per:
                rr = ov.rollout(acts, rec_reward=False, rec_done=False, rec_trunc=False)
                secs += rr["seconds"]
                done_steps += T * n_envs
            out[f"vec_{n_envs}_{label}"] = {"value": done_steps / secs, "unit": "env-steps/s", "cores": threads, "env_steps": done_steps,
                                            "seconds": round(secs, 2)}
    return out



def gather_rows(torch, dist, row):
    """all_gather of a small 1-D host tensor over ranks -> [world, len] host tensor (staged on the device for nccl)."""
    world = dist.get_world_size()
    t = row.cuda() if dist.get_backend() == "nccl" else row
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack([o.cpu() for o in out])


def verify_replicated(torch, dist, reduced_count, local_count, stats_f32):
    """What a sharded normaliser step must leave behind: the all-reduced batch count is the SUM of the ranks'
    local env counts, and the replicated running statistics are bit-identical on every rank (every rank applied the same
    float32 RunningMeanStd::updateFromMoment to the same all-reduced moments, running_mean_std.h:46-66).  Collective: every
    rank calls it; every rank gets the same verdict.  Returns (ok, reason)."""
    import numpy as np
    bits = torch.from_numpy(np.ascontiguousarray(stats_f32, np.float32).view(np.int32).astype(np.int64))
    row = torch.cat([torch.tensor([int(reduced_count), int(local_count)], dtype=torch.int64), bits])
    rows = gather_rows(torch, dist, row)
    total = int(rows[:, 1].sum())
    for r in range(rows.shape[0]):
        if int(rows[r, 0]) != total:
            return False, f"rank {r}: all-reduced batch count {int(rows[r, 0])} != sum of local env counts {total}"
        if not torch.equal(rows[r, 2:], rows[0, 2:]):
            k = int((rows[r, 2:] != rows[0, 2:]).nonzero()[0])
            return False, f"rank {r}: running statistic {k} differs from rank 0's (bits {int(rows[r, 2 + k]):#x} vs {int(rows[0, 2 + k]):#x})"
    return True, f"{rows.shape[0]} ranks, batch count {total}, {rows.shape[1] - 2} statistics bit-identical"


# ----------------------------------------------------------------------------------------------------------------------
# Self-launch: `bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself.
# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Parent of the N rank processes.  It never imports torch and never touches the GPU (the ranks are fresh children, not
    re-execs of a process that has initialised HIP).  Rank 0's stdout is relayed; the other ranks' stdout goes to stderr.
    Exit code: 0 only if every rank exited 0; a failing rank's code otherwise; 4 if the launch timed out."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, lines = [], [[] for _ in range(n)]

    def pump(rank, pipe):
        for raw in pipe:
            line = raw.rstrip("\n")
            lines[rank].append(line)
            if rank == 0 and not args.launch_dry_run and line.startswith("{"):   # the ONE JSON line; library banners go to stderr
                print(line, flush=True)
            else:
                print(f"[rank {rank}] {line}", file=sys.stderr, flush=True)

    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GYM_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE, text=True)
        t = threading.Thread(target=pump, args=(rank, p.stdout), daemon=True)
        t.start()
        procs.append((p, t))
    deadline = time.monotonic() + float(os.environ.get("GYM_BENCH_LAUNCH_TIMEOUT", "1500"))
    rc, failed_at = 0, None
    while True:
        codes = [p.poll() for p, _ in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed_at is None:
            failed_at = time.monotonic()
            rc = bad[0][1] if bad[0][1] > 0 else 1
            print(f"bench.py launcher: rank {bad[0][0]} exited with code {bad[0][1]}", file=sys.stderr, flush=True)
        if all(c is not None for c in codes):
            break
        timed_out = time.monotonic() > deadline
        # a dead rank leaves the others waiting in a collective: give them a few seconds to print, then end exactly the PIDs we started
        if timed_out or (failed_at is not None and time.monotonic() - failed_at > 20):
            if timed_out and rc == 0:
                rc = 4
                print("bench.py launcher: timed out waiting for the ranks", file=sys.stderr, flush=True)
            for p, _ in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + 10
            for p, _ in procs:
                try:
                    p.wait(max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    for _, t in procs:
        t.join(2)
    if args.launch_dry_run and rc == 0:
        ranks = [json.loads(ls[-1]) for ls in lines if ls]
        ok = sorted(r["rank"] for r in ranks) == list(range(n)) and all(r["world"] == n for r in ranks)
        print(json.dumps({"launch_dry_run": True, "n_gpus": n, "ranks_seen": len(ranks), "ok": ok, "ranks": sorted(ranks, key=lambda r: r["rank"])}), flush=True)
        rc = 0 if ok else 5
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=16 * 1024 * 1024)
    ap.add_argument("--env", default="cartpole", choices=["cartpole", "mountaincar"])
    ap.add_argument("--mode", default="lockstep", choices=["lockstep", "f32"])
    ap.add_argument("--reset-mode", default="rng", choices=["tape", "rng", "mt19937"],
                    help="CartPole reset source: in-kernel counter RNG (SURVEY.md §8d config 5), a device-resident reset tape (the form host-drawn "
                         "mt19937 resets take), or per-env std::mt19937 generators on the device (the reference's own streams, for ever)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (f32 / MountainCar) measurements")
    ap.add_argument("--launch-dry-run", action="store_true", help="launcher check: every rank prints its rank/world and exits (no GPU, no torch)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # N > 1 and nobody launched us as a rank: start the N ranks ourselves, BEFORE any torch / GPU call in this process
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.launch_dry_run:
        if os.environ.get("GYM_BENCH_DRY_RUN_FAIL_RANK") == os.environ.get("RANK", "0"):   # launcher test hook: this rank dies
            sys.exit(7)
        if os.environ.get("GYM_BENCH_DRY_RUN_HANG_RANK") == os.environ.get("RANK", "0"):   # launcher test hook: this rank never returns
            time.sleep(3600)
        verified = None
        if os.environ.get("GYM_BENCH_DRY_RUN_VERIFY") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # launcher test hook: the N>1 self-check (verify_replicated, as measure_config3 runs it) over a real gloo group on the CPU,
            # with fabricated per-rank results; GYM_BENCH_DRY_RUN_CORRUPT_RANK makes one rank's statistics / batch count wrong
            import numpy as np
            import torch
            import torch.distributed as dist
            dist.init_process_group("gloo")
            rk, wd = dist.get_rank(), dist.get_world_size()
            stats = np.arange(8, dtype=np.float32) * 0.25 + 1.0
            count = 1000 * wd
            corrupt = os.environ.get("GYM_BENCH_DRY_RUN_CORRUPT_RANK")
            if corrupt is not None and int(corrupt) == rk:
                if os.environ.get("GYM_BENCH_DRY_RUN_CORRUPT_WHAT", "stats") == "stats":
                    stats[3] = np.nextafter(stats[3], np.float32(9))   # one ulp: still a failure
                else:
                    count -= 1000                                         # an all-reduce that lost a rank's contribution
            ok, why = verify_replicated(torch, dist, count, 1000, stats)
            dist.barrier()
            dist.destroy_process_group()
            if not ok:
                print(f"bench.py: rank {rk}: sharded-normaliser verification FAILED: {why}", file=sys.stderr, flush=True)
                sys.exit(6)
            verified = why
        d = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
             "world": int(os.environ.get("WORLD_SIZE", "1")), "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}
        if verified is not None:
            d["verified"] = verified
        print(json.dumps(d), flush=True)
        return
    if os.environ.get("GYM_BENCH_TRACE"):   # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GYM_BENCH_TRACE"]), exit=True)

    # Rank 0's stdout carries exactly ONE line, the JSON record: whatever libraries print on fd 1 (Gloo / RCCL banners, C++ iostreams)
    # is sent to stderr from here on, and the record is written to the saved descriptor.
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from gym_amd import _lib as L
    from gym_amd.engine import Engine, EngineConfig
    from gym_amd.vec_env import allreduce_sum

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running as rank {rank} of {world} (the launcher's world wins)", file=sys.stderr, flush=True)
    backend = "none"
    # GYM_BENCH_FORCE_DIST=1: a 1-GPU rehearsal of the multi-rank code path — the process group is created even for ONE rank, so the
    # barriers, the max-over-ranks reduction, dist.all_reduce on the engine's device `moments` view over RCCL, the exchange
    # verification and the collective timing all run exactly as they will on N GPUs
    dist_on = world > 1 or os.environ.get("GYM_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # Rehearsal knobs for a 1-GPU box only (the real N>1 run is one rank per GPU over RCCL): all ranks on device 0, gloo.
        if os.environ.get("GYM_BENCH_SINGLE_DEVICE") == "1":
            local_rank = 0
        backend = os.environ.get("GYM_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    n = args.envs_per_gpu
    failures = []   # self-checks that did not hold (N > 1 exchange verification): the line still comes out, the exit code is 6
    ranks_seen = dist.get_world_size() if dist_on else 1
    if ranks_seen != world:
        raise RuntimeError(f"torch.distributed reports {ranks_seen} ranks, the environment says {world}")
    PREWARM_TAPE_STEPS = 64   # engine steps a finite reset tape is sized for beyond W + K (the rest of the pre-warm is a copy loop)

    def build(env, mode, reset_mode=None):
        reset_mode = reset_mode or args.reset_mode
        kind = L.ENV_CARTPOLE if env == "cartpole" else L.ENV_MOUNTAINCAR
        wrappers = L.WRAP_TIME_LIMIT | L.WRAP_MONITOR | (L.WRAP_CLIP_REWARD if env == "mountaincar" else 0)
        rm = L.RESET_COUNTER_RNG
        if env == "cartpole" and reset_mode in ("tape", "mt19937"):
            rm = L.RESET_TAPE if reset_mode == "tape" else L.RESET_MT19937
        cfg = EngineConfig(env_kind=kind, n_envs=n, precision=L.PRECISION_LOCKSTEP_F64 if mode == "lockstep" else L.PRECISION_F32,
                           reset_mode=rm, wrappers=wrappers, max_episode_steps=500 if env == "cartpole" else 200, env_offset=rank * n, seed=0)
        eng = Engine(cfg, local_rank)
        if rm == L.RESET_MT19937:
            eng.seed(0)   # VecEnv::seed(0): env i <- std::mt19937(0 + global index), seeded on the device
        if env == "cartpole" and reset_mode == "tape":
            # reset tape rows: every episode lasts >= 8 steps, so (W+K)/8 + 2 rows can never run out
            eng.generate_reset_tape((args.steps + args.warmup + PREWARM_TAPE_STEPS) // 8 + 2)
        eng.reset()
        n_act = 2 if env == "cartpole" else 3
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        acts = [torch.randint(0, n_act, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]
        return eng, acts

    def measure(env, mode, reset_mode=None, steps=None, inspect=None):
        eng, acts = build(env, mode, reset_mode)
        tape_mode = env == "cartpole" and (reset_mode or args.reset_mode) == "tape"
        steps = steps or args.steps
        dt, kern_ms = time_engine(torch, dist, eng, acts, steps, args.warmup, world, PREWARM_TAPE_STEPS if tape_mode else None)
        err = eng.poll_error()
        eps = eng.episode_stats()
        li = eng.launch_info()
        seen = inspect(eng) if inspect else None
        eng.close()
        del eng, acts
        torch.cuda.empty_cache()
        bps = BYTES_PER_STEP[(env, mode)]
        achieved = bps * n / (kern_ms * 1e-3) / 1e9
        # residency plan: the 4-B lanes an engine beyond the Infinity Cache keeps IN that cache never reach HBM while
        # steps run back to back; what HBM itself carries is the rest.  None when the engine does not stream (everything is cached).
        kept = bin(li["kept_lanes"] & 7).count("1") if li["streams_non_temporally"] else None
        return {
            "value": world * n * steps / dt, "ms_per_step": dt / steps * 1e3, "kernel_ms": kern_ms, "timed_region_s": dt, "steps": steps,
            "achieved_gbs": achieved, "bytes_per_env_step": bps, "error_flag": err, "episodes": eps[0],
            "hbm_only_bytes_per_env_step": None if kept is None else bps - 4 * kept, "launch": li, "inspected": seen,
        }

    def measure_cache_swept(steps=40):
        """The headline step when something else sweeps the 256 MiB Infinity Cache between two steps (here a 512 MiB read-modify-write
        with the default cache policy, standing in for a trainer's policy network): the residency plan's lanes are then fetched
        from HBM again.  Only the step kernel is timed (HIP events around it); the sweep is not."""
        eng, acts = build("cartpole", "lockstep", "rng")
        scratch = torch.zeros(128 << 20, dtype=torch.float32, device="cuda")
        for k in range(5):
            scratch.add_(1.0); eng.step(acts[k % len(acts)])
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for k in range(steps):
            scratch.add_(1.0)
            ev[k][0].record(); eng.step(acts[k % len(acts)]); ev[k][1].record()
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        eng.close()
        del scratch, eng, acts
        torch.cuda.empty_cache()
        gbs = BYTES_PER_STEP[("cartpole", "lockstep")] * n / (ms * 1e-3) / 1e9
        return {"kernel_ms": round(ms, 4), "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                "note": "not `value`: a 512 MiB cache sweep runs between the steps; the step kernel alone is timed"}

    def measure_config3(n3, steps, form="moments"):
        """BASELINE configs[2]: MountainCar + TimeLimit(200) + ClipReward + Monitor + obs/return normaliser; with more than one
        rank the per-step batch moments + episode statistics (12 doubles) are all-reduced over RCCL between the step and the
        normalise kernels.  n3 envs PER GPU."""
        w = (L.WRAP_TIME_LIMIT | L.WRAP_CLIP_REWARD | L.WRAP_MONITOR | L.WRAP_NORM_OBS | L.WRAP_NORM_REWARD | L.WRAP_NORM_TRAINING)
        eng = Engine(EngineConfig(env_kind=L.ENV_MOUNTAINCAR, n_envs=n3, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_COUNTER_RNG,
                                  wrappers=w, max_episode_steps=200, env_offset=rank * n3), local_rank)
        if dist_on and form == "rows":
            eng.set_exchange_form(L.EXCHANGE_ROWS)   # no reduce launch: the partial rows themselves are all-reduced (gymhip.h)

        def do_reset():
            if dist_on:
                eng.reset_begin()
                allreduce_sum(eng.moments)
                eng.reset_finish()
            else:
                eng.reset()

        def do_step(a):
            if dist_on:
                eng.step_begin(a)
                allreduce_sum(eng.moments)
                eng.step_finish()
            else:
                eng.step(a)   # single GPU: step kernel + normalise kernel (partial rows folded in its prologue)

        do_reset()
        g = torch.Generator(device="cuda").manual_seed(77 + rank)
        acts = [torch.randint(0, 3, (n3,), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]
        for k in range(50 if n3 <= (1 << 21) else 10):
            do_step(acts[k % 4])
        torch.cuda.synchronize()
        if dist_on:   # barrier, untimed re-warm, barrier: as in time_engine (the idle wait at the first one costs the next ~25 ms their clocks)
            rank_barrier(torch, dist)
            for k in range(200 if n3 <= (1 << 21) else 40):
                do_step(acts[k % 4])
            torch.cuda.synchronize()
            rank_barrier(torch, dist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            do_step(acts[k % 4])
        torch.cuda.synchronize()
        if dist_on:
            rank_barrier(torch, dist)
        dt = time.perf_counter() - t0
        if dist_on:
            dt = max_over_ranks(torch, dist, dt)
        stats = eng.get_norm_stats()
        out = {}
        if dist_on:
            # Self-check of the exchange (no multi-rank RCCL run exists before the 8-GPU box sees this): the all-reduced batch count must be
            # the sum of the ranks' env counts and the replicated statistics bit-identical on every rank.  A failure fails the run.
            ok, why = verify_replicated(torch, dist, float(eng.moments[0 if form == "moments" else 15].item()), n3, stats)
            out["allreduce_verified"], out["allreduce_check"] = ok, why
            if not ok:
                failures.append(f"config 3 at {n3} envs/GPU: {why}")
            # the collective alone: the engine's own payload (2F+6 doubles), back to back on the step's stream
            payload = torch.zeros_like(eng.moments)
            for _ in range(20):
                allreduce_sum(payload)
            torch.cuda.synchronize()
            rank_barrier(torch, dist)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 200
            t0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                allreduce_sum(payload)
            e1.record()
            torch.cuda.synchronize()
            out["collective_us"] = round(max_over_ranks(torch, dist, (time.perf_counter() - t0) / reps * 1e6), 2)
            out["collective_us_gpu_events"] = round(e0.elapsed_time(e1) / reps * 1e3, 2)
        eng.close()
        bps = 16 + 16 + 4 + 8 + 8 + 4 + 4 + (8 + 8 + 4 + 4)   # step kernel 60 B + normalise kernel 24 B
        out.update({"value": world * n3 * steps / dt, "unit": "env-steps/s", "envs_per_gpu": n3, "ms_per_step": round(dt / steps * 1e3, 4),
                    "bytes_per_env_step": bps, "achieved_GBs": round(bps * n3 * steps / dt / 1e9, 1), "frac": round(bps * n3 * steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                    "collective": f"all_reduce({eng.moments_len} x f64)/step, backend {backend}, exchange form {form}" if dist_on else None, "obs_count": float(stats[4])})
        return out

    def measure_config2():
        """BASELINE configs[1]: CartPole N = 65,536, lockstep, tape resets.  (a) one launch per VecEnv::step — 5.5 MB per
        launch, i.e. launch-latency-bound, no roofline fraction to show; (b) the rollout form, K = 500 steps per launch
        from a u8 action tape with the state held in registers (what the parity test of this config uses)."""
        n2, K = 65536, 500
        eng = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n2, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_TAPE,
                                  wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500, env_offset=rank * n2, seed=0), local_rank)
        eng.generate_reset_tape((3 * K + 2000) // 8 + 2)
        eng.reset()
        g = torch.Generator(device="cuda").manual_seed(7 + rank)
        a32 = torch.randint(0, 2, (n2,), dtype=torch.int32, device="cuda", generator=g)
        tape = torch.randint(0, 2, (K, n2), dtype=torch.uint8, device="cuda", generator=g)
        for _ in range(100):
            eng.step(a32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            eng.step(a32)
        torch.cuda.synchronize()
        per_step = (time.perf_counter() - t0) / 1000
        eng.rollout(tape)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            eng.rollout(tape)
        torch.cuda.synchronize()
        roll = (time.perf_counter() - t0) / (2 * K)
        err = eng.poll_error()
        eng.close()
        # the same size with the reference's own reset streams resident on the device (GYMHIP_RESET_MT19937): 12,000 steps, i.e. every env's
        # generator block is regenerated >= 3 times inside the timed region (one regeneration per 156 episodes of an env)
        mt = Engine(EngineConfig(env_kind=L.ENV_CARTPOLE, n_envs=n2, precision=L.PRECISION_LOCKSTEP_F64, reset_mode=L.RESET_MT19937,
                                 wrappers=L.WRAP_TIME_LIMIT | L.WRAP_MONITOR, max_episode_steps=500, env_offset=rank * n2), local_rank)
        mt.seed(0)
        mt.reset()
        acts4 = [torch.randint(0, 2, (n2,), dtype=torch.int32, device="cuda", generator=g) for _ in range(4)]
        for k in range(100):
            mt.step(acts4[k % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(12000):
            mt.step(acts4[k % 4])
        torch.cuda.synchronize()
        mt_step = (time.perf_counter() - t0) / 12000
        cur = mt.counters()["cursor"]
        mt_rows, mt_rows_median = int(cur.min()), int(cur.float().median())
        err |= mt.poll_error()
        mt.close()
        return {"envs": n2, "step_api": {"value": n2 / per_step, "unit": "env-steps/s (per GPU)", "us_per_step": round(per_step * 1e6, 2)},
                "rollout_500_steps_per_launch": {"value": n2 / roll, "unit": "env-steps/s (per GPU)", "us_per_step": round(roll * 1e6, 3)},
                "step_api_device_mt19937_resets": {"value": n2 / mt_step, "unit": "env-steps/s (per GPU)", "us_per_step": round(mt_step * 1e6, 2), "steps": 12000,
                                                   "regenerations_per_env_median": mt_rows_median // 156, "regenerations_per_env_at_least": mt_rows // 156},
                "error_flag": err}

    def measure_rollout_big(K=32):
        """The harness form at the headline size: gymhip_rollout, K steps per launch from a u8 action tape with the (exact, fp64) state in
        registers and no per-step observation / reward / done traffic — what the arithmetic alone allows; NOT the VecEnv::step path."""
        eng, _ = build("cartpole", "lockstep", "rng")
        g = torch.Generator(device="cuda").manual_seed(11 + rank)
        tape = torch.randint(0, 2, (K, n), dtype=torch.uint8, device="cuda", generator=g)
        eng.rollout(tape)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.rollout(tape)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        err = eng.poll_error()
        eng.close()
        del tape
        return {"value": n * K / (ms * 1e-3), "unit": "env-steps/s (per GPU)", "envs": n, "steps_per_launch": K, "us_per_step": round(ms / K * 1e3, 2), "error_flag": err}

    def measure_pcie_inclusive(steps):
        """What the rates become if a caller keeps actions / results / frames on the HOST (never `value`):
        (a) the headline CartPole step with int32 actions from pinned host memory and obs/reward/done copied back to it;
        (b) the Atari step with palette-index screens uploaded from pinned host memory every step."""
        out = {}
        eng, acts = build("cartpole", "lockstep")
        h_act = torch.empty(n, dtype=torch.int32).pin_memory()
        h_act.copy_(acts[0].cpu())
        d_act = torch.empty(n, dtype=torch.int32, device="cuda")
        h_obs, h_rew, h_done = (torch.empty(sh, dtype=torch.float32).pin_memory() for sh in ((n, 4), (n,), (n,)))
        def host_step():
            d_act.copy_(h_act, non_blocking=True)
            eng.step(d_act)
            h_obs.copy_(eng.obs, non_blocking=True); h_rew.copy_(eng.reward, non_blocking=True); h_done.copy_(eng.done, non_blocking=True)
        for _ in range(3):
            host_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            host_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out["cartpole_lockstep_host_actions_and_results"] = {"value": n / dt, "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
                                                             "pcie_bytes_per_env_step": 28, "pcie_GBs": round(28 * n / dt / 1e9, 1)}
        eng.close()
        del eng, acts, d_act
        torch.cuda.empty_cache()
        from gym_amd.atari import AtariPreprocessor
        n4 = 4096
        pp = AtariPreprocessor(n4, device=local_rank)
        pp.set_palette(torch.randint(0, 1 << 24, (256,), dtype=torch.int64, generator=torch.Generator().manual_seed(3)).numpy().astype("uint32"))
        h_scr = (torch.randint(0, 128, (n4, 2, 210, 160), dtype=torch.uint8) * 2).pin_memory()
        d_scr = torch.empty_like(h_scr, device="cuda")
        d_scr.copy_(h_scr); pp.reset(d_scr)
        def host_frames_step():
            d_scr.copy_(h_scr, non_blocking=True)
            pp.step(d_scr)
        for _ in range(3):
            host_frames_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            host_frames_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out["atari_index_screens_from_host"] = {"value": n4 / dt, "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
                                                "pcie_bytes_per_env_step": 67200, "pcie_GBs": round(67200 * n4 / dt / 1e9, 1)}
        pp.close()
        return out

    def measure_atari(n4, steps):
        """BASELINE configs[3]: 4096 envs, MaxAndSkip max + WarpFrame 84x84 gray + FrameStack(4) ring write, synthetic frames."""
        from gym_amd.atari import AtariPreprocessor
        pp = AtariPreprocessor(n4, device=local_rank)
        g = torch.Generator(device="cuda").manual_seed(5 + rank)
        frames = torch.randint(0, 256, (n4, 2, 210, 160, 3), dtype=torch.uint8, device="cuda", generator=g)
        pp.reset(frames)
        for _ in range(5):
            pp.step(frames)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            pp.step(frames)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        out = torch.empty((n4, 4, 84, 84), dtype=torch.float32, device="cuda")
        e0.record()
        for _ in range(steps):
            pp.stacked_nchw_f32(out)
        e1.record()
        torch.cuda.synchronize()
        ms_mat = e0.elapsed_time(e1) / steps
        out16 = torch.empty((n4, 4, 84, 84), dtype=torch.bfloat16, device="cuda")
        pp.stacked_nchw_16(torch.bfloat16, out16)
        e0.record()
        for _ in range(steps):
            pp.stacked_nchw_16(torch.bfloat16, out16)
        e1.record()
        torch.cuda.synchronize()
        ms_mat16 = e0.elapsed_time(e1) / steps
        del out16
        out8 = torch.empty((n4, 84, 84, 4), dtype=torch.uint8, device="cuda")   # the channels-last uint8 view VecFrameStack itself returns
        pp.stacked_nhwc(out8)
        e0.record()
        for _ in range(steps):
            pp.stacked_nhwc(out8)
        e1.record()
        torch.cuda.synchronize()
        ms_nhwc = e0.elapsed_time(e1) / steps
        del out8
        # the same step fed with the emulator's palette-index screens (1 B/pixel) and the in-kernel colour table
        pal = torch.randint(0, 1 << 24, (256,), dtype=torch.int64, generator=torch.Generator().manual_seed(3)).numpy().astype("uint32")
        pp.set_palette(pal)
        del frames
        screens = torch.randint(0, 128, (n4, 2, 210, 160), dtype=torch.uint8, device="cuda", generator=g) * 2
        for _ in range(5):
            pp.step(screens)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            pp.step(screens)
        e1.record()
        torch.cuda.synchronize()
        ms_idx = e0.elapsed_time(e1) / steps
        bps_idx = 2 * 210 * 160 + 84 * 84
        pp.close()
        bps = 2 * 210 * 160 * 3 + 84 * 84
        return {"value": n4 / (ms * 1e-3), "unit": "env-steps/s (per GPU)", "envs": n4, "kernel_ms": round(ms, 4), "bytes_per_env_step": bps,
                "achieved_GBs": round(bps * n4 / (ms * 1e-3) / 1e9, 1), "frac": round(bps * n4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "indexed_screens": {"value": n4 / (ms_idx * 1e-3), "kernel_ms": round(ms_idx, 4), "bytes_per_env_step": bps_idx,
                                    "achieved_GBs": round(bps_idx * n4 / (ms_idx * 1e-3) / 1e9, 1)},
                "materialize_nchw_bf16_ms": round(ms_mat16, 4),
                "materialize_nchw_f32_ms": round(ms_mat, 4),
                "materialize_nhwc_u8_ms": round(ms_nhwc, 4),
                "materialize_GBs": round((4 * 7056 + 4 * 4 * 7056) * n4 / (ms_mat * 1e-3) / 1e9, 1)}

    main_m = measure(args.env, args.mode)

    # PMC-measured HBM bytes per launch of the headline kernel (profiles/traffic.json; valid for the profiled launch size only)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"{args.env}_{args.mode}" + ({"tape": "_tape", "mt19937": "_mt19937"}.get(args.reset_mode, "") if args.env == "cartpole" else ""), {})
            if rec.get("envs_per_launch") == n:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    extra, cpu = {}, None
    printed = [False]

    def emit():
        if rank != 0 or printed[0]:
            return
        printed[0] = True
        line = make_line()
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()

    def make_line():
        return {
            "metric": f"env-steps/sec (whole node), {'CartPole' if args.env == 'cartpole' else 'MountainCar'} vec_env", "value": main_m["value"], "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_m["ms_per_step"],
            "timed_region_s": round(main_m["timed_region_s"], 6), "prewarm_s": PREWARM_SECONDS,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.mode == "lockstep" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.env} vec_env, {n} envs/GPU, TimeLimit+Monitor, fused step+auto-reset ({ {'rng': 'in-kernel counter-RNG', 'tape': 'reset-tape', 'mt19937': 'per-env device-resident std::mt19937'}[args.reset_mode] } resets), {args.mode} arithmetic, "
                                   f"one kernel launch per VecEnv::step (BASELINE configs[4])",
                       "envs_per_gpu": n, "global_envs": n * world, "parallelism": f"env-shard x{world}, no data-path collective",
                       "ranks_seen": ranks_seen, "backend": backend, "launched_by": "bench.py" if os.environ.get("GYM_BENCH_SELF_LAUNCHED") else ("torchrun/env" if world > 1 else ("single process, forced 1-rank process group" if dist_on else "single process")),
                       "error_flag": main_m["error_flag"], "episodes_finished_rank0": main_m["episodes"]},
            "roofline": {"bound": "hbm", "achieved": round(main_m["achieved_gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(main_m["achieved_gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "step_kernel", "kernel_ms": round(main_m["kernel_ms"], 4),
                         "algorithmic_bytes_per_env_step": main_m["bytes_per_env_step"],
                         # `frac` is an EFFECTIVE rate: with steps back to back the residency plan serves the kept 4-B lanes from the
                         # Infinity Cache.  frac_hbm_only prices only what HBM itself carries; frac_cache_swept is the same kernel
                         # timed with a 512 MiB sweep of that cache between every two steps (what a policy network does).
                         "hbm_only_bytes_per_env_step": main_m["hbm_only_bytes_per_env_step"],
                         "frac_hbm_only": None if main_m["hbm_only_bytes_per_env_step"] is None else
                         round(main_m["achieved_gbs"] * main_m["hbm_only_bytes_per_env_step"] / main_m["bytes_per_env_step"] / HBM_PEAK_GBS, 4),
                         "frac_cache_swept": (extra.get("cartpole_lockstep_cache_swept_between_steps") or {}).get("frac"),
                         "kept_lanes": main_m["launch"]["kept_lanes"], "grid": main_m["launch"]["grid"]},
            "cpu_baseline": cpu,
            "extra": extra,
        }

    # The headline is measured; everything below is secondary.  A watchdog on every rank guarantees that the ONE JSON line
    # still comes out (and the job ends) if a secondary measurement stalls, e.g. a rank that failed before a collective.
    import threading

    in_flight = ["(none)"]

    def on_stall():
        # the headline line still comes out, but the run is NOT a success: exit non-zero and say what hung
        extra["error"] = f"secondary measurement {in_flight[0]!r} exceeded the time limit on rank {rank}; headline unaffected"
        emit()
        print(f"bench.py: rank {rank} stalled in {in_flight[0]!r}", file=sys.stderr, flush=True)
        os._exit(3)

    dog = threading.Timer(float(os.environ.get("GYM_BENCH_EXTRA_TIMEOUT", "300" if world == 1 else "150")), on_stall)
    dog.daemon = True
    dog.start()

    spent = extra.setdefault("seconds_per_measurement", {})   # wall-clock of each secondary measurement on rank 0 (sizes the watchdog)

    def guarded(key, fn, *a):
        in_flight[0] = key
        t0 = time.perf_counter()
        try:
            extra[key] = fn(*a)
        except Exception as e:  # noqa: BLE001 — report, never lose the headline line
            extra[key] = {"error": f"{type(e).__name__}: {e}"}
        spent[key] = round(time.perf_counter() - t0, 2)

    def classic(env, mode, reset_mode=None, steps=None, inspect=None):
        m = measure(env, mode, reset_mode, steps, inspect)
        return {**(m["inspected"] or {}), "value": m["value"], "unit": "env-steps/s", "kernel_ms": round(m["kernel_ms"], 4), "steps": m["steps"],
                "bytes_per_env_step": m["bytes_per_env_step"], "achieved_GBs": round(m["achieved_gbs"], 1),
                "frac": round(m["achieved_gbs"] / HBM_PEAK_GBS, 4), "error_flag": m["error_flag"]}

    def mt19937_resets():
        """The reference-exact reset source at scale: per-env std::mt19937 states on the device (GYMHIP_RESET_MT19937).  An env's
        624-word generator block is regenerated once per 156 of its episodes (~3,500 random-policy steps); the timed region is
        20,000 steps, so that every env regenerates its block at least 3 times inside it (reported from the envs' cursors)."""
        def regenerations(eng):
            cur = eng.counters()["cursor"]
            return {"episodes_per_env_at_least": int(cur.min()), "regenerations_per_env_at_least": int(cur.min()) // 156, "regenerations_per_env_median": int(cur.float().median()) // 156}
        out = classic("cartpole", "lockstep", "mt19937", steps=max(args.steps, 20000), inspect=regenerations)
        out["note"] = "reset states = std::mt19937(seed + env index) per env, seeded / twisted / tempered on the GPU; bit-identical to a host-drawn tape (tests/test_gpu_mt19937.py)"
        return out

    if not args.no_extra and args.env == "cartpole" and args.mode == "lockstep":
        other = "tape" if args.reset_mode == "rng" else "rng"
        guarded(f"cartpole_lockstep_{other}_resets", classic, "cartpole", "lockstep", other)
        for env, mode in (("cartpole", "f32"), ("mountaincar", "lockstep"), ("mountaincar", "f32")):
            guarded(f"{env}_{mode}", classic, env, mode)
        guarded(f"cartpole_f32_{other}_resets", classic, "cartpole", "f32", other)
        guarded("cartpole_lockstep_mt19937_resets", mt19937_resets)
        guarded("cartpole_lockstep_cache_swept_between_steps", measure_cache_swept)
        guarded("config2_cartpole_lockstep_65536", measure_config2)
        guarded("cartpole_lockstep_rollout_32_steps_per_launch", measure_rollout_big)
        guarded("config3_mountaincar_vecnormalize_131072_per_gpu", measure_config3, 1 << 17, max(args.steps, 500))   # configs[2]'s per-GPU shard on 8 GPUs
        if dist_on:   # the exchange form without the reduce launch (32 KB payload): measured next to the default on every multi-rank run
            guarded("config3_mountaincar_vecnormalize_131072_per_gpu_rows_exchange", measure_config3, 1 << 17, max(args.steps, 500), "rows")
        guarded("config3_mountaincar_vecnormalize_1M_per_gpu", measure_config3, 1 << 20, max(args.steps, 200))
        guarded("config3_mountaincar_vecnormalize_16M_per_gpu", measure_config3, 1 << 24, max(20, args.steps // 4))
        if rank == 0:
            guarded("config4_atari_preprocess_4096", measure_atari, 4096, max(20, args.steps // 4))
        if rank == 0 and world == 1:
            guarded("pcie_inclusive_never_value", measure_pcie_inclusive, 10)

    dog.cancel()   # the other ranks wait at the barrier below while rank 0 times the CPU baseline
    if rank == 0 and not args.no_cpu_baseline:   # any world: no SCALE line carries a null baseline
        in_flight[0] = "cpu_baseline"
        try:
            c = cpu_baseline(args.cpu_seconds)
            big, c0 = c["vec_65536_allcores"], c["config0_single_env_5e6_steps"]
            cpu = {"value": big["value"], "unit": "env-steps/s", "cores": big["cores"], "kind": "port",
                   "sample": f"CartPole+TimeLimit(500)+Monitor, 65536 envs (configs[1]), {big['env_steps']} env-steps in {big['seconds']}s, OpenMP over envs "
                             f"(ASyncVecEnv shape); 1 core: {c['vec_65536_1core']['value']:.3e}; N=1028: {c['vec_1028_1core']['value']:.3e} (1 core) / "
                             f"{c['vec_1028_allcores']['value']:.3e} ({c['vec_1028_allcores']['cores']} cores); configs[0] single env 5e6 steps, 1 core: "
                             f"{c0['value']:.3e} steps/s (golden prefix ok: {c0.get('golden_prefix_ok')}); sched_getaffinity={c['nproc_affinity']}, "
                             f"os.cpu_count()={c['os_cpu_count']}, cpu={c['cpu_model']}",
                   "detail": c}
        except Exception as e:  # noqa: BLE001
            cpu = {"error": f"{type(e).__name__}: {e}"}

    if failures:
        extra["error"] = "; ".join(failures)
    emit()
    if dist_on:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    if failures:
        print(f"bench.py: rank {rank}: " + "; ".join(failures), file=sys.stderr, flush=True)
        sys.exit(6)


if __name__ == "__main__":
    main()

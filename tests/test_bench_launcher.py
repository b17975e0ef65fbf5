"""bench.py's own launcher (`--gpus N` with no WORLD_SIZE in the environment): N fresh rank processes, one JSON line
relayed, non-zero exit when a rank fails or the launch times out.  CPU only: `--launch-dry-run` ranks print their
rank/world and exit without importing torch or touching a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


@pytest.mark.parametrize("n", [2, 8])
def test_gpus_n_launches_n_ranks(n):
    r = _run(["--gpus", str(n), "--launch-dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1                                                   # ONE line on stdout
    d = json.loads(lines[0])
    assert d["ok"] and d["n_gpus"] == n and d["ranks_seen"] == n
    assert [x["rank"] for x in d["ranks"]] == list(range(n)) == [x["local_rank"] for x in d["ranks"]]
    assert all(x["world"] == n for x in d["ranks"]) and len({x["master"] for x in d["ranks"]}) == 1
    assert d["ranks"][0]["master"].startswith("127.0.0.1:")


def test_single_gpu_and_torchrun_style_env_do_not_relaunch():
    r = _run(["--gpus", "1", "--launch-dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout)["world"] == 1
    # launched as a rank by somebody else (the driver's torchrun form): runs as that rank, starts nothing
    r = _run(["--gpus", "4", "--launch-dry-run"], {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    d = json.loads(r.stdout)
    assert r.returncode == 0 and d == {"rank": 2, "local_rank": 2, "world": 4, "master": "127.0.0.1:29999"}


def test_failing_rank_makes_the_launch_fail():
    r = _run(["--gpus", "4", "--launch-dry-run"], {"GYM_BENCH_DRY_RUN_FAIL_RANK": "2"})
    assert r.returncode == 7 and "rank 2 exited with code 7" in r.stderr


def test_hung_rank_hits_the_launch_timeout():
    r = _run(["--gpus", "2", "--launch-dry-run"], {"GYM_BENCH_DRY_RUN_HANG_RANK": "1", "GYM_BENCH_LAUNCH_TIMEOUT": "2"})
    assert r.returncode == 4 and "timed out" in r.stderr


# ---- the N > 1 self-check (bench.py verify_replicated, what measure_config3 runs after its timed loop) over a real gloo group ----
def test_sharded_verification_passes_on_consistent_ranks():
    r = _run(["--gpus", "2", "--launch-dry-run"], {"GYM_BENCH_DRY_RUN_VERIFY": "1"}, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ok"] and all("2 ranks, batch count 2000" in x["verified"] for x in d["ranks"])


@pytest.mark.parametrize("what", ["stats", "count"])
def test_a_rank_whose_verification_fails_makes_the_launch_fail(what):
    # one rank's replicated statistics differ by one ulp / the all-reduced batch count lost a rank's contribution:
    # every rank sees the same gathered table, reports the failure and exits 6; the launcher relays that code
    r = _run(["--gpus", "2", "--launch-dry-run"], {"GYM_BENCH_DRY_RUN_VERIFY": "1", "GYM_BENCH_DRY_RUN_CORRUPT_RANK": "1",
                                                    "GYM_BENCH_DRY_RUN_CORRUPT_WHAT": what}, timeout=300)
    assert r.returncode == 6, (r.returncode, r.stderr[-2000:])
    assert "verification FAILED" in r.stderr
    assert ("running statistic 3 differs" in r.stderr) if what == "stats" else ("all-reduced batch count" in r.stderr)

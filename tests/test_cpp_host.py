"""The C++ host mirror (gym_amd/include/gym/hip_vec_env.h): builds against libtorch on CPU; on the GPU a driver shaped
like the reference's own smoke test (test/classic_control.cpp:69-99) must reproduce the oracle's streams."""
import os
import subprocess

import numpy as np
import pytest

from tests.oracle_lib import CARTPOLE, MOUNTAINCAR, F_CLIP_REWARD, F_MONITOR, F_TIME_LIMIT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin", "hip_vec_env_demo")
FE_BIN = os.path.join(ROOT, "tests", "cpp", "bin", "atari_frontend_demo")


def test_cpp_mirror_compiles_and_links():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(BIN) and os.path.exists(FE_BIN)


def _run(env, n, steps, norm, ckpt_dir=None):
    r = subprocess.run([BIN, env, str(n), str(steps), str(int(norm))] + ([str(ckpt_dir)] if ckpt_dir else []), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return [l.split() for l in r.stdout.strip().splitlines()]


@pytest.mark.gpu
@pytest.mark.parametrize("env,norm", [("cartpole", False), ("mountaincar", True), ("mountaincar", 2)])   # 2 = composed through the gym::HipTrainingVecNormRO wrapper class, as the reference composes TrainingVecNormRO over a vec env
def test_cpp_training_loop_matches_oracle(oracle, env, norm, tmp_path):
    assert os.path.exists(BIN), "build tests/cpp first (python -c 'import __graft_entry__ as g; g.build()')"
    n, steps = 2048, 260
    cart = env == "cartpole"
    n_act = 2 if cart else 3
    v = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, n, F_TIME_LIMIT | F_MONITOR | (0 if cart else F_CLIP_REWARD), 500 if cart else 200)
    v.seed(7)
    if norm:
        v.attach_norm()
    lines = _run(env, n, steps, norm, tmp_path if norm else None)
    if norm:
        # the C++ mirror's save(): files in the reference's archive format, reloaded by its own load() and readable here
        from gym_amd.checkpoint import load_running_mean_std
        assert lines[-2] == ["ckpt", "1"] and lines[-1] == ["update", "1", "1"]
        lines = lines[:-2]
    obs0 = v.norm_reset()[0] if norm else v.reset()
    assert lines[0][0] == "reset" and np.isclose(float(lines[0][1]), obs0.astype(np.float64).sum(), rtol=1e-6, atol=1e-3)
    idx = np.arange(n, dtype=np.int64)
    csv_rows = {0: [], 3: []}
    tot = np.zeros(3)
    for t in range(steps):
        a = ((idx * 7 + t * 3) % n_act).astype(np.int32)
        r = v.norm_step(a) if norm else v.step(a)
        dd = r["done"] != 0
        tot += [dd.sum(), r["ep_r"][dd].astype(np.float64).sum(), r["ep_l"][dd].astype(np.float64).sum()]
        for i in csv_rows:
            if r["done"][i]:
                csv_rows[i].append((float(r["ep_r"][i]), float(r["ep_l"][i])))
        tag, tt, dones, obs_sum, rew_sum = lines[1 + t]
        assert tag == "step" and int(tt) == t
        assert int(dones) == int(r["done"].sum()), t
        assert np.isclose(float(obs_sum), r["obs"].astype(np.float64).sum(), rtol=1e-5, atol=2e-2 if norm else 1e-3), t
        assert np.isclose(float(rew_sum), r["reward"].astype(np.float64).sum(), rtol=1e-5, atol=1e-3), t
    d = r["done"] != 0
    assert lines[-1][0] == "episodes" and [float(x) for x in lines[-1][1:]] == list(tot)   # HipVecEnv::episodeStatistics()
    lines = lines[:-1]
    assert lines[-1][0] == "infos" and int(lines[-1][1]) == int(d.sum()) and float(lines[-1][2]) == float(r["ep_l"][d].sum())
    if norm:
        # gym/monitor_csv.h: the reference's per-env Monitor file (monitor.h:20-35) for envs 0 and 3, rows = the oracle's episode records
        for i, rows in csv_rows.items():
            text = (tmp_path / "monitor" / f"env{i}.csv").read_text()
            head = f"t_start: 12345\nid: env{i}\nt,r,l"
            assert text.startswith(head) and len(rows) >= 1
            got = [l.split(",") for l in text[len(head):].splitlines()]
            assert len(got) == len(rows)
            for (tt, rr, ll), (er, el) in zip(got, rows):
                assert float(tt) > 0 and rr == "%g" % er and ll == "%g" % el
        st = v.norm_stats()
        F = 2
        om, ov_, oc = load_running_mean_std(str(tmp_path / "observation.pt"))
        rm, rv, rc = load_running_mean_std(str(tmp_path / "returns.pt"))
        assert om.shape == (F,) and rm.shape == ()
        assert np.allclose(om, st[:F], rtol=1e-5, atol=1e-7) and np.allclose(ov_, st[F:2 * F], rtol=1e-5) and oc == st[2 * F]
        assert np.isclose(float(rm), st[2 * F + 1], rtol=1e-5) and np.isclose(float(rv), st[2 * F + 2], rtol=1e-5) and rc == st[2 * F + 3]


def _fake_palette():
    """The demo's colour table: FakeEmulator::mix(77 + k) & 0xFFFFFF on even entries, 0 on odd ones."""
    def mix(z):
        m = (1 << 64) - 1
        z = (z + 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return z ^ (z >> 31)
    return np.array([0 if k & 1 else mix(77 + k) & 0xFFFFFF for k in range(256)], np.uint32)


VE_BIN = os.path.join(ROOT, "tests", "cpp", "bin", "hip_atari_vec_env_demo")


@pytest.mark.gpu
@pytest.mark.parametrize("life_loss,fire", [(True, False), (False, True)])
def test_hip_atari_vec_env_is_a_vecenv_and_matches_the_reference_chain(oracle, tmp_path, life_loss, fire):
    """gym::HipAtariVecEnv<Emu> : gym::VecEnv<false> — the Atari stack held the way the reference's trainer holds it
    (unique_ptr<VecEnv<false>>, test/atari_env.cpp:48-60) and driven only through the base-class interface (seed / reset /
    step(actions) / step(index, .) / spaces): observation [N,4,84,84] f32 on the GPU, reward / done [N,1] f32, info maps with
    "ale.lives" and "episode" — against the oracle's restatement of AtariWrapper::make + SyncVecEnv + VecFrameStack +
    VecNormAndPermute over the scripted emulator."""
    test_atari_front_end_matches_the_reference_chain(oracle, tmp_path, life_loss, fire, exe=VE_BIN)


@pytest.mark.gpu
def test_atari_front_end_max_as_written(oracle, tmp_path):
    """AtariFrontEndOptions::max_as_written + stack_as_written: the chain as the reference runs today — MaxAndSkipEnv's two
    buffers are cv::Mat headers onto AtariEnv's one frame buffer (atari_env.cpp:11-17 returns `data` by value), so cv::max sees
    the last emulated frame twice; VecFrameStack writes [prev, new, new, new] into an int8 buffer and VecNormAndPermute divides
    the signed bytes — against the oracle chain with the same aliasing and the as-written stack restatement."""
    test_atari_front_end_matches_the_reference_chain(oracle, tmp_path, True, True, as_written=True)


@pytest.mark.gpu
@pytest.mark.parametrize("life_loss,fire", [(True, False), (True, True), (False, True)])
def test_atari_front_end_matches_the_reference_chain(oracle, tmp_path, life_loss, fire, exe=None, as_written=False):
    """(f4) gym::AtariFrontEnd — NOOP / MaxAndSkip / Monitor / EpisodicLife / FireReset control flow on the host, pinned
    double-buffered upload of palette-index screens, colour + max + warp + frame stack + permute on the GPU — against the
    oracle's class-by-class restatement of AtariWrapper::make under SyncVecEnv auto-reset, over a scripted emulator."""
    exe = exe or FE_BIN
    assert os.path.exists(exe), "build tests/cpp first"
    n, T, seed = 6, 90, 4
    out = tmp_path / "fe.bin"
    r = subprocess.run([exe, str(n), str(T), str(seed), str(int(life_loss)), str(int(fire)), str(out)] + (["1"] if as_written else []),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stderr[-2000:])
    raw = np.fromfile(out, np.uint8)
    assert tuple(raw[:16].view(np.int64)) == (n, T)
    pos = 16
    def take(count, dt):
        nonlocal pos
        nbytes = count * np.dtype(dt).itemsize
        v = raw[pos:pos + nbytes].view(dt); pos += nbytes
        return v
    chain = oracle.atari_chain(n, seed, 1000, _fake_palette(), life_loss=life_loss, fire=fire)
    if as_written:
        chain.set_max_as_written()
    stack = np.zeros((n, 84, 84, 4), np.uint8)
    f0 = chain.reset()
    oracle.framestack_reset(stack, f0)
    assert np.array_equal(take(n * 7056, np.uint8).reshape(n, 84, 84), f0)
    dones = eps = life_dones = 0
    for t in range(T):
        a = (np.arange(n) * 7 + t * 3) % 6
        o = chain.step(a)
        (oracle.framestack_update_as_written if as_written else oracle.framestack_update)(stack, o["frame"])
        assert np.array_equal(take(n * 7056, np.uint8).reshape(n, 84, 84), o["frame"]), t
        assert np.array_equal(take(n, np.float32), o["reward"]), t
        assert np.array_equal(take(n, np.uint8), o["done"]), t
        assert np.array_equal(take(n, np.uint8), o["has_ep"]), t
        assert np.array_equal(take(n, np.float32), o["ep_r"]) and np.array_equal(take(n, np.float32), o["ep_l"]), t
        assert np.array_equal(take(n, np.int32), o["lives"]), t
        dones += int(o["done"].sum()); eps += int(o["has_ep"].sum()); life_dones += int((o["done"] & ~o["has_ep"]).sum())
    final = take(n * 4 * 7056, np.uint8).reshape(n, 4, 84, 84)
    assert np.array_equal(final, stack.transpose(0, 3, 1, 2))                       # VecNormAndPermute view of the VecFrameStack
    assert pos == raw.size
    assert eps >= 2 and dones >= eps and (life_dones > 0) == life_loss              # the script exercised the branches
    if exe == VE_BIN:
        # reset(int index): env 1 alone re-runs its outermost reset (sync_env.h:128-139); its stack starts over — zeros + the fresh
        # frame — the other envs keep theirs, and the next step shifts everybody
        ri = np.fromfile(str(out) + ".ri", np.uint8)
        f1 = chain.reset_index(1)
        stack[1] = 0
        stack[1, :, :, -1] = f1
        assert np.array_equal(ri[:4 * 7056].reshape(4, 84, 84), stack[1].transpose(2, 0, 1))
        o = chain.step((np.arange(n) * 5 + 1) % 6)
        oracle.framestack_update(stack, o["frame"])
        assert np.array_equal(ri[4 * 7056:].reshape(n, 4, 84, 84), stack.transpose(0, 3, 1, 2))


@pytest.mark.gpu
def test_cpp_rccl_allreduce_binding_on_the_moments_buffer():
    """INTEGRATION.md's sharded-normaliser snippet, compiled without libtorch and run on RCCL itself (one-rank
    communicator): step_begin -> ncclAllReduce(moments, f64, sum) on the launch stream -> step_finish must equal the
    unsplit step bit for bit."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "rccl_moments_demo")
    assert os.path.exists(exe), "build tests/cpp first"
    n, T = 30000, 40
    r = subprocess.run([exe, str(n), str(T)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    comm = [l for l in r.stdout.strip().splitlines() if l.startswith("rccl_comm")][-1]
    assert comm.split() == ["rccl_comm", "ranks", "1", "rank", "0"]                              # RCCL reports the communicator size it was given
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("rccl_identical")][-1]   # RCCL may print banners of its own
    tag, same, _, steps, _, count = line.split()
    assert same == "1" and int(steps) == T
    assert float(count) > n * T                                                               # the statistics moved: (T+1) batches of n rows
    forms = {l.split()[1]: dict(zip(l.split()[2::2], l.split()[3::2])) for l in r.stdout.splitlines() if l.startswith("rccl_form")}
    assert forms["moments"]["stats_bit_identical"] == "1" and forms["moments"]["payload_bytes"] == "80"
    # the exchange form without the reduce launch: the partial rows themselves are all-reduced (32 KB here), raw streams identical
    assert forms["rows"]["raw_identical"] == "1" and forms["rows"]["replicated"] == "1" and forms["rows"]["payload_bytes"] == str(256 * 16 * 8)
    assert float(forms["rows"]["worst_stat_rel"]) <= 1e-6


@pytest.mark.gpu
def test_cpp_rccl_demo_takes_a_rank_count_and_refuses_more_gpus_than_are_visible():
    """The N-rank form of the call sequence: `rccl_moments_demo n steps N` drives N GPUs from one process (ncclCommInitAll, one
    group call per exchange) against one engine that holds all N n envs.  This box has one GPU: N = 1 runs, N = 99 must fail
    with its own exit code instead of quietly running fewer ranks; the timing leg (eager / HIP graph with the RCCL call inside /
    unsplit) prints one line per exchange form."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "rccl_moments_demo")
    r = subprocess.run([exe, "4096", "5", "99"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 6 and "asked for 99 GPUs" in r.stderr
    r = subprocess.run([exe, "131072", "5", "1", "300"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    times = [dict(zip(l.split()[1::2], l.split()[2::2])) for l in r.stdout.splitlines() if l.startswith("rccl_time")]
    assert [t["form"] for t in times] == ["moments", "rows"] and all(t["ranks"] == "1" and float(t["eager_us"]) > 0 for t in times)
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("rccl_")))


@pytest.mark.gpu
def test_cpp_step_latency_demo_runs_clean():
    """BASELINE configs[1] as a C++ caller drives it (plain HIP + the C ABI, no Python in the loop): completes with a clean
    error flag; the rate it prints is recorded in DESIGN.md, not asserted beyond a sanity bound."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "step_latency_demo")
    assert os.path.exists(exe), "build tests/cpp first"
    r = subprocess.run([exe, "65536", "2000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    f = r.stdout.strip().split()
    assert f[0] == "step_latency" and int(f[-1]) == 0 and 0.5 < float(f[6]) < 200.0


@pytest.mark.gpu
@pytest.mark.parametrize("env,seed", [("cartpole", 5), ("cartpole", -1), ("mountaincar", 3)])
def test_single_env_face_matches_the_oracle_env_bit_for_bit(oracle, env, seed):
    """gym::HipCartPoleEnv / HipMountainCarEnv : gym::Env<vector<double>, int> (gym/hip_env.h) driven like
    test/classic_control.cpp:57-63 — reset, step, reset on done — against ONE oracle env without auto-reset: every double of
    every CartPole observation identical (MountainCar: <= 1e-14), rewards and dones identical, terminal observations returned
    (not replaced)."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "hip_env_demo")
    assert os.path.exists(exe), "build tests/cpp first"
    cart = env == "cartpole"
    steps, n_act = (400, 2) if cart else (1300, 3)
    r = subprocess.run([exe, env, str(seed), str(steps)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-1500:])
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    v = oracle.vec(CARTPOLE if cart else MOUNTAINCAR, 1, 0, 0, threads=1)
    if seed >= 0:
        v.seed(seed)                                   # never seeded (-1): the default-constructed mt19937 (env.h:83)
    v.set_auto_reset(False)
    _, o64 = v.reset(want64=True)
    k = 0
    assert lines[k][0] == "reset" and [float(x) for x in lines[k][1:]] == list(o64[0]); k += 1
    episodes = 0
    for t in range(steps):
        a = ((t * 2654435761 & 0xFFFFFFFF) >> 7) % n_act
        o = v.step(np.array([a], np.int32), want64=True)
        tag, tt, aa, rew, done, *obs = lines[k]; k += 1
        assert tag == "step" and int(tt) == t and int(aa) == a
        assert float(rew) == o["reward"][0] and int(done) == int(o["done"][0]), t
        if cart:
            assert [float(x) for x in obs] == list(o["obs64"][0]), t        # %.17g round-trips a double exactly
        else:   # MountainCar uses the double cos(3 pos) unrounded: the device library's and glibc's differ in the last bit now and then
            assert np.abs(np.array([float(x) for x in obs]) - o["obs64"][0]).max() <= 1e-14, t
        if o["done"][0]:
            episodes += 1
            _, r64 = v.reset_index(0)
            assert lines[k][0] == "reset" and [float(x) for x in lines[k][1:]] == list(r64); k += 1
    assert lines[k] == ["episodes", str(episodes)] and episodes >= (5 if cart else 0)
    assert "already returned done = True" in r.stderr or episodes < 2     # the misuse path printed the reference's warning


FS_BIN = os.path.join(ROOT, "tests", "cpp", "bin", "hip_vec_frame_stack_demo")


def _scripted_image(n, H, W, t):
    """tests/cpp/hip_vec_frame_stack_demo.cpp ScriptedImageVecEnv::frame, in numpy (uint32 arithmetic)."""
    e = (np.arange(n, dtype=np.uint64)[:, None] * 2654435761) & 0xFFFFFFFF
    q = (np.arange(H * W, dtype=np.uint64)[None, :] * 2246822519) & 0xFFFFFFFF
    v = e ^ np.uint64((t * 40503) & 0xFFFFFFFF) ^ q
    v ^= v >> np.uint64(15); v = (v * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF); v ^= v >> np.uint64(13)
    return ((v >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8).reshape(n, H, W)


REF_DROPIN = os.path.join(ROOT, "oracle", "_ref", "ref_dropin_tu")


@pytest.mark.gpu
@pytest.mark.parametrize("templates", ["mirror", "reference"])
@pytest.mark.parametrize("mode,where", [("plain", "host"), ("plain", "device"), ("as_written", "host"), ("norm", "device"), ("norm_as_written", "host"), ("clear", "host"),
                                        ("reset_index_as_written", "host")])
def test_hip_vec_frame_stack_cpp_wraps_any_vec_env(oracle, tmp_path, mode, where, templates):
    """gym::HipVecFrameStack : VecEnvWrapper<false> (gym/hip_vec_frame_stack.h) held as unique_ptr<VecEnv<false>> over a scripted
    VecEnv<false> of {H, W, 1} byte images: the reference's VecFrameStack, and VecNormAndPermute(VecFrameStack(.)), as wrappers of
    their own — observation space, every observation against the oracle's frame stack (both views), reward / done untouched.
    "reference": the same probe (tests/cpp/frame_stack_probe.h) compiled with GYMHIP_USE_REFERENCE_HEADERS — the wrapper then derives
    from the reference's own VecEnvWrapper<false> and builds the reference's own Box spaces (oracle/_ref/ref_dropin_tu framestack)."""
    n, H, W, k, steps = 3, 16, 12, 4, 11
    out = tmp_path / "fs.bin"
    if templates == "reference":
        if not os.path.exists(REF_DROPIN):
            pytest.skip("oracle/_ref not built (it is built in the container that mounts the reference and travels with the tree)")
        cmd = [REF_DROPIN, "framestack"]
    else:
        assert os.path.exists(FS_BIN), "build tests/cpp first"
        cmd = [FS_BIN]
    r = subprocess.run(cmd + [mode, str(n), str(H), str(W), str(k), str(steps), where, str(out)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    lines = r.stdout.strip().splitlines()
    norm, aw = mode.startswith("norm"), mode in ("as_written", "norm_as_written")
    assert lines[0] == (f"space {k} {H} {W} type 6 nEnvs {n}" if norm else f"space {H} {W} {k} type 1 nEnvs {n}")      # c10::kFloat = 6, kChar = 1
    assert lines[1] == (f"obs dtype Float dims {n} {k} {H} {W} cuda 1" if norm else f"obs dtype {'Char' if aw else 'Byte'} dims {n} {H} {W} {k} cuda 1")
    bcast = mode == "reset_index_as_written"
    ri_dims = f"{n} {H} {W} {k}" if bcast else (f"{k} {H} {W}" if norm else f"{H} {W} {k}")
    assert lines[2] == f"reset_index dims {ri_dims}"
    raw = np.fromfile(out, np.uint8)
    per = n * H * W * k * (4 if norm else 1)
    obs = raw[:per * (steps + 1)].reshape(steps + 1, per)
    rd = raw[per * (steps + 1):].view(np.float32).reshape(steps, 2, n)
    stack = np.zeros((n, H, W, k), np.uint8)
    for t in range(steps + 1):
        f = _scripted_image(n, H, W, t)
        if t == 0:
            oracle.framestack_reset(stack, f)
        else:
            (oracle.framestack_update_as_written if aw else oracle.framestack_update)(stack, f)
            done = np.array([(t + e) % 5 == 0 for e in range(n)])
            if mode == "clear":
                stack[done, :, :, :-1] = 0                                   # SB3's clear: a finished env keeps only its newest frame
            assert np.array_equal(rd[t - 1, 0], np.arange(n, dtype=np.float32) + t) and np.array_equal(rd[t - 1, 1], done.astype(np.float32))
        if norm:
            want = (oracle.norm_permute_as_written if aw else oracle.norm_permute)(stack)
            assert np.array_equal(obs[t].view(np.float32).reshape(n, k, H, W), want), t
        else:
            assert np.array_equal(obs[t].reshape(n, H, W, k), stack), t
    if aw:
        return   # the reset(int index) leg is checked on the intended view (the as-written VIEW has its own first-channel bookkeeping)
    # reset(int index) (sync_env.h:128-139 under vec_frame_stack.cpp:34-37): env 1 alone starts over — zeros + its fresh frame, the
    # others untouched — or, as the reference's code has it, every stack shifts and the ONE frame is broadcast over the batch
    ri = np.fromfile(str(out) + ".ri", np.uint8)
    fresh = _scripted_image(2, H, W, 1001)[1]                                   # the inner env's reset(1): env 1's frame at its own clock
    if bcast:
        oracle.framestack_update(stack, np.ascontiguousarray(np.broadcast_to(fresh, (n, H, W))))
        first, rest = ri[:per].reshape(n, H, W, k), ri[per:]
        assert np.array_equal(first, stack)
    else:
        stack[1] = 0
        stack[1, :, :, -1] = fresh
        one = per // n
        first, rest = ri[:one], ri[one:]
        want1 = oracle.norm_permute(stack)[1] if norm else stack[1]
        assert np.array_equal(first.view(np.float32).reshape(k, H, W) if norm else first.reshape(H, W, k), want1)
    oracle.framestack_update(stack, _scripted_image(n, H, W, steps + 1))
    if mode == "clear":
        stack[np.array([(steps + 1 + e) % 5 == 0 for e in range(n)]), :, :, :-1] = 0
    want = oracle.norm_permute(stack) if norm else stack
    assert np.array_equal(rest.view(np.float32).reshape(n, k, H, W) if norm else rest.reshape(n, H, W, k), want)

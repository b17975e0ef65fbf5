"""GPU parity of the Atari preprocessing kernels (BASELINE config 4) against the
CPU oracle.  Contract: resized uint8 frames within +-1 LSB (they are in fact compared exactly, since both sides follow
the same op order), frame stack / permute / wrapper lanes exact.

The oracle's resize/gray restate OpenCV 4.x, which is NOT in this image: parity with OpenCV itself is unpinned
(oracle/gym_oracle.cpp header); hand-derived known answers live in tests/test_oracle_atari.py.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 210, 160


def counter_frames(seed, n, t=0):
    """Synthetic frames u8 [n][2][210][160][3] from a counter hash of (seed, env, t, pixel)."""
    idx = np.arange(n * 2 * H * W * 3, dtype=np.uint64) + np.uint64(seed * 0x9E3779B1 + t * 0x85EBCA77)
    with np.errstate(over="ignore"):
        z = idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(32))
    return (z & np.uint64(255)).astype(np.uint8).reshape(n, 2, H, W, 3)


def structured_frames():
    """Edge cases: constants, gradients, single-pixel impulses, one frame dominating the max."""
    fr = np.zeros((8, 2, H, W, 3), np.uint8)
    fr[1] = 255
    fr[2, 0] = 17; fr[2, 1] = 200                                   # max picks frame 1 everywhere
    fr[3, 0, :, :, 0] = np.arange(W, dtype=np.uint8)[None, :]       # horizontal ramp in R
    fr[3, 1, :, :, 1] = (np.arange(H) % 256).astype(np.uint8)[:, None]  # vertical ramp in G
    fr[4, 0, 100, 77] = (255, 255, 255)                             # impulse
    fr[4, 1, 0, 0] = (255, 0, 0); fr[4, 1, H - 1, W - 1] = (0, 0, 255)  # corners
    rng = np.random.default_rng(3)
    fr[5] = rng.integers(0, 256, fr[5].shape, dtype=np.uint8)
    fr[6, 0] = fr[5, 1]; fr[6, 1] = fr[5, 0]                        # max is symmetric
    fr[7, :, ::2] = 255                                             # stripes: exercises the 5:2 vertical taps
    return fr


@pytest.mark.parametrize("gray_bits", [15, 14])
def test_preprocess_structured_and_random(oracle, gray_bits):
    from gym_amd.atari import AtariPreprocessor
    fr = np.concatenate([structured_frames(), counter_frames(7, 24)], axis=0)
    n = fr.shape[0]
    pp = AtariPreprocessor(n, gray_bits=gray_bits)
    got = pp.preprocess(torch.from_numpy(fr).cuda()).cpu().numpy()
    want = oracle.atari_preprocess(fr, gray15=int(gray_bits == 15))
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1, f"max LSB error {diff.max()}"
    assert (diff != 0).mean() == 0.0, "same op order on both sides: expected bit-exact"
    assert np.array_equal(got[5], got[6])                           # max(a,b) == max(b,a)
    assert (got[0] == 0).all() and (got[1] == 255).all() and (got[2] == 200).all()


def synthetic_palette(seed):
    """256 words 0x00RRGGBB in the shape of the emulator's NTSC table (atari_ntsc_rgb_palette.h): 128 colours on the even
    indices, zeros on the odd ones; the colours themselves are seeded random (the kernel is table-agnostic)."""
    rng = np.random.default_rng(seed)
    pal = np.zeros(256, np.uint32)
    pal[0::2] = rng.integers(0, 1 << 24, 128, dtype=np.uint32)
    pal[0] = 0x000000; pal[14] = 0xECECEC                           # the table's first row ends: black and near-white
    return pal


def colour_on_host(screens, pal):
    """getScreenRGB2 (ale_c_wrapper.cpp:28-41): index -> (R, G, B) bytes."""
    z = pal[screens]
    return np.stack([(z >> 16) & 255, (z >> 8) & 255, z & 255], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("gray_bits", [15, 14])
def test_indexed_screens_equal_host_coloured_rgb(oracle, gray_bits):
    """Palette-index screens + in-kernel colour table == colouring on the host and running the RGB path == the oracle."""
    from gym_amd.atari import AtariPreprocessor
    n = 24
    pal = synthetic_palette(11)
    rng = np.random.default_rng(5)
    scr = (rng.integers(0, 128, (n, 2, H, W), dtype=np.uint8) * 2).astype(np.uint8)   # ALE emits even indices
    scr[0] = 0; scr[1] = 14; scr[2, 0] = 0; scr[2, 1] = 14                              # constants, one frame dominating
    scr[3] = rng.integers(0, 256, scr[3].shape, dtype=np.uint8)                          # odd indices too (table entry 0)
    scr[4, :, 50:120, 30:100] = 14; scr[4, 1, 60:70] = 0                                 # blocks: large same-colour regions
    rgb = colour_on_host(scr, pal)
    pp = AtariPreprocessor(n, gray_bits=gray_bits)
    with pytest.raises(Exception):
        pp.preprocess(torch.from_numpy(scr).cuda())                                      # no colour table yet
    pp.set_palette(pal)
    got = pp.preprocess(torch.from_numpy(scr).cuda()).cpu().numpy()
    via_rgb = pp.preprocess(torch.from_numpy(rgb).cuda()).cpu().numpy()
    want = oracle.atari_preprocess(rgb, gray15=int(gray_bits == 15))
    assert np.array_equal(got, via_rgb)
    assert np.array_equal(got, want)
    assert (got[0] == 0).all() and len(np.unique(got[1])) == 1
    # the stack path: reset + steps on screens vs on host-coloured frames
    pp2 = AtariPreprocessor(n, gray_bits=gray_bits)
    pp.reset(torch.from_numpy(scr).cuda()); pp2.reset(torch.from_numpy(rgb).cuda())
    for t in range(5):
        s2 = np.roll(scr, t + 1, axis=2)
        pp.step(torch.from_numpy(s2).cuda()); pp2.step(torch.from_numpy(colour_on_host(s2, pal)).cuda())
    assert np.array_equal(pp.stacked_nhwc().cpu().numpy(), pp2.stacked_nhwc().cpu().numpy())


def test_frame_stack_ring_matches_shift_semantics(oracle):
    """reset + 9 steps (> 2 ring revolutions for k=4): NHWC stack and the NCHW/255 float view vs the oracle."""
    from gym_amd.atari import AtariPreprocessor
    n, k = 16, 4
    pp = AtariPreprocessor(n, n_stack=k)
    stack = np.zeros((n, 84, 84, k), np.uint8)
    f0 = counter_frames(1, n, 0)
    pp.reset(torch.from_numpy(f0).cuda())
    oracle.framestack_reset(stack, oracle.atari_preprocess(f0))
    assert np.array_equal(pp.stacked_nhwc().cpu().numpy(), stack)
    assert (stack[..., :k - 1] == 0).all()
    for t in range(1, 10):
        f = counter_frames(1, n, t)
        pp.step(torch.from_numpy(f).cuda())
        oracle.framestack_update(stack, oracle.atari_preprocess(f))
        assert np.array_equal(pp.stacked_nhwc().cpu().numpy(), stack), t
        assert np.array_equal(pp.stacked_nchw_f32().cpu().numpy(), oracle.norm_permute(stack)), t


@pytest.mark.parametrize("k", [1, 3, 8])
def test_frame_stack_other_depths(oracle, k):
    """k = 8 is the depth the reference's own Atari test stacks (test/atari_env.cpp:50-51: VecFrameStack(..., 8))."""
    from gym_amd.atari import AtariPreprocessor
    n = 4
    pp = AtariPreprocessor(n, n_stack=k)
    stack = np.zeros((n, 84, 84, k), np.uint8)
    f0 = counter_frames(2, n, 0)
    pp.reset(torch.from_numpy(f0).cuda())
    oracle.framestack_reset(stack, oracle.atari_preprocess(f0))
    for t in range(1, 2 * k + 3):
        f = counter_frames(2, n, t)
        pp.step(torch.from_numpy(f).cuda())
        oracle.framestack_update(stack, oracle.atari_preprocess(f))
        assert np.array_equal(pp.stacked_nhwc().cpu().numpy(), stack)
        assert np.array_equal(pp.stacked_nchw_f32().cpu().numpy(), oracle.norm_permute(stack))


@pytest.mark.parametrize("k", [2, 3, 4, 8])
def test_frame_stack_as_written_view(oracle, k):
    """gymhip_atari_set_stack_view(GYMHIP_STACK_AS_WRITTEN): the reference's VecFrameStack + VecNormAndPermute outputs as its
    code computes them today (channel 0 = the previous call's channel 1, channels 1.. = the newest frame; int8 buffer, so
    / 255 divides signed bytes), against the oracle's literal restatements — which tests/test_oracle_atari.py pins to the
    torch calls themselves.  A second reset() starts over; switching back gives the intended view of the same ring."""
    from gym_amd.atari import AtariPreprocessor
    n = 3
    pp = AtariPreprocessor(n, n_stack=k)
    for round_ in range(2):
        stack = np.zeros((n, 84, 84, k), np.uint8)
        intended = np.zeros((n, 84, 84, k), np.uint8)
        f0 = counter_frames(5 + round_, n, 0)
        pp.reset(torch.from_numpy(f0).cuda())
        oracle.framestack_reset(stack, oracle.atari_preprocess(f0))
        oracle.framestack_reset(intended, oracle.atari_preprocess(f0))
        for t in range(0, k + 4):
            if t:
                f = counter_frames(5 + round_, n, t)
                pp.step(torch.from_numpy(f).cuda())
                g = oracle.atari_preprocess(f)
                oracle.framestack_update_as_written(stack, g)
                oracle.framestack_update(intended, g)
            pp.set_stack_view("as_written")
            got = pp.stacked_nhwc()
            assert got.dtype == torch.int8 and np.array_equal(got.view(torch.uint8).cpu().numpy(), stack), (round_, t)
            want = oracle.norm_permute_as_written(stack)
            assert want.min() < 0                                                           # pixels >= 128 really come out negative
            assert np.array_equal(pp.stacked_nchw_f32().cpu().numpy(), want), (round_, t)
            for dt in (torch.float16, torch.bfloat16):
                assert torch.equal(pp.stacked_nchw_16(dt).cpu(), torch.from_numpy(want).to(dt)), (round_, t, dt)
            pp.set_stack_view("intended")
            assert np.array_equal(pp.stacked_nhwc().cpu().numpy(), intended), (round_, t)
            assert np.array_equal(pp.stacked_nchw_f32().cpu().numpy(), oracle.norm_permute(intended)), (round_, t)
    with pytest.raises(Exception):
        AtariPreprocessor(n, n_stack=1).set_stack_view("as_written")
    with pytest.raises(Exception):
        AtariPreprocessor(n, n_stack=4, clear_on_done=True).set_stack_view("as_written")


def test_clear_on_done_option():
    """clear_on_done=1 (SB3 semantics; the reference never clears): finished envs see zeros in the older channels."""
    from gym_amd.atari import AtariPreprocessor
    n = 6
    pp = AtariPreprocessor(n, clear_on_done=True)
    ref = AtariPreprocessor(n, clear_on_done=False)
    for t in range(3):
        f = torch.from_numpy(counter_frames(4, n, t)).cuda()
        (pp.reset if t == 0 else pp.step)(f)
        (ref.reset if t == 0 else ref.step)(f)
    done = torch.tensor([0, 1, 0, 0, 1, 0], dtype=torch.uint8, device="cuda")
    f = torch.from_numpy(counter_frames(4, n, 3)).cuda()
    pp.step(f, done)
    ref.step(f, done)
    a, b = pp.stacked_nhwc().cpu().numpy(), ref.stacked_nhwc().cpu().numpy()
    d = done.cpu().numpy().astype(bool)
    assert np.array_equal(a[~d], b[~d])
    assert np.array_equal(a[d][..., -1], b[d][..., -1]) and (a[d][..., :-1] == 0).all() and (b[d][..., :-1] != 0).any()


@pytest.mark.parametrize("grid", [1, 3, 5, 64])
def test_persistent_kernel_walks_frames_exactly(oracle, monkeypatch, grid):
    """The Atari geometry's RGB path is a persistent kernel (atari_preprocess_fixed_kernel): a workgroup walks frames
    blockIdx, += grid, parks each finished plane in LDS and stores it from inside the next frame's load stream.  With the
    grid forced small (GYMHIP_ATARI_GRID, read per launch) one workgroup handles many frames — 1 workgroup for all 13,
    uneven shares, more workgroups than frames — and every plane must still be the oracle's, in the right ring slot, with
    clear-on-done applied to the right envs."""
    from gym_amd.atari import AtariPreprocessor
    monkeypatch.setenv("GYMHIP_ATARI_GRID", str(grid))
    n = 13
    f0, f1 = counter_frames(21, n, 0), counter_frames(21, n, 1)
    f0[:8] = structured_frames()
    pp = AtariPreprocessor(n, clear_on_done=True)
    assert np.array_equal(pp.preprocess(torch.from_numpy(f0).cuda()).cpu().numpy(), oracle.atari_preprocess(f0))
    stack = np.zeros((n, 84, 84, 4), np.uint8)
    pp.reset(torch.from_numpy(f0).cuda())
    oracle.framestack_reset(stack, oracle.atari_preprocess(f0))
    done = np.zeros(n, np.uint8); done[[0, 4, 12]] = 1
    pp.step(torch.from_numpy(f1).cuda(), torch.from_numpy(done).cuda())
    oracle.framestack_update(stack, oracle.atari_preprocess(f1))
    stack[done.astype(bool), :, :, :-1] = 0
    assert np.array_equal(pp.stacked_nhwc().cpu().numpy(), stack)
    # the generic kernel (one workgroup per frame; GYMHIP_ATARI_GENERIC is read per launch too) on the same frames: identical planes
    fixed = pp.preprocess(torch.from_numpy(f1).cuda()).clone()
    monkeypatch.setenv("GYMHIP_ATARI_GENERIC", "1")
    assert torch.equal(pp.preprocess(torch.from_numpy(f1).cuda()), fixed)


@pytest.mark.parametrize("n", [1, 2, 767, 768, 769, 1537, 2311])
def test_persistent_kernel_at_its_own_grid_equals_the_generic_kernel(monkeypatch, n):
    """Env counts around the persistent grid (768 workgroups on an MI355X: fewer frames than workgroups, exactly one each, one
    more, two rounds and a bit, three rounds and a bit): the planes equal the one-workgroup-per-frame kernel's, which the
    other tests pin against the oracle."""
    from gym_amd.atari import AtariPreprocessor
    g = torch.Generator(device="cuda").manual_seed(n)
    frames = torch.randint(0, 256, (n, 2, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    pp = AtariPreprocessor(n)
    fixed = pp.preprocess(frames).clone()
    pp.reset(frames); pp.step(frames)
    ring_fixed = pp.stacked_nhwc().clone()
    monkeypatch.setenv("GYMHIP_ATARI_GENERIC", "1")
    assert torch.equal(pp.preprocess(frames), fixed)
    pp.reset(frames); pp.step(frames)
    assert torch.equal(pp.stacked_nhwc(), ring_fixed)
    assert torch.equal(ring_fixed[..., -1], fixed) and torch.equal(ring_fixed[..., -2], fixed) and (ring_fixed[..., :2] == 0).all()


def test_a_recorded_ring_revolution_replays_like_eager_steps():
    """HIP graph capture of the image path.  The ring's `head` is HOST state (gymhip.h), so a recording of one step would
    replay into the same slot; a recording of n_stack consecutive steps — a whole revolution, each with its NCHW view — leaves
    `head` where it was and replays coherently.  The calls neither allocate nor synchronise, so capture succeeds; two replays
    over fresh frames equal eager stepping bit for bit (persistent kernel included: 13 envs, forced grid of 5 workgroups)."""
    import os
    from gym_amd.atari import AtariPreprocessor
    n, k = 13, 4
    os.environ["GYMHIP_ATARI_GRID"] = "5"
    try:
        pp, ref = AtariPreprocessor(n, n_stack=k), AtariPreprocessor(n, n_stack=k)
        f0 = torch.from_numpy(counter_frames(31, n, 0)).cuda()
        pp.reset(f0); ref.reset(f0)
        fbuf = [torch.empty_like(f0) for _ in range(k)]
        out = [torch.empty((n, k, 84, 84), dtype=torch.float32, device="cuda") for _ in range(k)]
        t = 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                   # warm-up revolution on the capture stream, eager
            for j in range(k):
                fbuf[j].copy_(torch.from_numpy(counter_frames(31, n, t + j)).cuda())
                pp.step(fbuf[j]); pp.stacked_nchw_f32(out[j])
                ref.step(fbuf[j])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        t += k
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for j in range(k):
                pp.step(fbuf[j]); pp.stacked_nchw_f32(out[j])
        for _ in range(2):
            for j in range(k):
                fbuf[j].copy_(torch.from_numpy(counter_frames(31, n, t + j)).cuda())
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for j in range(k):
                ref.step(fbuf[j])
                assert torch.equal(out[j], ref.stacked_nchw_f32()), (t, j)
            t += k
        assert torch.equal(pp.stacked_nhwc(), ref.stacked_nhwc())
    finally:
        del os.environ["GYMHIP_ATARI_GRID"]


def test_config4_full_size_properties():
    """BASELINE config 4 at full size (4096 envs, 826 MB of frames): size-independent properties —
    (1) every env given the same frame pair yields the same 84x84 image (and equals the small-batch result);
    (2) max idempotence: (f, f) == preprocess of f alone; (3) monotonicity: raising pixel values never lowers outputs."""
    from gym_amd.atari import AtariPreprocessor
    n = 4096
    base = torch.from_numpy(counter_frames(9, 2)).cuda()              # [2][2][H][W][3]
    frames = base[:1].expand(n, 2, H, W, 3).contiguous()
    frames[1::2] = base[1]
    pp = AtariPreprocessor(n)
    out = pp.preprocess(frames)
    small = AtariPreprocessor(2).preprocess(base)
    assert torch.equal(out[0::2], small[0].expand(n // 2, 84, 84)) and torch.equal(out[1::2], small[1].expand(n // 2, 84, 84))
    same = frames.clone(); same[:, 1] = same[:, 0]
    lone = pp.preprocess(same)
    brighter = torch.maximum(same, torch.full_like(same, 128))
    assert (pp.preprocess(brighter) >= lone).all()
    pp.reset(frames)
    pp.step(same)
    st = pp.stacked_nhwc()
    assert torch.equal(st[..., -1], lone) and torch.equal(st[..., -2], out) and (st[..., :2] == 0).all()


def test_frames_beyond_4_gib_index_arithmetic():
    """Maximum sizes: 24,000 envs = 4.84 GB of RGB frames, so frame pairs sit beyond byte offsets 2^31 and 2^32 (env 10,652 and
    env 21,304 straddle them); batch results must equal the small-batch results of the same frames at the first envs, across
    both boundaries and at the tail — for the preprocess output, the ring after reset + 2 steps, and the NCHW f32 view."""
    from gym_amd.atari import AtariPreprocessor
    n, m = 24000, 96
    g = torch.Generator(device="cuda").manual_seed(3)
    frames = [torch.randint(0, 256, (n, 2, H, W, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(2)]
    assert frames[0].numel() > (1 << 32)
    big = AtariPreprocessor(n)
    out = big.preprocess(frames[0])
    big.reset(frames[0]); big.step(frames[1]); big.step(frames[0])
    nhwc, nchw = big.stacked_nhwc(), big.stacked_nchw_f32()
    for off in (0, (1 << 31) // (2 * H * W * 3) - m // 2, (1 << 32) // (2 * H * W * 3) - m // 2, n - m):
        small = AtariPreprocessor(m)
        assert torch.equal(small.preprocess(frames[0][off:off + m].contiguous()), out[off:off + m]), off
        small.reset(frames[0][off:off + m].contiguous()); small.step(frames[1][off:off + m].contiguous()); small.step(frames[0][off:off + m].contiguous())
        assert torch.equal(small.stacked_nhwc(), nhwc[off:off + m]) and torch.equal(small.stacked_nchw_f32(), nchw[off:off + m]), off


def test_host_env_post_step_lanes(oracle):
    """EpisodicLife + Monitor + ClipReward lanes over scripted (reward, done, lives) streams."""
    from gym_amd.atari import host_env_post_step
    n, T = 4096, 300
    rng = np.random.default_rng(0)
    lives = np.full(n, 5, np.int32)
    st_o = dict(lives=np.zeros(n, np.int32), was_real_done=np.ones(n, np.uint8), ret=np.zeros(n, np.float32), len=np.zeros(n, np.float32))
    st_g = {k: torch.from_numpy(v.copy()).cuda() for k, v in st_o.items()}
    saw_life_done = saw_real_done = 0
    for t in range(T):
        rew = rng.choice(np.array([0.0, 0.0, 1.0, -3.0, 7.5, 0.25], np.float32), n).astype(np.float32)
        lose = rng.random(n) < 0.03
        lives = np.where(lose, lives - 1, lives).astype(np.int32)
        done = (lives <= 0).astype(np.uint8)
        want = oracle.host_env_post_step(rew, done, lives, st_o, clip=1)
        got = host_env_post_step(torch.from_numpy(rew).cuda(), torch.from_numpy(done).cuda(), torch.from_numpy(lives).cuda(), st_g, True)
        for key in ("reward", "done", "has_ep", "ep_r", "ep_l"):
            assert np.array_equal(got[key].cpu().numpy(), want[key]), (t, key)
        for key in st_o:
            assert np.array_equal(st_g[key].cpu().numpy(), st_o[key]), (t, key)
        saw_life_done += int(((want["done"] != 0) & (done == 0)).sum())
        saw_real_done += int(done.sum())
        lives = np.where(done != 0, 5, lives).astype(np.int32)       # real reset restores the lives
    assert saw_life_done > 0 and saw_real_done > 0


def test_atari_errors():
    from gym_amd import _lib as L
    from gym_amd.atari import AtariPreprocessor
    with pytest.raises(L.GymhipError):
        AtariPreprocessor(4, dst_hw=(84, 83))
    with pytest.raises(L.GymhipError):
        AtariPreprocessor(4, gray_bits=12)
    pp = AtariPreprocessor(4)
    with pytest.raises(L.GymhipError, match="before reset"):
        pp.step(torch.zeros((4, 2, H, W, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        pp.preprocess(torch.zeros((4, 2, H, W, 4), dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k", [4, 3])
def test_nchw_16_bit_views_round_the_f32_view_once(oracle, dtype, k):
    """gymhip_atari_materialize_nchw_16: x / 255 in float32 (vec_atari.cpp:74-76, the oracle's norm_permute), rounded once to
    nearest-even half / bfloat16 — bit-exact against that rounding of the oracle's tensor; every byte value is covered."""
    from gym_amd.atari import AtariPreprocessor
    n = 5
    pp = AtariPreprocessor(n, n_stack=k)
    stack = np.zeros((n, 84, 84, k), np.uint8)
    f0 = counter_frames(3, n, 0)
    pp.reset(torch.from_numpy(f0).cuda())
    oracle.framestack_reset(stack, oracle.atari_preprocess(f0))
    rows = np.arange(210)
    for t in range(1, k + 3):
        f = counter_frames(3, n, t)
        if t >= 3:   # gray bands, a different ramp per env and step: together they put every byte value 0..255 into the ring
            v = ((rows[None, :] * 5 // 4 + 51 * np.arange(n)[:, None] + 17 * t) % 256).astype(np.uint8)
            f = np.broadcast_to(v[:, None, :, None, None], (n, 2, 210, 160, 3)).copy()
        pp.step(torch.from_numpy(f).cuda())
        oracle.framestack_update(stack, oracle.atari_preprocess(f))
    want = torch.from_numpy(oracle.norm_permute(stack)).to(dtype)          # torch rounds f32 -> f16 / bf16 to nearest even
    got = pp.stacked_nchw_16(dtype).cpu()
    assert got.dtype == dtype and got.shape == (n, k, 84, 84)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert len(np.unique(stack)) >= 200                                     # the frames exercise (nearly) every byte value
    with pytest.raises(ValueError):
        pp.stacked_nchw_16(torch.float32)


@pytest.mark.parametrize("src_hw,dst_hw", [((210, 160), (84, 84)), ((250, 160), (84, 84)), ((210, 160), (105, 80)), ((210, 160), (96, 96)),
                                            ((224, 240), (84, 84)), ((84, 84), (84, 84)), ((100, 64), (42, 32))])
def test_other_geometries_match_the_oracle(oracle, src_hw, dst_hw):
    """WarpFrameEnv takes (width, height) arguments (warp_frame.h:22-30); the generic INTER_AREA tables cover any
    down-scale up to 3x per axis: fractional, integer (2x), identity and mixed scales, bit-exact against the oracle.
    (At an exact 2x scale OpenCV itself takes its integer fast path, whose SIMD form rounds .5 ties up where the generic
    path rounds to even: oracle and kernel both follow the generic tables there — within the +-1 LSB contract, and as
    unpinned against OpenCV as the rest of this row.)"""
    from gym_amd.atari import AtariPreprocessor
    n = 3
    rng = np.random.default_rng(11)
    f = rng.integers(0, 256, (n, 2, src_hw[0], src_hw[1], 3), dtype=np.uint8)
    pp = AtariPreprocessor(n, src_hw=src_hw, dst_hw=dst_hw)
    got = pp.preprocess(torch.from_numpy(f).cuda()).cpu().numpy()
    want = oracle.atari_preprocess(f, dh=dst_hw[0], dw=dst_hw[1])
    assert got.shape == want.shape == (n, dst_hw[0], dst_hw[1])
    assert np.array_equal(got, want)


def _planes(n, H, W, t, seed=5):
    return np.random.default_rng(seed * 1000 + t).integers(0, 256, (n, H, W, 1), dtype=np.uint8)


@pytest.mark.parametrize("k,hw,view", [(4, (84, 84), "intended"), (4, (84, 84), "as_written"), (3, (64, 64), "intended"), (1, (8, 12), "intended"), (8, (20, 4), "as_written")])
def test_push_already_preprocessed_planes_is_vec_frame_stack_on_its_own(oracle, k, hw, view):
    """gymhip_atari_push_reset / _push: VecFrameStack (vec_frame_stack.cpp:22-32) over observations that are already H x W
    single-channel images — planes from the host or the device, both views, against the oracle's frame stack; and the
    result equals feeding the fused preprocess path frames whose preprocessing is those planes (same ring, same views)."""
    from gym_amd.atari import AtariPreprocessor
    n, (H, W) = 5, hw
    pp = AtariPreprocessor(n, n_stack=k, src_hw=hw, dst_hw=hw)
    pp.set_stack_view(view)
    stack = np.zeros((n, H, W, k), np.uint8)
    p0 = _planes(n, H, W, 0)
    pp.push_reset(torch.from_numpy(p0))                                   # a host tensor
    oracle.framestack_reset(stack, p0[..., 0])
    assert np.array_equal(pp.stacked_nhwc().cpu().numpy().view(np.uint8), stack)
    upd = oracle.framestack_update_as_written if view == "as_written" else oracle.framestack_update
    for t in range(1, 2 * k + 3):
        p = _planes(n, H, W, t)
        pp.push(torch.from_numpy(p).cuda() if t % 2 else torch.from_numpy(p).view(torch.int8))   # device uint8 / host int8 (the reference's kChar)
        upd(stack, p[..., 0])
        assert np.array_equal(pp.stacked_nhwc().cpu().numpy().view(np.uint8), stack), t
        want = oracle.norm_permute_as_written(stack) if view == "as_written" else oracle.norm_permute(stack)
        assert np.array_equal(pp.stacked_nchw_f32().cpu().numpy(), want), t
    with pytest.raises(ValueError):
        pp.push(torch.zeros((n, H, W + 1, 1), dtype=torch.uint8))
    with pytest.raises(Exception):
        AtariPreprocessor(n, n_stack=k, src_hw=hw, dst_hw=hw).push(torch.from_numpy(p0))     # push before reset


def test_push_clear_on_done_and_mixing_with_the_fused_path():
    """clear_on_done through the push path; and pushes interleave with fused steps on one ring (the head is shared)."""
    from gym_amd.atari import AtariPreprocessor
    n, k = 6, 4
    pp = AtariPreprocessor(n, n_stack=k, clear_on_done=True)
    frames = [counter_frames(9, n, t) for t in range(4)]
    planes = [pp.preprocess(torch.from_numpy(f).cuda()).clone() for f in frames]
    ref = AtariPreprocessor(n, n_stack=k, clear_on_done=True)
    done = torch.tensor([1, 0, 0, 1, 0, 0], dtype=torch.uint8)
    pp.push_reset(planes[0]); ref.reset(torch.from_numpy(frames[0]).cuda())
    pp.push(planes[1]); ref.step(torch.from_numpy(frames[1]).cuda())
    pp.step(torch.from_numpy(frames[2]).cuda(), done.cuda()); ref.step(torch.from_numpy(frames[2]).cuda(), done.cuda())   # a fused step in between
    assert torch.equal(pp.stacked_nhwc(), ref.stacked_nhwc())
    pp.push(planes[3], 1 - done); ref.step(torch.from_numpy(frames[3]).cuda(), (1 - done).cuda())
    a = pp.stacked_nhwc()
    assert torch.equal(a, ref.stacked_nhwc())
    d = (1 - done).bool()
    assert (a[d][..., :-1] == 0).all() and (a[~d][..., :-1] != 0).any()


class _ImageVecEnv:
    """A host VecEnv of single-channel images (WarpFrame's {H, W, 1} space), scripted: what HipVecFrameStack wraps."""

    def __new__(cls, n, H, W, device):
        from gym_amd import spaces as S
        from gym_amd.vec_env import VecEnv, VecEnvStepReturn

        class Env(VecEnv):
            def __init__(self):
                super().__init__(n, S.makeBoxSpace(0, 255, [H, W, 1], dtype=np.uint8), S.makeDiscreteSpace(3))
                self.t = 0

            def _obs(self):
                x = torch.from_numpy(_planes(n, H, W, self.t, seed=8))
                return x.to(device)

            def reset(self, index=None):
                self.t = 0
                return self._obs()

            def stepAsync(self, actions):
                self.actions = actions

            def stepWait(self):
                self.t += 1
                done = torch.tensor([[float((self.t + e) % 4 == 0)] for e in range(n)])
                return VecEnvStepReturn(self._obs(), torch.full((n, 1), float(self.t)), done, [{} for _ in range(n)])

        return Env()


@pytest.mark.parametrize("device", ["cpu", "cuda"])
@pytest.mark.parametrize("norm,view", [(False, "intended"), (True, "intended"), (False, "as_written"), (True, "as_written")])
def test_hip_vec_frame_stack_wraps_any_image_vec_env(oracle, device, norm, view):
    """HipVecFrameStack = VecFrameStack [+ VecNormAndPermute] as a VecEnvWrapper over an arbitrary VecEnv of {H, W, 1} byte images:
    spaces (stacked_observation.h:54-63, vec_atari.cpp:54-58), observations against the oracle, reward / done / info untouched."""
    from gym_amd.atari import HipVecFrameStack
    n, H, W, k = 4, 84, 84, 4
    env = HipVecFrameStack.make(_ImageVecEnv(n, H, W, device), k, norm_and_permute=norm, view=view)
    assert env.nEnvs() == n and env.observationSpace().size() == ([k, H, W] if norm else [H, W, k]) and env.actionSpace().n == 3
    assert env.observationSpace().type() == (np.float32 if norm else np.int8)                     # make_shape<uint8_t> reports kChar
    stack = np.zeros((n, H, W, k), np.uint8)
    obs = env.reset()
    oracle.framestack_reset(stack, _planes(n, H, W, 0, seed=8)[..., 0])
    permute = oracle.norm_permute_as_written if view == "as_written" else oracle.norm_permute
    check = lambda o: np.array_equal(o.cpu().numpy(), permute(stack)) if norm else np.array_equal(o.cpu().numpy().view(np.uint8), stack)
    assert obs.is_cuda and obs.dtype == (torch.float32 if norm else torch.int8 if view == "as_written" else torch.uint8) and check(obs)
    for t in range(1, 11):
        r = env.step(torch.zeros(n, dtype=torch.int64))
        (oracle.framestack_update_as_written if view == "as_written" else oracle.framestack_update)(stack, _planes(n, H, W, t, seed=8)[..., 0])
        assert check(r.observation), t
        assert r.reward.tolist() == [[float(t)]] * n and r.done.tolist() == [[float((t + e) % 4 == 0)] for e in range(n)] and len(r.info) == n
    with pytest.raises(RuntimeError):
        env.reset(0)
    with pytest.raises(RuntimeError):
        HipVecFrameStack(_ImageVecEnv(n, H, W, device), k, channelOrder="first")


@pytest.mark.parametrize("view", ["intended", "as_written"])
@pytest.mark.parametrize("offset", [1, 2, 3, 4, 8])
def test_nhwc_view_into_a_misaligned_byte_buffer(view, offset):
    """The channels-last view written at an arbitrary BYTE offset (a slice of a larger uint8 tensor): offsets that are not a multiple
    of 4 must take the byte path (the packed 32-bit stores would be misaligned), 4 / 8 the dword path; all equal the aligned result."""
    from gym_amd.atari import AtariPreprocessor
    n = 37
    pp = AtariPreprocessor(n, device=0)
    g = torch.Generator(device="cuda").manual_seed(offset)
    pp.reset(torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda", generator=g))
    for _ in range(3):
        pp.step(torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda", generator=g))
    pp.set_stack_view(view)
    want = pp.stacked_nhwc().view(torch.uint8)
    numel = want.numel()
    big = torch.full((numel + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    base = (-big.data_ptr()) % 16                    # make the buffer start 16-B aligned, then add the offset under test
    out = big[base + offset: base + offset + numel]
    assert out.data_ptr() % 16 == offset
    pp.stacked_nhwc(out.view(want.shape) if view == "intended" else out.view(torch.int8).view(want.shape))
    torch.cuda.synchronize()
    assert torch.equal(out, want.reshape(-1))
    assert (big[:base + offset] == 0xAB).all() and (big[base + offset + numel:] == 0xAB).all()   # nothing written outside

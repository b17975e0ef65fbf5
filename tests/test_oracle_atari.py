"""CPU: hand-derived known answers for the oracle's Atari preprocessing restatement.

OpenCV is not in this image and its version is unpinned in the reference (CMakeLists.txt:28-29), so parity with
cv::cvtColor / cv::resize themselves is UNPINNED; what is pinned here is the published algorithm: the 15-/14-bit
BT.601 fixed-point gray, computeResizeAreaTab's per-axis tables, and area averaging against an independent float64
box-integration written below (+-1 LSB)."""
import numpy as np
import pytest

from tests.test_gpu_atari import counter_frames, structured_frames


def test_gray_known_answers(oracle):
    px = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]], np.uint8)
    g15 = oracle.gray(px, 1)
    assert list(g15) == [0, 255, 76, 150, 29, (10 * 9798 + 20 * 19235 + 30 * 3735 + 16384) >> 15]
    g14 = oracle.gray(px, 0)
    assert list(g14) == [0, 255, 76, 150, 29, (10 * 4899 + 20 * 9617 + 30 * 1868 + 8192) >> 14]
    rng = np.random.default_rng(0)
    rnd = rng.integers(0, 256, (100000, 3), dtype=np.uint8)
    a, b = oracle.gray(rnd, 1).astype(int), oracle.gray(rnd, 0).astype(int)
    assert np.abs(a - b).max() <= 1                                   # the two OpenCV generations differ by <= 1 LSB
    y = np.rint(rnd @ np.array([0.299, 0.587, 0.114])).astype(int)
    assert np.abs(a - y).max() <= 1


def test_area_tables(oracle):
    # vertical 210 -> 84 is an exact 5:2 ratio: rows {0,1,2(half)} then {2(half),3,4}, weights {1,1,.5}/2.5
    si, di, al = oracle.area_table(210, 84)
    assert len(si) == 84 * 3
    assert list(si[:6]) == [0, 1, 2, 2, 3, 4] and list(di[:6]) == [0, 0, 0, 1, 1, 1]
    assert np.allclose(al[:6], [0.4, 0.4, 0.2, 0.2, 0.4, 0.4], rtol=1e-7)
    # horizontal 160 -> 84 (scale 40/21): every destination's weights sum to 1, 2..3 taps each, sources stay in range
    si, di, al = oracle.area_table(160, 84)
    sums = np.bincount(di, weights=al.astype(np.float64), minlength=84)
    assert np.allclose(sums, 1.0, atol=1e-6)
    counts = np.bincount(di, minlength=84)
    assert counts.min() >= 2 and counts.max() == 3 and si.min() == 0 and si.max() == 159
    assert len(si) == 240          # per 21 outputs: 40 source pixels + 20 straddled twice
    # integer ratio degenerates to plain block means
    si, di, al = oracle.area_table(8, 4)
    assert list(si) == list(range(8)) and np.allclose(al, 0.5)


def box_resize_f64(img, dh, dw):
    """Independent float64 area integration of the pixel grid (each source pixel is a unit box)."""
    H, W = img.shape
    def weights(s, d):
        m = np.zeros((d, s))
        scale = s / d
        for i in range(d):
            lo, hi = i * scale, (i + 1) * scale
            for j in range(int(np.floor(lo)), min(int(np.ceil(hi)), s)):
                m[i, j] = max(0.0, min(hi, j + 1) - max(lo, j)) / scale
        return m
    return weights(H, dh) @ img.astype(np.float64) @ weights(W, dw).T


def test_area_resize_against_box_integration(oracle):
    rng = np.random.default_rng(1)
    for img in (rng.integers(0, 256, (210, 160), dtype=np.uint8), np.full((210, 160), 137, np.uint8),
                (np.add.outer(np.arange(210), np.arange(160)) % 256).astype(np.uint8)):
        got = oracle.area_resize(img, 84, 84).astype(np.float64)
        want = box_resize_f64(img, 84, 84)
        assert np.abs(got - want).max() <= 0.5 + 1e-3                 # correct rounding of the exact area mean (+ float noise)
    assert (oracle.area_resize(np.full((210, 160), 137, np.uint8), 84, 84) == 137).all()
    imp = np.zeros((210, 160), np.uint8); imp[2, 0] = 250             # row 2 is split 50/50 between output rows 0 and 1
    out = oracle.area_resize(imp, 84, 84)
    w_x = 1.0 / (160 / 84)
    assert out[0, 0] == round(250 * 0.2 * w_x) and out[1, 0] == round(250 * 0.2 * w_x) and out[2:, :].sum() == 0 and out[:, 1:].sum() == 0


def test_preprocess_is_max_then_gray_then_resize(oracle):
    fr = np.concatenate([structured_frames(), counter_frames(3, 2)])
    out = oracle.atari_preprocess(fr)
    for e in range(fr.shape[0]):
        mx = np.maximum(fr[e, 0], fr[e, 1])
        assert np.array_equal(out[e], oracle.area_resize(oracle.gray(mx), 84, 84))
    assert (out[0] == 0).all() and (out[1] == 255).all() and (out[2] == 200).all()


def test_frame_stack_intended_semantics(oracle):
    """Intended (SB3) semantics: reset -> [0,0,0,f1]; updates shift left.  The reference as written yields [3,4,4,4]
    after frames 1..4 — a documented divergence; the as-written form is a separate view (next test)."""
    n, hw, k = 2, 6, 4
    stack = np.zeros((n, hw, k), np.uint8)
    f = lambda v: np.full((n, hw), v, np.uint8)
    oracle.framestack_reset(stack, f(1))
    assert (stack[0, 0] == [0, 0, 0, 1]).all()
    for v in (2, 3, 4):
        oracle.framestack_update(stack, f(v))
    assert (stack[1, 3] == [1, 2, 3, 4]).all()
    oracle.framestack_update(stack, f(5))
    assert (stack[0, 5] == [2, 3, 4, 5]).all()
    nchw = oracle.norm_permute(stack.reshape(n, 2, 3, k))
    assert nchw.shape == (n, k, 2, 3) and nchw.dtype == np.float32
    assert np.array_equal(nchw[0, :, 1, 2], (np.array([2, 3, 4, 5], np.float32) / np.float32(255)))


@pytest.mark.parametrize("k", [2, 3, 4, 8])
def test_frame_stack_as_written_equals_the_literal_torch_calls(oracle, k):
    """The reference's StackedObservation over a channels-last space IS four torch calls (stacked_observation.h:96-99,
    106-109, 126-130); its header cannot be compiled here (common/utils.h needs OpenCV), so the as-written restatement is
    pinned against those calls issued literally through torch's CPU ops, on the int8 (kChar) buffer the reference
    allocates (spaces/util.h:60-63, stacked_observation.h:208): reset -> zero, y[..., -C:] = x; update -> roll(y, -C, -1),
    then stack(y, x, stackDim = -1) = y[..., -(-1):] = x.  Frames 1,2,3,4 -> [3,4,4,4]."""
    import torch
    n, H, W = 3, 5, 4
    rng = np.random.default_rng(k)
    frames = rng.integers(0, 256, (7, n, H, W, 1), dtype=np.uint8)
    y = torch.zeros((n, H, W, k), dtype=torch.int8)
    x0 = torch.from_numpy(frames[0]).view(torch.int8)
    y[...] = 0
    y[..., -x0.size(-1):] = x0                                   # reset: stack(y, x, x.size(stackDim), false)
    stack = np.zeros((n, H * W, k), np.uint8)
    oracle.framestack_reset(stack, frames[0].reshape(n, H * W))
    assert np.array_equal(y.view(torch.uint8).numpy().reshape(n, H * W, k), stack)
    stack_dim = -1
    for t in range(1, 7):
        x = torch.from_numpy(frames[t]).view(torch.int8)
        y = torch.roll(y, -x.size(stack_dim), stack_dim)          # :127-128
        y[..., -stack_dim:] = x                                   # :129 -> stack(y, x, size = stackDim, false): Slice(-size)
        oracle.framestack_update_as_written(stack, frames[t].reshape(n, H * W))
        assert np.array_equal(y.view(torch.uint8).numpy().reshape(n, H * W, k), stack), t
        # VecNormAndPermute::observation on that int8 tensor (vec_atari.cpp:74-76): permute, then y / 255 — signed bytes
        want = (y.permute(0, 3, 1, 2) / 255).numpy()
        assert want.dtype == np.float32 and np.array_equal(oracle.norm_permute_as_written(stack.reshape(n, H, W, k)), want), t
    if k == 4:
        s = np.zeros((1, 1, 4), np.uint8)
        oracle.framestack_reset(s, np.full((1, 1), 1, np.uint8))
        for v in (2, 3, 4):
            oracle.framestack_update_as_written(s, np.full((1, 1), v, np.uint8))
        assert (s[0, 0] == [3, 4, 4, 4]).all()

"""Split-double state format (include/gymhip.h `state`; gym_amd/csrc/classic_control.hip split_residual/split_join).

A numpy restatement of the two device functions, checked for exactness on the values the format must carry: every
double d with a normal float image round-trips bit-for-bit through (float32 f, int32 k), |k| <= 2^29.  The device
functions themselves are exercised on the GPU by tests/test_gpu_classic_lockstep.py (state_f64 vs the oracle's doubles).
"""
import numpy as np


def split(d):
    f = d.astype(np.float32)
    e = ((f.view(np.uint32) >> 23) & 0xFF).astype(np.int64)
    ok = (e != 0) & (e != 255)
    k = d.view(np.int64) - f.astype(np.float64).view(np.int64)       # distance between the two bit patterns
    k = np.where(ok, k, 0)
    assert np.abs(k).max() <= 2 ** 29
    return f, k.astype(np.int32), ok


def join(f, k):
    return (f.astype(np.float64).view(np.int64) + k.astype(np.int64)).view(np.float64)


def roundtrip(d):
    f, k, ok = split(d)
    back = join(f, k)
    assert np.array_equal(back[ok].view(np.uint64), d[ok].view(np.uint64))
    return ok


def test_random_doubles_roundtrip_exactly():
    rng = np.random.default_rng(0)
    mant = rng.random(1 << 20) + 1.0
    expo = rng.integers(-120, 120, size=mant.size)
    d = np.ldexp(mant, expo) * rng.choice([-1.0, 1.0], size=mant.size)
    assert roundtrip(d).all()


def test_state_like_values_roundtrip_exactly():
    rng = np.random.default_rng(1)
    x = rng.uniform(-0.05, 0.05, 1 << 18).astype(np.float32).astype(np.float64)     # reset draws are floats widened
    v = x + 0.02 * rng.standard_normal(x.size)                                      # then double Euler updates
    assert roundtrip(np.concatenate([x, v, v * 1e-9, v * 1e6])).all()


def test_binade_edges_and_ties():
    vals = []
    for e in (-126, -100, -1, 0, 1, 60, 126):
        p = np.ldexp(1.0, e)
        vals += [p, np.nextafter(p, 0), np.nextafter(p, np.inf), p * (1 + 2.0 ** -24), p * (1 + 2.0 ** -24 + 2.0 ** -52),
                 p * (1 + 2.0 ** -23), p * (2 - 2.0 ** -24), p * (2 - 2.0 ** -25), p * (2 - 2.0 ** -52), p * (1 + 3 * 2.0 ** -25)]
    d = np.array(vals + [-v for v in vals])
    ok = roundtrip(d)
    # values that round (float) below the normal range are outside the format; everything else must be inside
    assert ok[np.abs(d.astype(np.float32)) >= np.float32(2.0 ** -126)].all()


def test_zero_keeps_its_sign_and_tiny_values_are_flagged():
    d = np.array([0.0, -0.0, 1e-40, -1e-300])
    f, k, ok = split(d)
    assert not ok.any() and (k == 0).all()
    back = join(f, k)
    assert np.array_equal(back[:2].view(np.uint64), d[:2].view(np.uint64))          # +-0 exact; nonzero tiny -> error bit 2

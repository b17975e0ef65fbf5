"""The one arithmetic difference between the HIP CartPole step and the reference: cartpole.cpp:39-40 narrows glibc's double
cos/sin of the pole angle to float; the kernel narrows its own double polynomials (pole_sincos, the fdlibm kernels without
argument reduction).  Both are < 1 ulp in double, so the floats can differ only when the true value lies within ~1 double
ulp of a float rounding tie.  This file QUANTIFIES that:

  * CPU: the kernel polynomial's host twin vs glibc over a dense sweep of the whole reachable angle range — mismatches counted
    (measured once over 1.6e10 angles x 2 functions: 0; profiles/NOTES.md, former DESIGN §7) — and over frozen adversarial near-tie angles, where
    mismatches DO occur and their number is pinned;
  * GPU: the device function equals the host twin bit for bit in double, so the CPU counts are the device's counts.
"""
import ctypes as C
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THETA_MAX = 0.2095


def test_dense_sweep_has_no_float_mismatch_against_glibc(oracle):
    r = oracle.sincos_mismatch_sweep(-THETA_MAX, THETA_MAX, 400_000_000)
    assert r["sin_mismatches"] == 0 and r["cos_mismatches"] == 0, r
    assert r["sin_max_ulp"] <= 1.0 and r["cos_max_ulp"] <= 1.0, r           # the doubles themselves never differ by more than 1 ulp


def test_adversarial_near_tie_angles_match_the_frozen_glibc_results(oracle):
    g = np.load(os.path.join(G, "sincos_glibc_ties.npz"), allow_pickle=False)
    th = g["theta"]
    gs, gc = oracle.glibc_sincos_f32(th)
    assert np.array_equal(gs, g["glibc_sin_f32"]) and np.array_equal(gc, g["glibc_cos_f32"])      # this host's glibc == the frozen one
    ps, pc = oracle.pole_sincos(th)
    ms, mc = ps.astype(np.float32) != gs, pc.astype(np.float32) != gc
    assert np.array_equal(np.packbits(ms), g["twin_sin_differs"]) and np.array_equal(np.packbits(mc), g["twin_cos_differs"])
    # where they differ it is by exactly one float ulp, and the doubles by at most one double ulp
    d = np.abs(ps.astype(np.float32)[ms].view(np.int32) - gs[ms].view(np.int32))
    assert ms.sum() > 0 and (d == 1).all()
    assert np.abs(ps - np.sin(th)).max() <= 2 * np.spacing(0.21)


@pytest.mark.gpu
def test_device_pole_sincos_equals_the_host_twin_bit_for_bit(oracle):
    import torch
    from gym_amd import _lib as L
    lib = L.load()
    g = np.load(os.path.join(G, "sincos_glibc_ties.npz"), allow_pickle=False)
    rng = np.random.default_rng(99)
    th = np.concatenate([g["theta"], rng.uniform(-THETA_MAX, THETA_MAX, 1 << 21), [0.0, -0.0, THETA_MAX, -THETA_MAX, 1e-300, 2.0, -7.5]])
    d = torch.from_numpy(th).cuda()
    s, c = torch.empty_like(d), torch.empty_like(d)
    L.check(lib.gymhip_diag_pole_sincos(C.c_void_p(d.data_ptr()), d.numel(), C.c_void_p(s.data_ptr()), C.c_void_p(c.data_ptr()), None), "diag")
    torch.cuda.synchronize()
    inside = np.abs(th) < 0.3
    ps, pc = oracle.pole_sincos(th[inside])
    assert np.array_equal(s.cpu().numpy()[inside].view(np.int64), ps.view(np.int64))
    assert np.array_equal(c.cpu().numpy()[inside].view(np.int64), pc.view(np.int64))
    # beyond 0.3 rad the kernel falls back to the device library (never reached from a step: |theta| <= 0.2095 there)
    assert np.allclose(s.cpu().numpy()[~inside], np.sin(th[~inside]), rtol=1e-15) and np.allclose(c.cpu().numpy()[~inside], np.cos(th[~inside]), rtol=1e-15)
    # and the narrowed values against glibc on the dense part: no mismatch
    dense = slice(g["theta"].size, g["theta"].size + (1 << 21))
    gs, gc = oracle.glibc_sincos_f32(th[dense])
    assert np.array_equal(s.cpu().numpy()[dense].astype(np.float32), gs) and np.array_equal(c.cpu().numpy()[dense].astype(np.float32), gc)

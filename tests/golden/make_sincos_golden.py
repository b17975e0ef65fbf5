#!/usr/bin/env python3
"""Writes tests/golden/sincos_glibc_ties.npz: pole angles chosen so that sin(theta) or cos(theta) lies within a few double
ulps of a float32 rounding tie — the only place where two < 1 ulp double implementations can narrow to different floats
(cartpole.cpp:39-40 narrows glibc's double cos/sin to float) — with glibc's narrowed results as computed in the build
container's image (the same image runs on the GPU box), and the mask of angles where the HIP kernel's polynomial
(its host twin in the checker library) narrows differently.  Run from the repo root: python tests/golden/make_sincos_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
THETA_MAX = 0.2095   # |theta| <= theta_thr = 0.20943951 at step entry (cartpole.h:38)


def adversarial_angles(n_each=4000, seed=12345):
    rng = np.random.default_rng(seed)

    def ties(lo, hi, n):   # exact doubles halfway between two adjacent floats
        f = rng.uniform(lo, hi, n).astype(np.float32)
        return (f.astype(np.float64) + np.nextafter(f, np.float32(np.inf)).astype(np.float64)) / 2

    th_s = np.arcsin(ties(1e-4, 0.2079, n_each)) * rng.choice([-1.0, 1.0], n_each)
    th_c = np.arccos(ties(0.97814, 0.99999, n_each)) * rng.choice([-1.0, 1.0], n_each)

    def neighbours(t):   # the angle and its +-1, +-2 double neighbours
        out, a, b = [t], t, t
        for _ in range(2):
            a, b = np.nextafter(a, np.inf), np.nextafter(b, -np.inf)
            out += [a.copy(), b.copy()]
        return np.concatenate(out)

    th = np.concatenate([neighbours(th_s), neighbours(th_c)])
    return th[np.abs(th) <= THETA_MAX]


def main():
    o = Oracle()
    th = adversarial_angles()
    gs, gc = o.glibc_sincos_f32(th)
    ps, pc = o.pole_sincos(th)
    ms, mc = ps.astype(np.float32) != gs, pc.astype(np.float32) != gc
    np.savez_compressed(os.path.join(OUT, "sincos_glibc_ties.npz"), theta=th, glibc_sin_f32=gs, glibc_cos_f32=gc,
                        twin_sin_differs=np.packbits(ms), twin_cos_differs=np.packbits(mc))
    print(f"{th.size} near-tie angles: kernel polynomial narrows differently from glibc for {int(ms.sum())} sines, {int(mc.sum())} cosines")


if __name__ == "__main__":
    main()

"""Times the Atari preprocess step (4096 envs): RGB frames, palette-index screens with random indices (every LDS table
lookup a potential bank conflict: the worst case) and with blocky screens (large same-colour regions, like real frames)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gym_amd.atari import AtariPreprocessor

n = int(os.environ.get("ATARI_N", 4096))   # ATARI_N / ATARI_RGB_ONLY: sweeps
pp = AtariPreprocessor(n)
pp.set_palette(np.random.default_rng(0).integers(0, 1 << 24, 256, dtype=np.uint32))


def timed(x, bytes_per_env, label, steps=50):
    pp.reset(x)
    for _ in range(5):
        pp.step(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        pp.step(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label:34s} {ms:.4f} ms  {bytes_per_env * n / ms / 1e6:8.1f} GB/s  {n / ms * 1e3:.3e} env-steps/s", flush=True)


rgb = torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda")
timed(rgb, 208656, "rgb frames")
del rgb
if os.environ.get("ATARI_RGB_ONLY"):
    sys.exit(0)
idx = torch.randint(0, 128, (n, 2, 210, 160), dtype=torch.uint8, device="cuda") * 2
timed(idx, 74256, "index screens, random")
blocks = torch.randint(0, 128, (n, 2, 14, 10), dtype=torch.uint8, device="cuda") * 2
blocky = blocks.repeat_interleave(15, dim=2).repeat_interleave(16, dim=3).contiguous()
timed(blocky, 74256, "index screens, 15x16 colour blocks")

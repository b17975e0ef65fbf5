"""RGB Atari preprocess only (4096 envs, 30 steps): a target for rocprofv3 passes over the preprocess kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.atari import AtariPreprocessor

n = 4096
pp = AtariPreprocessor(n)
rgb = torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda")
pp.reset(rgb)
for _ in range(5):
    pp.step(rgb)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    pp.step(rgb)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
print(f"rgb preprocess {ms:.4f} ms  {208656 * n / ms / 1e6:.1f} GB/s  frac {208656 * n / ms / 1e6 / 8000:.3f}", flush=True)

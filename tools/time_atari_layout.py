"""Does the plane layout matter to the preprocess kernel?  Same kernel, same frames: planes into the ring (env-major [N][k][7056]:
concurrently written planes 28,224 B apart) against planes into a dense [N][7056] buffer (gymhip_atari_preprocess)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.atari import AtariPreprocessor

n = int(os.environ.get("ATARI_N", 4096))
pp = AtariPreprocessor(n)
rgb = torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda")
dense = torch.empty((n, 84, 84), dtype=torch.uint8, device="cuda")
pp.reset(rgb)


def timed(fn, label, steps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label:40s} {ms:.4f} ms  {208656 * n / ms / 1e6:8.1f} GB/s", flush=True)


for _ in range(2):
    timed(lambda: pp.step(rgb), "ring [N][4][7056] (stride 28,224 B)")
    timed(lambda: pp.preprocess(rgb, dense), "dense [N][7056]")

"""The Atari path as a pipeline, 4096 envs: preprocess+stack (gymhip_atari_step) then the NCHW view, per step, alternating between
two frame batches — per-kernel and per-step times.  GYMHIP_LIB selects a library variant."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.atari import AtariPreprocessor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
view = sys.argv[2] if len(sys.argv) > 2 else "f32"
pp = AtariPreprocessor(n)
frames = [torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
pp.reset(frames[0])
out = torch.empty((n, 4, 84, 84), dtype=torch.float32 if view == "f32" else torch.bfloat16, device="cuda")
mat = (lambda: pp.stacked_nchw_f32(out)) if view == "f32" else (lambda: pp.stacked_nchw_16(torch.bfloat16, out))
steps = 60
for k in range(10):
    pp.step(frames[k & 1]); mat()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
for k in range(steps):
    ev[k][0].record(); pp.step(frames[k & 1]); ev[k][1].record(); mat(); ev[k][2].record()
torch.cuda.synchronize()
pre = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
m = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
tot = ev[0][0].elapsed_time(ev[-1][2]) / steps
print(f"lib={os.environ.get('GYMHIP_LIB', 'default')} view={view} preprocess {pre:.4f} ms  materialise {m:.4f} ms  per step {tot:.4f} ms  -> {n / tot * 1e3:.3e} env-steps/s", flush=True)

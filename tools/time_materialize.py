"""Times the NCHW materialisation of the frame-stack ring (VecNormAndPermute fused, 4096 envs): f32 and bf16 views.
Launch-shape overrides GYMHIP_MAT_SPLIT / GYMHIP_MAT_GRID are read by the library per call (tools/sweep_materialize.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.atari import AtariPreprocessor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pp = AtariPreprocessor(n)
frames = torch.randint(0, 256, (n, 2, 210, 160, 3), dtype=torch.uint8, device="cuda")
pp.reset(frames)
for _ in range(4):
    pp.step(frames)
del frames
out32 = torch.empty((n, 4, 84, 84), dtype=torch.float32, device="cuda")
out16 = torch.empty((n, 4, 84, 84), dtype=torch.bfloat16, device="cuda")


def timed(fn, nbytes, label, steps=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label} split={os.environ.get('GYMHIP_MAT_SPLIT', 'default')} grid={os.environ.get('GYMHIP_MAT_GRID', 'default')} {ms:.4f} ms {nbytes / ms / 1e6:.1f} GB/s frac={nbytes / ms / 1e6 / 8000:.3f}", flush=True)


timed(lambda: pp.stacked_nchw_f32(out32), n * 28224 * 5, "materialize_nchw_f32 ")
timed(lambda: pp.stacked_nchw_16(torch.bfloat16, out16), n * 28224 * 3, "materialize_nchw_bf16")
out8 = torch.empty((n, 84, 84, 4), dtype=torch.uint8, device="cuda")
timed(lambda: pp.stacked_nhwc(out8), n * 28224 * 2, "materialize_nhwc_u8  ")
pp.set_stack_view("as_written")
out8s = torch.empty((n, 84, 84, 4), dtype=torch.int8, device="cuda")
timed(lambda: pp.stacked_nhwc(out8s), n * 7056 * 6, "materialize_nhwc_as_written")

"""Python owner of a gymhip_atari handle: MaxAndSkip max + WarpFrame + VecFrameStack [+ VecNormAndPermute] on device.

Mirrors the reference's Atari wrapper chain for the part that is not emulation; ALE itself stays on
the host and hands over frames u8 [N][2][210][160][3] in device memory.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .vec_env import VecEnv, VecEnvWrapper


class AtariPreprocessor:
    def __init__(self, n_envs: int, n_stack: int = 4, src_hw=(210, 160), dst_hw=(84, 84), gray_bits: int = 15,
                 clear_on_done: bool = False, device: int = 0):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.GymhipError("no GPU visible: the gymhip engine has no CPU fallback")
        self.device = torch.device("cuda", device)
        c = L.AtariConfig()
        c.struct_size = C.sizeof(L.AtariConfig)
        c.n_envs, c.src_h, c.src_w, c.dst_h, c.dst_w = n_envs, src_hw[0], src_hw[1], dst_hw[0], dst_hw[1]
        c.n_stack, c.gray_bits, c.clear_on_done = n_stack, gray_bits, int(clear_on_done)
        self.cfg = c
        h = C.c_void_p()
        L.check(self.lib.gymhip_atari_create(C.byref(c), device, C.byref(h)), "gymhip_atari_create")
        self.h = h
        self.n, self.k, self.dst_hw, self.src_hw = n_envs, n_stack, dst_hw, src_hw

    def close(self):
        if getattr(self, "h", None):
            self.lib.gymhip_atari_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        # the caller's current stream as a raw hipStream_t; the private fast path saves ~1 us per call at launch-bound sizes
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return C.c_void_p(raw(self.device.index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_palette(self, zrgb256):
        """The emulator's colour table: 256 words 0x00RRGGBB (the reference's rgb_palette, atari_ntsc_rgb_palette.h).
        After this, preprocess/reset/step also accept index screens uint8 [N, 2, H, W] (ale getScreen().getArray())."""
        import numpy as np
        pal = np.ascontiguousarray(zrgb256, dtype=np.uint32)
        if pal.shape != (256,):
            raise ValueError("palette must hold 256 entries")
        L.check(self.lib.gymhip_atari_set_palette(self.h, pal.ctypes.data_as(C.c_void_p)), "gymhip_atari_set_palette")

    def _frames(self, frames: torch.Tensor):
        """-> (device pointer, indexed?).  RGB frames are [N, 2, H, W, 3]; index screens [N, 2, H, W]."""
        rgb = (self.n, 2, self.src_hw[0], self.src_hw[1], 3)
        if frames.dtype != torch.uint8 or tuple(frames.shape) not in (rgb, rgb[:-1]) or not frames.is_contiguous() or frames.device != self.device:
            raise ValueError(f"frames must be a contiguous uint8 {rgb} (RGB) or {rgb[:-1]} (palette index) tensor on {self.device}")
        return C.c_void_p(frames.data_ptr()), frames.dim() == 4

    def preprocess(self, frames: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.n, *self.dst_hw), dtype=torch.uint8, device=self.device)
        ptr, indexed = self._frames(frames)
        fn = self.lib.gymhip_atari_preprocess_indexed if indexed else self.lib.gymhip_atari_preprocess
        L.check(fn(self.h, ptr, C.c_void_p(out.data_ptr()), self._stream()), "gymhip_atari_preprocess")
        return out

    def reset(self, frames: torch.Tensor):
        ptr, indexed = self._frames(frames)
        fn = self.lib.gymhip_atari_reset_indexed if indexed else self.lib.gymhip_atari_reset
        L.check(fn(self.h, ptr, self._stream()), "gymhip_atari_reset")

    def step(self, frames: torch.Tensor, done: torch.Tensor | None = None):
        dp = None
        if done is not None:
            if done.dtype != torch.uint8 or done.numel() != self.n or not done.is_contiguous():
                raise ValueError("done must be a contiguous uint8 [n_envs] tensor")
            dp = C.c_void_p(done.data_ptr())
        ptr, indexed = self._frames(frames)
        fn = self.lib.gymhip_atari_step_indexed if indexed else self.lib.gymhip_atari_step
        L.check(fn(self.h, ptr, dp, self._stream()), "gymhip_atari_step")

    def _planes(self, planes: torch.Tensor):
        if planes.dtype not in (torch.uint8, torch.int8) or planes.numel() != self.n * self.dst_hw[0] * self.dst_hw[1]:
            raise ValueError(f"planes must be a byte tensor of {self.n} x {self.dst_hw[0]} x {self.dst_hw[1]} elements ([N, H, W, 1])")
        return planes.to(self.device).contiguous()

    def push_reset(self, planes: torch.Tensor):
        """VecFrameStack::reset over observations that are ALREADY dst_h x dst_w single-channel images (any inner env's
        [N, H, W, 1] batch, host or device): zero the stack, the planes become the newest frame (gymhip_atari_push_reset)."""
        t = self._planes(planes)
        L.check(self.lib.gymhip_atari_push_reset(self.h, C.c_void_p(t.data_ptr()), self._stream()), "gymhip_atari_push_reset")

    def push(self, planes: torch.Tensor, done: torch.Tensor | None = None):
        """VecFrameStack::stepWait over such observations: shift by one frame, the planes become the newest."""
        dp = None
        if done is not None:
            done = done.reshape(-1).to(self.device, torch.uint8).contiguous()
            if done.numel() != self.n:
                raise ValueError("done must hold n_envs flags")
            dp = C.c_void_p(done.data_ptr())
        t = self._planes(planes)
        L.check(self.lib.gymhip_atari_push(self.h, C.c_void_p(t.data_ptr()), dp, self._stream()), "gymhip_atari_push")

    def head(self) -> int:
        ring, head = C.c_void_p(), C.c_int32()
        L.check(self.lib.gymhip_atari_stack_buffer(self.h, C.byref(ring), C.byref(head)), "gymhip_atari_stack_buffer")
        return head.value

    def stacked_nhwc(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """VecFrameStack observation [N, 84, 84, k]: uint8, oldest frame in channel 0 — or, in the as-written view, int8 (the
        reference's kChar buffer; same bytes) with the reference's channel contents."""
        if out is None:
            out = torch.empty((self.n, *self.dst_hw, self.k), dtype=torch.int8 if getattr(self, "view", "intended") == "as_written" else torch.uint8,
                              device=self.device)
        L.check(self.lib.gymhip_atari_materialize_nhwc(self.h, C.c_void_p(out.data_ptr()), self._stream()), "gymhip_atari_materialize_nhwc")
        return out

    def set_stack_view(self, view: str):
        """"intended" (default): the shift semantics StackedObservation means to implement.  "as_written": what the reference's
        code computes today, bug for bug — channel 0 = the previous call's channel 1, channels 1.. = the newest frame
        (stacked_observation.h:96-99,126-130), held in an int8 buffer (spaces/util.h:60-63) so that VecNormAndPermute's / 255
        (vec_atari.cpp:74-76) divides signed bytes.  The ring is shared; the view can be switched at any time."""
        if view not in ("intended", "as_written"):
            raise ValueError("view must be 'intended' or 'as_written'")
        L.check(self.lib.gymhip_atari_set_stack_view(self.h, L.STACK_AS_WRITTEN if view == "as_written" else L.STACK_INTENDED), "gymhip_atari_set_stack_view")
        self.view = view

    def stacked_nchw_f32(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """VecNormAndPermute(VecFrameStack(...)) observation: float32 [N, k, 84, 84] in [0, 1]."""
        if out is None:
            out = torch.empty((self.n, self.k, *self.dst_hw), dtype=torch.float32, device=self.device)
        L.check(self.lib.gymhip_atari_materialize_nchw_f32(self.h, C.c_void_p(out.data_ptr()), self._stream()), "gymhip_atari_materialize_nchw_f32")
        return out

    def stacked_nchw_16(self, dtype: torch.dtype = torch.bfloat16, out: torch.Tensor | None = None) -> torch.Tensor:
        """The same observation rounded once to float16 / bfloat16 [N, k, 84, 84] for half-precision policies."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("dtype must be torch.float16 or torch.bfloat16")
        if out is None:
            out = torch.empty((self.n, self.k, *self.dst_hw), dtype=dtype, device=self.device)
        if out.dtype != dtype or not out.is_contiguous() or out.numel() != self.n * self.k * self.dst_hw[0] * self.dst_hw[1]:
            raise ValueError("out must be a contiguous [N, k, H, W] tensor of the requested dtype")
        code = L.OUT_F16 if dtype == torch.float16 else L.OUT_BF16
        L.check(self.lib.gymhip_atari_materialize_nchw_16(self.h, C.c_void_p(out.data_ptr()), code, self._stream()), "gymhip_atari_materialize_nchw_16")
        return out


class HipVecFrameStack(VecEnvWrapper):
    """gym::VecFrameStack<false> (wrappers/vec_env/vec_frame_stack.cpp:10-32 over common/stacked_observation.h:96-130) on its own
    — and, with norm_and_permute, the pair VecNormAndPermute(VecFrameStack(venv)) (vec_atari.cpp:70-81) as one object — over ANY
    VecEnv whose observation space is a channels-last single-channel byte image {H, W, 1} (WarpFrame's layout).  The inner batch
    [N, H, W, 1] may be a host or a device tensor; the stack is the device ring (one plane copy per step), the returned
    observation a device tensor overwritten by the next step: uint8 [N, H, W, nStack] (view "as_written": int8 with the
    reference's channel contents), or float32 [N, nStack, H, W] = x / 255.  reward / done / info pass through.  C++ twin:
    gym/hip_vec_frame_stack.h.  Channels-first spaces, a channelOrder other than "last" and reset(index) are rejected — the
    reference corrupts memory, always throws, and smears one env's frame over every stack there."""

    def __init__(self, venv: VecEnv, nStack: int = 4, channelOrder: str | None = None, norm_and_permute: bool = False, view: str = "intended",
                 clear_on_done: bool = False, device: int = 0):
        from . import spaces as S
        import numpy as np
        if channelOrder not in (None, "last"):
            raise RuntimeError("HipVecFrameStack: only channels-last image spaces are supported")
        shape = tuple(venv.observationSpace().size())
        if len(shape) != 3 or shape[2] != 1:
            raise RuntimeError("HipVecFrameStack: the inner observation space must be a channels-last single-channel image {H, W, 1}")
        H, W = int(shape[0]), int(shape[1])
        inner = venv.observationSpace()
        rng = inner.getRange()[0] if isinstance(inner, S.Box) else S.Range(0, 255)
        space = S.makeBoxSpace(0.0, 1.0, [nStack, H, W]) if norm_and_permute else S.makeBoxSpace(rng.low, rng.high, [H, W, nStack], dtype=np.uint8)
        super().__init__(venv, space)
        self.pp = AtariPreprocessor(venv.nEnvs(), n_stack=nStack, src_hw=(H, W), dst_hw=(H, W), clear_on_done=clear_on_done, device=device)
        if view != "intended":
            self.pp.set_stack_view(view)
        self.norm_and_permute, self.clear_on_done = norm_and_permute, clear_on_done
        self._out = None

    @staticmethod
    def make(venv, nStack=4, channelOrder=None, **kw):   # vec_frame_stack.cpp:50-54
        return HipVecFrameStack(venv, nStack, channelOrder, **kw)

    def _observation(self):
        self._out = self.pp.stacked_nchw_f32(self._out) if self.norm_and_permute else self.pp.stacked_nhwc(self._out)
        return self._out

    def reset(self, index=None):
        if index is not None:
            raise RuntimeError("HipVecFrameStack: reset(index) is not supported (the stack advances for all envs together)")
        self.pp.push_reset(self.venv.reset())
        return self._observation()

    def stepWait(self):
        r = self.venv.stepWait()
        self.pp.push(r.observation, r.done if self.clear_on_done else None)
        return r._replace(observation=self._observation())


def host_env_post_step(reward, done, lives, state: dict, clip_reward: bool = True):
    """Fused Monitor -> EpisodicLife -> ClipReward lanes for host-emulated envs (gymhip_host_env_post_step).
    `state`: dict(lives i32, was_real_done u8, ret f32, len f32) device tensors, updated in place."""
    lib = L.load()
    n = reward.numel()
    dev = reward.device
    out = dict(reward=torch.empty(n, dtype=torch.float32, device=dev), done=torch.empty(n, dtype=torch.uint8, device=dev),
               has_ep=torch.empty(n, dtype=torch.uint8, device=dev), ep_r=torch.empty(n, dtype=torch.float32, device=dev),
               ep_l=torch.empty(n, dtype=torch.float32, device=dev))
    p = lambda t: C.c_void_p(t.data_ptr())
    L.check(lib.gymhip_host_env_post_step(n, p(reward), p(done), p(lives), p(state["lives"]), p(state["was_real_done"]), p(state["ret"]),
                                          p(state["len"]), int(clip_reward), p(out["reward"]), p(out["done"]), p(out["has_ep"]),
                                          p(out["ep_r"]), p(out["ep_l"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
            "gymhip_host_env_post_step")
    return out

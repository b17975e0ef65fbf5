"""Host-side mirror of the reference's spaces (spaces/space.h:20-104, box.h:15-54, box.tpp:7-18, discrete.h:14-31,
discrete.cpp:15-21, util.h:18-97, common/utils.h:82-122).  Only what the hot path's VecEnv interface exposes: Box / Discrete
with size(), type(), getRange(), sample(), clone(), as_().  Checked against what the reference's own spaces code reports
(tests/golden/ref_spaces.json; tests/test_ref_pin.py)."""
from __future__ import annotations

import copy
from dataclasses import dataclass

import numpy as np

FLT_MAX = float(np.finfo(np.float32).max)
FLT_MIN = float(np.finfo(np.float32).tiny)   # numeric_limits<float>::min(): smallest POSITIVE float (box.h:31-34 quirk)


@dataclass
class Range:
    low: float
    high: float


class Space:
    """spaces/space.h:20-104.  `sample()` draws from the space's own generator, which the reference seeds from
    std::random_device (space.h:64-67) — i.e. it is NOT reproducible there either; lockstep runs supply an action tape."""

    def __init__(self, shape, dtype):
        self._shape = [int(s) for s in shape]
        self._dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()
        self._name = "observation"

    def size(self):
        return list(self._shape)

    def type(self):
        return self._dtype

    def keys(self):
        return [self._name]

    def as_(self, cls):
        """space->as<T>() (space.h:72-74): dynamic_cast, None on mismatch."""
        return self if isinstance(self, cls) else None

    def clone(self):
        return copy.deepcopy(self)


# util.h:58-102 make_shape<T>: the dtype a space reports.  uint8_t maps to c10::kChar (int8) in the reference — kept.
_REPORTED_DTYPE = {np.dtype(np.uint8): np.dtype(np.int8)}


def make_shape_dtype(dtype):
    d = np.dtype(dtype)
    return _REPORTED_DTYPE.get(d, d)


class Box(Space):
    def __init__(self, low=None, high=None, shape=(1,), dtype=np.float32):
        super().__init__(shape, make_shape_dtype(dtype))
        self._value_dtype = np.dtype(dtype)
        vectorised = low is not None and np.ndim(low) > 0
        if vectorised and len(self._shape) != 1:
            # box.tpp:7-18: a vector of ranges needs a shape with exactly one dim
            raise RuntimeError("a zero dim shape is not allowed" if len(self._shape) == 0 else
                               "initializing a Box with a vectorized range requires a shape with only one dim")
        if low is None and high is None:
            # Box(Shape) ctor: one range {numeric_limits<T>::min(), max()} (box.h:31-34)
            info = np.finfo(dtype) if np.issubdtype(np.dtype(dtype), np.floating) else np.iinfo(dtype)
            lo = float(info.tiny) if np.issubdtype(np.dtype(dtype), np.floating) else int(info.min)
            self._ranges = [Range(lo, info.max)]
        else:
            low_a, high_a = np.atleast_1d(np.asarray(low, np.float64)), np.atleast_1d(np.asarray(high, np.float64))
            if low_a.shape != high_a.shape:
                raise ValueError("Box: low and high must have the same length")   # box.tpp:10-17
            self._ranges = [Range(float(l), float(h)) for l, h in zip(low_a, high_a)]

    def getRange(self):
        return list(self._ranges)

    def sample(self):
        lo = np.array([r.low for r in self._ranges], np.float64)
        hi = np.array([r.high for r in self._ranges], np.float64)
        return self._rng.uniform(np.maximum(lo, -1e30), np.minimum(hi, 1e30)).astype(self._value_dtype)

    def __repr__(self):
        return f"Box(shape={self._shape}, dtype={self._dtype}, ranges={self._ranges})"


class Discrete(Space):
    def __init__(self, n: int):
        if n <= 0:
            raise ValueError("discrete space only accepts dim > 0")   # discrete.cpp:19-20 (std::invalid_argument)
        super().__init__((int(n),), np.int32)                          # discrete.cpp:16: the shape is {n}
        self.n = int(n)

    def sample(self):
        return int(self._rng.integers(0, self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


def makeBoxSpace(*args, dtype=np.float32):
    """common/utils.h:82-117 overloads: (length) | (shape list) | (lows, highs, length) | (low, high, shape list)."""
    if len(args) == 1:
        shape = args[0] if isinstance(args[0], (list, tuple)) else (args[0],)
        return Box(shape=shape, dtype=dtype)
    if len(args) == 3:
        low, high, shp = args
        shape = shp if isinstance(shp, (list, tuple)) else (shp,)
        return Box(low, high, shape, dtype)
    raise TypeError("makeBoxSpace: unsupported overload")


def makeDiscreteSpace(n: int):
    return Discrete(n)


def cartpole_spaces():
    """classic_control/cartpole.cpp:9-17: Box<float> +-{2*2.4, FLT_MAX, 2*theta_thr, FLT_MAX}, Discrete(2)."""
    theta2 = float(np.float32(np.float32(12 * 2 * np.pi / 360) * np.float32(2)))
    high = [float(np.float32(2.4) * np.float32(2)), FLT_MAX, theta2, FLT_MAX]
    return makeBoxSpace([-h for h in high], high, 4), makeDiscreteSpace(2)


def mountaincar_spaces():
    """classic_control/mountain_car.cpp:11-12: makeBoxSpace<float>(2) (default range), Discrete(3)."""
    return makeBoxSpace(2), makeDiscreteSpace(3)

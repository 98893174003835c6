"""gym.spaces when gym (or gymnasium) is installed, else a minimal stand-in with the same attributes
(reference env.py:35 imports gym.spaces; neither package exists in this image)."""
import numpy as np

try:                                    # pragma: no cover - depends on the environment
    from gym.spaces import Box, Discrete, Tuple  # type: ignore
    HAVE_GYM = "gym"
except Exception:                       # noqa: BLE001
    try:                                # pragma: no cover
        from gymnasium.spaces import Box, Discrete, Tuple  # type: ignore
        HAVE_GYM = "gymnasium"
    except Exception:                   # noqa: BLE001
        HAVE_GYM = None

        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.shape = tuple(shape) if shape is not None else np.shape(low)
                self.dtype = np.dtype(dtype)
                self.low = np.full(self.shape, low, dtype=self.dtype)
                self.high = np.full(self.shape, high, dtype=self.dtype)

            def sample(self):
                lo = np.where(np.isfinite(self.low), self.low, -1e6)
                hi = np.where(np.isfinite(self.high), self.high, 1e6)
                return np.random.uniform(lo, hi).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        class Discrete:
            def __init__(self, n):
                self.n = int(n)
                self.shape = ()
                self.dtype = np.dtype(np.int64)

            def sample(self):
                return int(np.random.randint(self.n))

            def contains(self, x):
                return 0 <= int(x) < self.n

            def __repr__(self):
                return f"Discrete({self.n})"

        class Tuple:
            def __init__(self, spaces):
                self.spaces = tuple(spaces)

            def sample(self):
                return tuple(s.sample() for s in self.spaces)

            def contains(self, x):
                return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

            def __len__(self):
                return len(self.spaces)

            def __getitem__(self, i):
                return self.spaces[i]

            def __repr__(self):
                return "Tuple(" + ", ".join(map(repr, self.spaces)) + ")"

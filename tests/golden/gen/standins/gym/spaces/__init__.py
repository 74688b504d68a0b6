import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)


class Discrete(Space):
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        self.spaces = dict(spaces or {}, **kw)

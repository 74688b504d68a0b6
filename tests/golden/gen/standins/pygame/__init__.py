"""Stand-in for ``pygame`` -- FIXTURE-GENERATION INFRASTRUCTURE ONLY.

pygame 2.1.2 (reference ``requirements.txt:5``) is not installed in the build
container and cannot be fetched.  The reference env only needs a sliver of it
for *state-affecting* arithmetic (``Rect`` integer semantics,
``transform.rotate`` bounding box, the clock).  This module restates exactly that
sliver so that the UNMODIFIED reference ``Game`` can
be executed here to emit golden vectors.  Everything at this boundary is
"parity unpinned": the semantics below are recalled from pygame 2.1.2's
``rect.c`` / ``transform.c`` and cannot be checked against the real library
offline.  Nothing in the product imports this module.
"""
import math
import types

import numpy as np


def _i(v):
    """pygame's pg_IntFromObj: C cast of a Python number -> truncation toward 0."""
    return int(v)


class Rect:
    __slots__ = ("x", "y", "w", "h")

    def __init__(self, *args):
        if len(args) == 1:
            a = args[0]
            if isinstance(a, Rect):
                args = (a.x, a.y, a.w, a.h)
            elif len(a) == 4:
                args = tuple(a)
            else:
                args = (a[0][0], a[0][1], a[1][0], a[1][1])
        elif len(args) == 2:
            args = (args[0][0], args[0][1], args[1][0], args[1][1])
        self.x, self.y, self.w, self.h = (_i(v) for v in args)

    # --- scalar attributes -------------------------------------------------
    @property
    def width(self):
        return self.w

    @width.setter
    def width(self, v):
        self.w = _i(v)

    @property
    def height(self):
        return self.h

    @height.setter
    def height(self, v):
        self.h = _i(v)

    @property
    def left(self):
        return self.x

    @left.setter
    def left(self, v):
        self.x = _i(v)

    @property
    def top(self):
        return self.y

    @top.setter
    def top(self, v):
        self.y = _i(v)

    @property
    def right(self):
        return self.x + self.w

    @right.setter
    def right(self, v):
        self.x = _i(v) - self.w

    @property
    def bottom(self):
        return self.y + self.h

    @bottom.setter
    def bottom(self, v):
        self.y = _i(v) - self.h

    @property
    def centerx(self):
        return self.x + (self.w >> 1)

    @property
    def centery(self):
        return self.y + (self.h >> 1)

    @property
    def center(self):
        return (self.x + (self.w >> 1), self.y + (self.h >> 1))

    @center.setter
    def center(self, v):
        self.x = _i(v[0]) - (self.w >> 1)
        self.y = _i(v[1]) - (self.h >> 1)

    @property
    def size(self):
        return (self.w, self.h)

    @property
    def topleft(self):
        return (self.x, self.y)

    @property
    def topright(self):
        return (self.x + self.w, self.y)

    @property
    def bottomleft(self):
        return (self.x, self.y + self.h)

    @property
    def bottomright(self):
        return (self.x + self.w, self.y + self.h)

    @property
    def midtop(self):
        return (self.x + (self.w >> 1), self.y)

    @property
    def midbottom(self):
        return (self.x + (self.w >> 1), self.y + self.h)

    @property
    def midleft(self):
        return (self.x, self.y + (self.h >> 1))

    @property
    def midright(self):
        return (self.x + self.w, self.y + (self.h >> 1))

    # --- methods -----------------------------------------------------------
    def copy(self):
        return Rect(self.x, self.y, self.w, self.h)

    def move_ip(self, *args):
        if len(args) == 1:
            dx, dy = args[0][0], args[0][1]
        else:
            dx, dy = args
        self.x += _i(dx)
        self.y += _i(dy)

    def colliderect(self, o):
        if self.w == 0 or self.h == 0 or o.w == 0 or o.h == 0:
            return False
        return (self.x < o.x + o.w and self.y < o.y + o.h and
                self.x + self.w > o.x and self.y + self.h > o.y)

    def collidelist(self, rects):
        for i, r in enumerate(rects):
            if self.colliderect(r):
                return i
        return -1

    def collidepoint(self, *args):
        if len(args) == 1:
            px, py = args[0][0], args[0][1]
        else:
            px, py = args
        return self.x <= px < self.x + self.w and self.y <= py < self.y + self.h

    def __iter__(self):
        return iter((self.x, self.y, self.w, self.h))

    def __repr__(self):
        return "<rect(%d, %d, %d, %d)>" % (self.x, self.y, self.w, self.h)


class Surface:
    def __init__(self, size=(64, 64)):
        self._w = _i(size[0])
        self._h = _i(size[1])

    def get_width(self):
        return self._w

    def get_height(self):
        return self._h

    def get_size(self):
        return (self._w, self._h)

    def get_rect(self, **kw):
        r = Rect(0, 0, self._w, self._h)
        for k, v in kw.items():
            setattr(r, k, v)
        return r

    def blit(self, *a, **k):
        return None

    def fill(self, *a, **k):
        return None


def _rotate_size(w, h, angle):
    """transform.c surf_rotate: `float angle`; exact quarter turns swap w/h."""
    a = float(np.float32(angle))
    if math.fmod(a, 90.0) == 0.0:
        q = int(a / 90.0) % 4
        return (h, w) if (q % 2) else (w, h)
    r = a * .01745329251994329
    s, c = math.sin(r), math.cos(r)
    cx, cy, sx, sy = c * w, c * h, s * w, s * h
    nx = int(max(abs(cx + sy), abs(cx - sy), abs(-cx + sy), abs(-cx - sy)))
    ny = int(max(abs(sx + cy), abs(sx - cy), abs(-sx + cy), abs(-sx - cy)))
    return nx, ny


transform = types.SimpleNamespace(
    scale=lambda img, size: Surface((size[0], size[1])),
    rotate=lambda img, angle: Surface(_rotate_size(img.get_width(), img.get_height(), angle)),
)

image = types.SimpleNamespace(load=lambda path: Surface((64, 64)))


class _Font:
    def render(self, *a, **k):
        return Surface((1, 1))


font = types.SimpleNamespace(init=lambda: None, SysFont=lambda *a, **k: _Font())
display = types.SimpleNamespace(set_mode=lambda size, *a, **k: Surface(size),
                                set_caption=lambda *a, **k: None,
                                update=lambda *a, **k: None)
event = types.SimpleNamespace(get=lambda: [])
joystick = types.SimpleNamespace(init=lambda: None, Joystick=lambda i: None)
draw = types.SimpleNamespace(circle=lambda *a, **k: None, line=lambda *a, **k: None,
                             lines=lambda *a, **k: None, rect=lambda *a, **k: None,
                             polygon=lambda *a, **k: None)


class _Clock:
    def tick(self, framerate=0):
        return 1

    def get_fps(self):
        return 0.0


# Deterministic clock: the generator script sets
# `time._ticks_fn` to "frame counter of the running env"; default is a counter
# that nobody advances.
def _default_ticks():
    return 0


time = types.SimpleNamespace(Clock=_Clock, get_ticks=lambda: time._ticks_fn(), _ticks_fn=_default_ticks)


def init():
    return (0, 0)


def quit():
    return None


QUIT = 256
KEYDOWN = 768
KEYUP = 769
K_LEFT, K_RIGHT, K_UP, K_DOWN = 1073741904, 1073741903, 1073741906, 1073741905
JOYAXISMOTION = 1536

"""Reset-time scenario generation on the host (``ftl_generate_scenarios``, include/ftl.h).

Replaces the scenario part of the reference's ``Game.reset()`` (ENV:434-543: robots, walls + rocks, finish point, D*
route, bears, initial leader trajectory) with a multi-threaded C++ generator whose draws come from a bit-compatible twin
of CPython's ``random``: ``generate_scenarios(cfg, [s])`` is the scenario of ``game.seed(s); game.reset()``, except for
the choice among equal-cost routes (the reference's depends on object ids)."""
import ctypes as C

import numpy as np
import torch

from . import _lib, abi
from .config import GameConfig

USABLE_MASK = abi.SCEN_DONE_AT_RESET | abi.SCEN_ROUTE_OVERFLOW | abi.SCEN_TRAJ_OVERFLOW | abi.SCEN_REF_RAISES


def scen_params(cfg: GameConfig) -> abi.ScenParams:
    """The generator's view of ``Game(**kwargs)`` (ENV:45-105, 283-357)."""
    kw, c = cfg.kwargs, cfg.c
    if kw["path_finding_algorythm"] not in ("dstar", "astar"):
        raise ValueError("path_finding_algorythm {} not in list:{}".format(kw["path_finding_algorythm"], ["astar", "dstar"]))
    fixed = None
    if kw["trajectory"] is not None:             # ENV:229, 469-470: the caller's way-points on every reset, no finish point, no planner
        fixed = np.ascontiguousarray(np.asarray(kw["trajectory"], np.float64).reshape(-1, 2))
    elif not kw["add_obstacles"] and kw["path_finding_algorythm"] == "dstar":
        # generate_trajectory_dstar reads the two bridge walls (ENV:1501) that _create_obstacles, skipped at ENV:464-465, never made:
        # the reference's reset() raises exactly this (so do its ids Test-Game-Neat-v0 and Test-Cont-Env-Auto-Follow-no-obstacles-v0)
        raise AttributeError("'Game' object has no attribute 'obstacles1'")
    ptm = cfg.pixels_to_meter
    sp = abi.ScenParams()
    sp.width, sp.height = int(kw["game_width"]), int(kw["game_height"])
    sp.step_grid = int(kw["step_grid"])
    sp.add_obstacles = int(bool(kw["add_obstacles"]))
    sp.obstacle_number = int(kw["obstacle_number"]) if kw["add_obstacles"] else 0      # ENV:323-324
    sp.add_bear, sp.bear_number = int(bool(kw["add_bear"])), int(kw["bear_number"])
    sp.bear_behind = int(bool(kw["bear_behind"]))
    sp.multiple_end_points = int(bool(kw["multiple_end_points"]))
    sp.path_finding_iterations = int(kw["path_finding_iterations"])
    sp.bridge_gap, sp.bridge_width = int(kw["bridge_size"][0]), int(kw["bridge_size"][1])
    sp.trajectory_saving_period = c.trajectory_saving_period
    sp.planner = 1 if kw["path_finding_algorythm"] == "astar" else 0
    if fixed is not None:
        sp.planner, sp.fixed_route, sp.fixed_route_len = 2, fixed.ctypes.data, len(fixed)
        sp._keep_alive = fixed
    sp.min_distance, sp.max_distance = c.min_distance, c.max_distance
    sp.leader_pos_epsilon, sp.leader_margin = float(kw["leader_pos_epsilon"]), float(kw["leader_margin"])
    sp.leader_w, sp.leader_h = kw["leader_size"][0] * ptm, kw["leader_size"][1] * ptm    # ENV:352-353
    sp.leader_max_speed = c.leader.max_speed
    return sp


def generate_scenarios(cfg: GameConfig, seeds, n_threads=0):
    """Scenarios of the python seeds ``seeds`` as host arrays (the layout of ``ftl_scenarios``) + ``status`` bits
    (``abi.SCEN_*``) + ``usable`` (route found, the reference's reset() neither raises nor ends the episode, nothing
    truncated)."""
    lib = _lib.load()
    c = cfg.c
    seeds = np.ascontiguousarray(np.asarray(seeds, np.int64).reshape(-1))
    n, R = len(seeds), cfg.n_robots
    out = dict(static_rects=np.zeros((n, c.n_static, 4), np.int32), robot_pos=np.zeros((n, R, 2), np.float32),
               robot_dir=np.zeros((n, R), np.float64), robot_rect=np.zeros((n, R, 4), np.int32),
               route=np.zeros((n, c.route_cap, 2), np.float64), route_len=np.zeros(n, np.int32),
               init_traj=np.zeros((n, c.init_traj_cap, 2), np.float32), init_traj_len=np.zeros(n, np.int32))
    status = np.zeros(n, np.uint8)
    s = abi.Scenarios()
    s.n_scenarios = n
    for k, v in out.items():
        setattr(s, k, v.ctypes.data)
    sp = scen_params(cfg)
    _lib.check(lib.ftl_generate_scenarios(C.byref(c), C.byref(sp), seeds.ctypes.data, n, int(n_threads), C.byref(s),
                                          status.ctypes.data), lib)
    out["seed"] = seeds
    out["status"] = status
    out["usable"] = ((status & abi.SCEN_FOUND) != 0) & ((status & USABLE_MASK) == 0)
    return out


class ScenarioRing:
    """Fresh worlds for a running batch: a device pool of two halves; the envs' auto-reset draws from one half while generator threads
    build the next scenarios on the host (``ftl_generate_scenarios``) and an asynchronous copy fills the other half; ``poll`` moves the
    reset window (``ftl_set_reset_window``) at a step boundary.  The reference draws a new world on EVERY reset() (ENV:461-492); here a
    world is drawn for every pool entry of every half, i.e. one per ``resets per half / half size`` episodes -- the generator's rate
    against the batch's reset rate decides that ratio (bench.py reports both).

    A half is only overwritten after ``horizon`` steps have passed since the window left it: by then every episode that started on it
    has ended (an episode lasts at most max_steps / frames_per_step + 1 steps), so no running env still reads its rects or route.

    ``seeds`` = iterator of python seeds; unusable scenarios (route not found, ...) are skipped, so a half may take more than ``half``
    seeds.  ``n_threads`` generator threads run inside one background python thread (the C call releases the GIL)."""

    def __init__(self, cfg: GameConfig, half, device, seeds, n_threads=0, chunk=None, record=False):
        import itertools
        import threading
        from .vec_game import ScenarioPool
        self.cfg, self.half, self.device = cfg, int(half), device
        self.pool = ScenarioPool.empty(cfg, 2 * self.half, device)
        self._seeds = iter(seeds)
        # one generator call per half as a rule (87 % of the seeds give a usable world): between calls the thread needs the interpreter lock,
        # which the thread that launches the steps gives up a few hundred times a second only
        self._chunk = chunk or max(256, int(self.half * 1.25) + 16)
        self._free = []               # pinned host buffer sets whose copy has landed (reused: pinning 80 MB holds the interpreter lock for tens of ms)
        self._n_threads = n_threads
        self._take = itertools.islice
        c = cfg.c
        fps = c.rand_fps_lo if c.rand_fps_hi > 0 else c.frames_per_step
        self.horizon = c.max_steps // max(fps, 1) + 2
        self._stream = torch.cuda.Stream(device=device)
        self._ready = None            # host arrays of the next half (pinned), produced by the worker
        self._lock = threading.Lock()
        self._stop = False
        self._pending = None          # (half index, cuda event) of a copy in flight
        self.active = 0
        self._left_at = {0: None, 1: -10 ** 9}   # step at which the window left each half (None: it is the active one)
        self.generated = 0            # scenarios generated so far (usable ones)
        self.swaps = 0
        self.history = [] if record else None     # (step, half index, host arrays) of every half that went live (tests)
        first = self._build_half()
        if record:
            self.history.append((0, 0, first))
        self.pool.write(0, first, self._stream)
        self._stream.synchronize()
        self._worker = threading.Thread(target=self._work, daemon=True)
        self._worker.start()

    def _build_half(self):
        keys = ("static_rects", "robot_pos", "robot_dir", "robot_rect", "route", "route_len", "init_traj", "init_traj_len")
        parts, have = [], 0
        while have < self.half:
            if self._stop:
                raise StopIteration("closed")
            seeds = list(self._take(self._seeds, self._chunk))
            if not seeds:
                raise StopIteration("the seed iterator of a ScenarioRing ran dry")
            g = generate_scenarios(self.cfg, seeds, self._n_threads)
            keep = np.nonzero(g["usable"])[0][:self.half - have]
            parts.append({k: g[k][keep] for k in keys})
            have += len(keep)
        with self._lock:
            host = self._free.pop() if self._free else None
        if host is None:
            host = {k: torch.empty((self.half,) + parts[0][k].shape[1:], dtype=torch.from_numpy(parts[0][k][:0]).dtype).pin_memory() for k in keys}
        for k in keys:                     # (large same-dtype copies: numpy releases the interpreter lock while they run)
            dst, at = host[k].numpy(), 0
            for p in parts:
                dst[at:at + len(p[k])] = p[k]
                at += len(p[k])
        self.generated += self.half
        return host

    def _work(self):
        import time
        while not self._stop:
            with self._lock:
                need = self._ready is None
            if need:
                try:
                    host = self._build_half()
                except StopIteration:
                    return
                with self._lock:
                    self._ready = host
            else:
                time.sleep(0.001)

    def _stride(self, env):
        """A step for the auto-reset's walk through a half that is coprime to its size (every entry gets visited) and close to n_envs
        (the default walk of a plain pool)."""
        import math
        s = max(env.n % self.half, 1)
        while math.gcd(s, self.half) != 1:
            s += 1
        return s

    def attach(self, env):
        """Load the ring's pool into ``env`` with the reset window on the first half."""
        env.load_scenarios(self.pool)
        env.set_reset_window(0, self.half, self._stride(env))

    def poll(self, env, step):
        """Call between steps with the number of steps taken so far.  Starts the copy of a finished half into the inactive half once
        that half is free, and moves the reset window once the copy has landed.  Returns True when the window moved."""
        other = 1 - self.active
        if self._pending is not None:
            h, ev = self._pending
            if ev.query():
                self._pending = None
                self._left_at[self.active] = step
                self.active = h
                self._left_at[h] = None
                env.set_reset_window(h * self.half, self.half, self._stride(env))
                self.swaps += 1
                if self.history is not None:
                    self.history.append((step, h, self._held))
                return True
            return False
        left = self._left_at[other]
        if left is not None and step - left >= self.horizon:
            with self._lock:
                host, self._ready = self._ready, None
            if host is not None:
                if self.history is None and getattr(self, "_held", None) is not None:
                    with self._lock:
                        self._free.append(self._held)      # its copy landed a window move ago
                self._held = host        # keep the pinned buffers alive until the copy has landed
                # `step` counts the steps the HOST has queued; the device may be many steps behind.  The copy must not overtake them:
                # the side stream waits for everything queued on the batch's stream so far (steps that may still read this half).
                self._stream.wait_stream(torch.cuda.current_stream(self.device))
                for st in getattr(env, "streams", ()):           # (a PipelinedVecGame steps its parts on streams of their own)
                    self._stream.wait_stream(st)
                self.pool.write(other * self.half, host, self._stream)
                ev = torch.cuda.Event()
                ev.record(self._stream)
                self._pending = (other, ev)
        return False

    def close(self):
        """Stop the generator thread and wait for it: a daemon thread killed at interpreter exit inside ftl_generate_scenarios would take
        the process down (its std::threads are still joinable)."""
        self._stop = True
        if self._worker.is_alive():
            self._worker.join(timeout=60)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

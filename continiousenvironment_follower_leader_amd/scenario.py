"""Reset-time scenario generation on the host (``ftl_generate_scenarios``, include/ftl.h).

Replaces the scenario part of the reference's ``Game.reset()`` (ENV:434-543: robots, walls + rocks, finish point, D*
route, bears, initial leader trajectory) with a multi-threaded C++ generator whose draws come from a bit-compatible twin
of CPython's ``random``: ``generate_scenarios(cfg, [s])`` is the scenario of ``game.seed(s); game.reset()``, except for
the choice among equal-cost routes (the reference's depends on object ids)."""
import ctypes as C

import numpy as np

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

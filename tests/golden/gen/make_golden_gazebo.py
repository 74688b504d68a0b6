#!/usr/bin/env python3
"""Golden vectors of the follower-relative ("Gazebo") tracker / ray sensors -- runs ONLY in the build container.

Drives the UNMODIFIED reference classes GazeboLeaderPositionsTracker_v2 and GazeboCorridor_Prev_lasers_v2
(/root/reference/src/arctic_gym/gazebo_utils/gazebo_tracker.py:13-297), constructed as arctic_env.py:62-90 does, on synthetic
follower / leader motion and lidar point pairs, and dumps inputs + outputs per call into tests/golden/gazebo_<name>.npz.

Two accommodations, both on the LIBRARY side of the boundary (the reference file is executed as it is):
  * the module is loaded by file path: `src/arctic_gym/__init__.py` imports ray, which is not installed;
  * numpy >= 2 raises ValueError for `ndarray != []` with unbroadcastable shapes, which the laser's scan() evaluates at
    gazebo_tracker.py:226; numpy 1.x -- what the reference ran on (ray 1.9.5 / torch 1.13 pin it) -- returned the scalar True with a
    DeprecationWarning.  The generator hands the module a numpy whose `array()` yields an ndarray subclass with exactly that legacy
    answer for a comparison with an empty list; every other operation is numpy 2.2's own.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")


class _LegacyNe(np.ndarray):
    def __ne__(self, other):
        if isinstance(other, list) and len(other) == 0:
            return True                      # numpy 1.x: "elementwise comparison failed; returning scalar instead"
        return np.ndarray.__ne__(self, other)


def load_reference():
    with contextlib.redirect_stdout(io.StringIO()):
        spec = importlib.util.spec_from_file_location("gazebo_tracker", "/root/reference/src/arctic_gym/gazebo_utils/gazebo_tracker.py")
        gt = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gt)
    legacy = types.ModuleType("numpy_legacy_ne")
    legacy.__dict__.update(np.__dict__)
    legacy.array = lambda *a, **k: np.array(*a, **k).view(_LegacyNe)
    gt.np = legacy
    return gt


LASERS = (dict(sensor_name="LeaderCorridor_Prev_lasers_v2_compas", react_to_green_zone=True, react_to_safe_corridor=True,
               react_to_obstacles=True, lasers_count=12, laser_length=10, max_prev_obs=10, pad_sectors=False),     # arctic_env.py:69-78
          dict(sensor_name="LaserPrevSensor_compas", react_to_green_zone=False, react_to_safe_corridor=False,
               react_to_obstacles=True, lasers_count=36, laser_length=15, max_prev_obs=10, pad_sectors=False))     # arctic_env.py:80-90
LASERS_PAD = (dict(LASERS[0], max_prev_obs=4, pad_sectors=True, lasers_count=20), dict(LASERS[1], max_prev_obs=4, lasers_count=24))


def run(name, seed, steps, lasers, max_pts=48):
    gt = load_reference()
    rng = np.random.default_rng(seed)
    trk = gt.GazeboLeaderPositionsTracker_v2(host_object="arctic_robot", sensor_name="LeaderTrackDetector", saving_period=5,
                                             corridor_width=2, corridor_length=25)                                 # arctic_env.py:62-66
    sens = [gt.GazeboCorridor_Prev_lasers_v2(host_object="arctic_robot", **kw) for kw in lasers]
    # world-frame motion: the follower chases a leader that wanders ahead of it; rocks are fixed world points seen by a fake lidar
    f = np.array([0.0, 0.0]); fyaw = rng.uniform(-0.5, 0.5)
    lead = f + 7.0 * np.array([np.cos(fyaw), np.sin(fyaw)]); lyaw = fyaw
    rocks = rng.uniform(-25, 60, (40, 2))
    rec = dict(leader=[], yaw=[], delta=[], pts1=[], pts2=[], n_pts=[], counter=[], hist=[], corr=[], n_hist=[], n_corr=[])
    outs = [[] for _ in sens]
    prev_f = f.copy()
    for t in range(steps):
        lyaw += rng.normal(0, 0.08)
        lead = lead + (0.35 if (t // 25) % 3 != 2 else 0.02) * np.array([np.cos(lyaw), np.sin(lyaw)])            # the leader pauses now and then
        want = np.arctan2(lead[1] - f[1], lead[0] - f[0])
        fyaw += np.clip((want - fyaw + np.pi) % (2 * np.pi) - np.pi, -0.12, 0.12)
        gap = np.linalg.norm(lead - f)
        f = f + (0.4 if gap > 6 else 0.1) * np.array([np.cos(fyaw), np.sin(fyaw)])
        delta = {"delta_x": np.float64(f[0] - prev_f[0]), "delta_y": np.float64(f[1] - prev_f[1])}               # arctic_env.py:165-175
        prev_f = f.copy()
        leader_rel = lead - f
        orient = np.array([0.0, 0.0, fyaw])
        # fake lidar: points on the rocks within 16 m, as pairs (one end, the other end) -- some closer than 0.5 m to their successor
        rel = rocks - f
        near = rel[np.linalg.norm(rel, axis=1) < 16.0]
        near = near[np.argsort(np.arctan2(near[:, 1], near[:, 0]))][:max_pts // 2]
        p1 = np.repeat(near, 2, axis=0) + np.tile(np.array([[0.0, 0.0], [0.3, 0.1]]), (len(near), 1)) if len(near) else np.zeros((0, 2))
        p2 = p1 + rng.uniform(0.4, 1.2, p1.shape)
        hist, corr = trk.scan(leader_rel, orient, delta)
        for k, s in enumerate(sens):
            with np.errstate(all="ignore"):
                outs[k].append(np.asarray(s.scan(orient, corr, list(p1), list(p2))).astype(np.float32 if not lasers[k]["pad_sectors"] else np.float64))
        rec["leader"].append(leader_rel.copy()); rec["yaw"].append(fyaw); rec["delta"].append([delta["delta_x"], delta["delta_y"]])
        a1 = np.zeros((max_pts, 2)); a2 = np.zeros((max_pts, 2)); a1[:len(p1)] = p1; a2[:len(p2)] = p2
        rec["pts1"].append(a1); rec["pts2"].append(a2); rec["n_pts"].append(len(p1))
        h = np.full((64, 2), np.nan); c = np.full((64, 4), np.nan)
        hh = np.array([np.asarray(p, np.float64) for p in hist]).reshape(-1, 2)
        cc = np.array([[q[0][0], q[0][1], q[1][0], q[1][1]] for q in corr], np.float64).reshape(-1, 4)
        h[:len(hh)] = hh; c[:len(cc)] = cc
        rec["hist"].append(h); rec["corr"].append(c); rec["n_hist"].append(len(hh)); rec["n_corr"].append(len(cc)); rec["counter"].append(trk.saving_counter)
    out = {k: np.array(v) for k, v in rec.items()}
    for k, o in enumerate(outs):
        out["laser%d" % k] = np.stack(o)
    import json
    out["meta"] = np.array(json.dumps(dict(name=name, seed=seed, steps=steps, max_pts=max_pts, numpy=np.__version__,
                                           lasers=[{k: v for k, v in kw.items() if k != "sensor_name"} for kw in lasers])))
    path = os.path.join(GOLDEN, "gazebo_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("gazebo", name, "steps", steps, "hist", out["n_hist"].min(), out["n_hist"].max(), "corr", out["n_corr"].max(),
          "pts", out["n_pts"].max(), "laser0 hit frac", float((out["laser0"] < lasers[0]["laser_length"] * 0.999).mean()),
          "laser1 hit frac", float((out["laser1"] < lasers[1]["laser_length"] * 0.999).mean()), os.path.getsize(path))


if __name__ == "__main__":
    run("arctic_s1", 1, 160, LASERS)
    run("arctic_s4", 4, 120, LASERS)
    run("pad_s7", 7, 100, LASERS_PAD)

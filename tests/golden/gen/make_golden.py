#!/usr/bin/env python3
"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

Executes the UNMODIFIED reference ``Game`` (``/root/reference/src/continuous_grid_arctic/
follow_the_leader_continuous_env.py``) on top of the build-authored stand-ins for the two
libraries the image lacks (``standins/pygame``, ``standins/gym``) and
dumps, per episode, the post-``reset()`` scenario plus the per-step trace
(action, obs, reward, done, info) as a small ``.npz`` under ``tests/golden/``.

Determinism policy:
  * ``pygame.time.get_ticks()`` seen by frame k (k = 1, 2, ... counted from ``reset()``) is k;
  * actions enter the reference as Python floats (f64);
  * ``seed(s)`` seeds ``random`` / ``np.random`` exactly as the reference does (``ENV:429-432``).

Nothing here is imported by the product; the GPU box never runs this file.

usage:  python tests/golden/gen/make_golden.py [--only NAME] [--pool P] [--jobs J]
"""
import argparse
import io
import json
import math
import os
import sys
import contextlib
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, os.path.dirname(os.path.dirname(GOLDEN)))

SENSORS_B = OrderedDict([
    ("LeaderPositionsTracker_v2", {
        "sensor_class": "LeaderPositionsTracker_v2", "eat_close_points": False,
        "generate_corridor": True, "saving_period": 8,
        "sensor_name": "LeaderPositionsTracker_v2", "start_corridor_behind_follower": True,
        "corridor_length": 250, "corridor_width": 30}),
    ("LeaderCorridor_lasers_all", {
        "sensor_name": "LeaderCorridor_lasers_all", "sensor_class": "LeaderCorridor_Prev_lasers_v2",
        "react_to_green_zone": True, "react_to_obstacles": True, "react_to_safe_corridor": True,
        "lasers_count": 12, "laser_length": 100, "max_prev_obs": 5, "use_prev_obs": True,
        "pad_sectors": False}),
    ("LeaderCorridor_lasers_obstacles", {
        "sensor_name": "LeaderCorridor_lasers_obstacles", "sensor_class": "LeaderCorridor_Prev_lasers_v2",
        "react_to_green_zone": False, "react_to_obstacles": True, "react_to_safe_corridor": False,
        "lasers_count": 24, "laser_length": 150, "max_prev_obs": 5, "use_prev_obs": True,
        "pad_sectors": False}),
])

SENSORS_D = OrderedDict([
    ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"])),
    ("LaserPrevSensor_180", {
        "sensor_name": "LaserPrevSensor_180", "sensor_class": "LeaderCorridor_Prev_lasers_v2",
        "react_to_green_zone": False, "react_to_obstacles": True, "react_to_safe_corridor": False,
        "lasers_count": 36, "laser_length": 200, "max_prev_obs": 5, "use_prev_obs": True,
        "pad_sectors": False, "first_laser_angle_offset": 0}),
])

# kwargs handed to the reference Game(**kwargs); "_post" = attribute patches applied to sensor
# objects after each reset (config D: the reference rejects lasers_count=180 at construction,
# SEN:761-762, so the documented interpretation sets it afterwards -- SURVEY.md 8(d)).
CONFIGS = {
    "A": dict(kwargs=dict(), post=None),
    "B": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B), post=None),
    "D": dict(kwargs=dict(bear_number=1, obstacle_number=100, follower_sensors=SENSORS_D),
              post={"LaserPrevSensor_180": dict(lasers_count=180, laser_period=2.0)}),
    # early-stopping variant of B (exercises ENV:1088-1107)
    "B_es": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B,
                             early_stopping={"max_distance_coef": 1.2, "low_reward": -100}), post=None),
    # short horizon (exercises finished_by_time ENV:1129-1134 and the warm_start gate)
    "B_short": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, max_steps=450, warm_start=100), post=None),
    # no dynamic obstacles (add_bear=False): long clean episodes -> "success" (ENV:1077-1087)
    "B_nobear": dict(kwargs=dict(add_bear=False, follower_sensors=SENSORS_B), post=None),
    # pad_sectors=True output layout (SEN:932-953) on the 12-ray sensor, tracker LAST in the dict (as in
    # server/config/3c1bc/params.json:26-60: the ray sensors are scanned BEFORE the tracker's second scan)
    "B_pad": dict(kwargs=dict(bear_number=2, follower_sensors=OrderedDict([
        ("LeaderCorridor_lasers_all", dict(SENSORS_B["LeaderCorridor_lasers_all"], pad_sectors=True)),
        ("LeaderCorridor_lasers_obstacles", dict(SENSORS_B["LeaderCorridor_lasers_obstacles"], lasers_count=20, react_to_obstacles="dynamic")),
        ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"]))])), post=None),
    # config E ("hardcore"): the parameter set of TestGameManual_gazebo (ENV:2015-2105) with manual_control=False --
    # speed / acceleration regimes, negative follower speed, 2 bears, 5 frames per step, early stopping
    "E": dict(kwargs=dict(pixels_to_meter=10, step_grid=10, max_steps=30000, framerate=90, frames_per_step=5,
                          min_distance=8, max_distance=15, max_dev=1, warm_start=0, follower_size=(1, 1), leader_size=(4, 2),
                          bear_size=(1.5, 1.5), follower_max_speed=2, leader_max_speed=1, negative_speed=True,
                          bear_max_speed=1.2, follower_max_rotation_speed=28.65, leader_max_rotation_speed=28.65,
                          follower_acceleration=1, leader_acceleration=1, leader_margin=1,
                          leader_speed_regime=OrderedDict([("0", [0.2, 1]), ("200", 1), ("1000", [0.5, 1]), ("1500", 0.75),
                                                           ("2300", 0), ("2500", 1), ("3000", [0.5, 1]), ("4000", [0.0, 0.5]),
                                                           ("5000", [0.4, 1])]),
                          obstacle_number=20, bear_number=2, bridge_size=[140, 40],
                          leader_acceleration_regime=OrderedDict([("0", 0), ("3100", 0.03), ("4500", 0)]),
                          early_stopping={"max_distance_coef": 4, "low_reward": -300},
                          follower_sensors=SENSORS_B), post=None),
    # config F: env_config.base_env_config of the shipped training run server/config/3c1bc/params.json (also "transformer"):
    # random_frames_per_step [30, 70], max_prev_obs 10, the v2 tracker registered LAST under the key "LeaderPositionsTracker",
    # random speed regimes (JSON string keys in the file's sorted order), negative follower speed, 2 bears, early stopping
    "F": dict(kwargs=dict(add_bear=True, add_obstacles=True, bear_behind=False, bear_max_speed=1.2, bear_number=2,
                          bear_size=[1.5, 1.5], bridge_size=[140, 40], constant_follower_speed=False,
                          early_stopping={"low_reward": -300, "max_distance_coef": 3.5}, follower_acceleration=1,
                          follower_max_rotation_speed=28.65, follower_max_speed=2,
                          follower_sensors=OrderedDict([
                              ("LeaderCorridor_lasers_all", dict(laser_length=100, lasers_count=12, max_prev_obs=10, pad_sectors=False,
                                                                 react_to_green_zone=True, react_to_obstacles=True, react_to_safe_corridor=True,
                                                                 sensor_class="LeaderCorridor_Prev_lasers_v2", sensor_name="LeaderCorridor_lasers_all",
                                                                 use_prev_obs=True)),
                              ("LeaderCorridor_lasers_obstacles", dict(laser_length=150, lasers_count=24, max_prev_obs=10, pad_sectors=False,
                                                                       react_to_green_zone=False, react_to_obstacles=True, react_to_safe_corridor=False,
                                                                       sensor_class="LeaderCorridor_Prev_lasers_v2",
                                                                       sensor_name="LeaderCorridor_lasers_obstacles", use_prev_obs=True)),
                              ("LeaderPositionsTracker", dict(corridor_length=250, corridor_width=30, eat_close_points=False, generate_corridor=True,
                                                              saving_period=8, sensor_class="LeaderPositionsTracker_v2",
                                                              sensor_name="LeaderPositionsTracker", start_corridor_behind_follower=True))]),
                          follower_size=[1, 1], framerate=5000, game_height=1000, game_width=1500, ignore_follower_collisions=False,
                          leader_acceleration=1, leader_margin=1, leader_max_rotation_speed=28.65, leader_max_speed=1, leader_size=[4, 2],
                          leader_speed_regime=OrderedDict([("0", [0.2, 1]), ("1000", [0.5, 1]), ("1500", 0.75), ("200", 1), ("2300", 0),
                                                           ("2500", 1), ("3000", [0.5, 1]), ("4000", [0.0, 0.5]), ("5000", [0.4, 1])]),
                          max_dev=1, max_distance=15, max_steps=30000, min_distance=8, move_bear_v4=True, multi_random_bears=False,
                          multiple_end_points=False, negative_speed=True, obstacle_number=20, path_finding_iterations=15000,
                          pixels_to_meter=10, random_frames_per_step=[30, 70], step_grid=10, warm_start=0), post=None),
    # config G: B plus the two other live sensor classes -- LeaderCorridor_lasers_v2 (current edges only, ray 0 straight
    # ahead, no history; SEN:736-807) once before and once after the tracker in dict order, and FollowerInfo (SEN:822-845)
    "G": dict(kwargs=dict(bear_number=1, follower_sensors=OrderedDict([
        ("lasers_now_first", {"sensor_class": "LeaderCorridor_lasers_v2", "react_to_obstacles": True, "react_to_green_zone": True,
                              "react_to_safe_corridor": True, "lasers_count": 20, "laser_length": 120}),
        ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"])),
        ("LeaderCorridor_lasers_all", dict(SENSORS_B["LeaderCorridor_lasers_all"])),
        ("FollowerInfo", {"sensor_class": "FollowerInfo"}),
        ("lasers_now", {"sensor_class": "LeaderCorridor_lasers_v2", "react_to_obstacles": "static", "lasers_count": 36,
                        "laser_length": 150}),
        ("LeaderCorridor_lasers", {"react_to_obstacles": True, "react_to_green_zone": True, "front_lasers_count": 5,
                                   "back_lasers_count": 2, "laser_length": 90})])), post=None),
    # three bears (odd index -> _move_bear_v4), sensors of B
    "B3": dict(kwargs=dict(bear_number=3, follower_sensors=SENSORS_B), post=None),
    # the A* planner variant of reset() (ENV:1632-1711 on utils/astar.py): greedy f = g + squared distance on a 20 px grid, two legs through
    # the bridge, a 1000-expansion cap that returns the path to the last expanded node
    "B_astar": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, path_finding_algorythm="astar"), post=None),
    # six bears: index 4 snakes at radius 500, index 5 chases four way-points drawn anew every frame (ENV:750-754)
    "B6": dict(kwargs=dict(bear_number=6, follower_sensors=SENSORS_B), post=None),
    # config C: B's tracker + Prev sensor + two LeaderCorridor_lasers_compas (SEN:1138-1288), one scanned before the tracker's second
    # scan of the step and one after it
    "C": dict(kwargs=dict(bear_number=1, follower_sensors=OrderedDict([
        ("compas_first", {"sensor_class": "LeaderCorridor_lasers_compas", "react_to_green_zone": True, "react_to_safe_corridor": True,
                          "react_to_obstacles": False, "lasers_count": 12, "laser_length": 90, "max_prev_obs": 5, "pad_sectors": False}),
        ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"])),
        ("LeaderCorridor_lasers_all", dict(SENSORS_B["LeaderCorridor_lasers_all"])),
        ("LeaderCorridor_lasers_compas", {"sensor_class": "LeaderCorridor_lasers_compas", "react_to_green_zone": True,
                                          "react_to_safe_corridor": True, "react_to_obstacles": False, "lasers_count": 20,
                                          "laser_length": 120, "max_prev_obs": 5, "pad_sectors": False,
                                          "first_laser_angle_offset": 0})])), post=None),
    # config L: lidar (SEN:18-145, offsets and distances-only) + the leader-track detectors (SEN:342-487) on the v2 tracker's history,
    # one detector registered BEFORE the tracker (it sees the history after the first scan of the step only)
    "L": dict(kwargs=dict(bear_number=2, follower_sensors=OrderedDict([
        ("track_old_first", {"sensor_class": "LeaderTrackDetector_vector", "position_sequence_length": 8, "detectable_positions": "old"}),
        ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"])),
        ("LaserSensor", {"sensor_class": "LaserSensor"}),
        ("lidar_front", {"sensor_class": "LaserSensor", "available_angle": 180, "angle_step": 15, "points_number": 10, "sensor_range": 3,
                         "return_only_distances": True}),
        ("track_new", {"sensor_class": "LeaderTrackDetector_vector", "position_sequence_length": 20, "detectable_positions": "new"}),
        ("radar_near", {"sensor_class": "LeaderTrackDetector_radar", "detectable_positions": "near", "radar_sectors_number": 36}),
        ("radar_old", {"sensor_class": "LeaderTrackDetector_radar", "position_sequence_length": 12, "radar_sectors_number": 18}),
        ("LeaderCorridor_lasers_all", dict(SENSORS_B["LeaderCorridor_lasers_all"]))])), post=None),
    # config L_all: LaserSensor(return_all_points=True) -- every marching point up to the first hit of every ray (SEN:112-113, 131-134), once as
    # points and once as distances, next to the sensors of B (two bears so that the lidars see moving rects)
    "L_all": dict(kwargs=dict(bear_number=2, follower_sensors=OrderedDict([
        ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"])),
        ("lidar_all", {"sensor_class": "LaserSensor", "available_angle": 120, "angle_step": 20, "points_number": 8, "sensor_range": 3,
                       "return_all_points": True}),
        ("lidar_all_d", {"sensor_class": "LaserSensor", "available_angle": 360, "angle_step": 45, "points_number": 12, "sensor_range": 5,
                         "return_all_points": True, "return_only_distances": True}),
        ("LeaderCorridor_lasers_all", dict(SENSORS_B["LeaderCorridor_lasers_all"]))])), post=None),
    # config T: the deprecated v1 tracker (SEN:148-229; scanned once per step, corridor half-width max_dev, eat_close_points) feeding a
    # lasers_v2 sensor, a LeaderCorridor_lasers one and both detectors (a Prev_lasers_v2 sensor raises at reset on this tracker: its
    # corridor holds one pair after the first scan, SEN:893/962)
    "T": dict(kwargs=dict(bear_number=1, follower_sensors=OrderedDict([
        ("LeaderPositionsTracker", {"sensor_class": "LeaderPositionsTracker", "saving_period": 4}),
        ("LeaderCorridor_lasers", {"sensor_class": "LeaderCorridor_lasers", "react_to_obstacles": "dynamic", "front_lasers_count": 3}),
        ("lasers_now", {"sensor_class": "LeaderCorridor_lasers_v2", "react_to_obstacles": True, "react_to_green_zone": True,
                        "react_to_safe_corridor": True, "lasers_count": 24, "laser_length": 130}),
        ("track_new", {"sensor_class": "LeaderTrackDetector_vector", "position_sequence_length": 16, "detectable_positions": "new"}),
        ("radar_new", {"sensor_class": "LeaderTrackDetector_radar", "position_sequence_length": 30, "detectable_positions": "new",
                       "radar_sectors_number": 20})])), post=None),
    # config M ("mixed"): a Prev_lasers_v2 sensor next to a LeaderCorridor_lasers_v2 one and FollowerInfo, every entry with a
    # "sensor_class" key so that ContinuousObserveModifier_sensorPrev constructs on it -- the wrapper must pick the Prev sensor
    # only (wrappers.py:204, 214)
    "M": dict(kwargs=dict(bear_number=1, follower_sensors=OrderedDict([
        ("LeaderPositionsTracker_v2", dict(SENSORS_B["LeaderPositionsTracker_v2"])),
        ("lasers_now", {"sensor_class": "LeaderCorridor_lasers_v2", "react_to_obstacles": True, "react_to_green_zone": True,
                        "react_to_safe_corridor": True, "lasers_count": 24, "laser_length": 130}),
        ("LeaderCorridor_lasers_all", dict(SENSORS_B["LeaderCorridor_lasers_all"])),
        ("FollowerInfo", {"sensor_class": "FollowerInfo"}),
        ("LeaderCorridor_lasers_obstacles", dict(SENSORS_B["LeaderCorridor_lasers_obstacles"], pad_sectors=True))])), post=None),
    # ---- constructor switches of the step path that no other config turns on (round 3) --------------------------------------------
    # config N: TestGameNEAT (ENV:2125-2129, "Test-Game-Neat-v0"): Discrete(5) actions decoded by ENV:360-367, 918-922, no rocks
    # (add_obstacles=False, ENV:322-323, 464-465), default three bears, no sensors.  "action": what Runner.step hands to Game.step.
    # With the default D* planner the reference's reset() raises AttributeError ('Game' object has no attribute 'obstacles1', ENV:1501:
    # generate_trajectory_dstar reads the bridge walls that _create_obstacles never made) -- so the registered ids
    # "Test-Game-Neat-v0" and "Test-Cont-Env-Auto-Follow-no-obstacles-v0" cannot reset; A* guards that code (ENV:1666).
    "N": dict(kwargs=dict(add_obstacles=False, discrete_action_space=True, path_finding_algorythm="astar",
                          early_stopping={"max_distance_coef": 1.2, "low_reward": -100}), post=None, action="discrete"),
    # constant_follower_speed (ENV:368-372, 910-911, 924-925): Box(1) action = rotation only, speed command 0.25 px/frame
    "B_cfs": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, constant_follower_speed=True), post=None, action="turn"),
    # ignore_follower_collisions (ENV:960): the follower drives through rocks and robots, no crash
    "B_nocoll": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, ignore_follower_collisions=True), post=None),
    # aggregate_reward (ENV:1136-1141): step() returns overall_reward instead of the last frame's reward
    "B_agg": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, aggregate_reward=True), post=None),
    # add_obstacles=False with the sensors of B (TestGameBaseAlgoNoObst, ENV:2109-2112): the static list is empty
    "B_noobst": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, add_obstacles=False, path_finding_algorythm="astar",
                                 early_stopping={"max_distance_coef": 1.2, "low_reward": -100}), post=None),
    # multiple_end_points (ENV:470-481, 1552-1592): three finish points, three D* legs chained into one route
    "B_mep": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B, multiple_end_points=True), post=None),
    # caller-supplied trajectory= (ENV:229, 469-470): no finish point, no planner; the leader heads for trajectory[1]
    "B_traj": dict(kwargs=dict(bear_number=1, follower_sensors=SENSORS_B,
                               trajectory=[(1400, 500), (1150, 500), (900, 500), (750, 500), (600, 500), (400, 420), (250, 300), (120, 200)]),
                   post=None),
}


def import_reference():
    import pygame  # stand-in
    with contextlib.redirect_stdout(io.StringIO()):
        import continuous_grid_arctic.follow_the_leader_continuous_env as ENV
    return pygame, ENV


class Runner:
    """One reference Game with the deterministic tick installed."""

    def __init__(self, config_name):
        self.pygame, ENV = import_reference()
        cfg = CONFIGS[config_name]
        self.cfg = cfg
        import copy
        # np.random.randint of random_frames_per_step (ENV:405 in the constructor, ENV:940 at the end of
        # every step) is replaced by the build's counter stream keyed on (rng_seed=0, env 0, reset number, frame) --
        # include/ftl.h ftl_rand_frames; installed BEFORE the constructor runs (resets = 0, frame 0 there)
        from continiousenvironment_follower_leader_amd.abi import rand_frames
        self.resets = 0
        self.game = None

        def randint(lo, hi=None, *a, **k):
            return rand_frames(0, 0, self.resets, 0 if self.game is None else self.game.step_count, int(lo), int(hi))
        ENV.np.random.randint = randint
        self.game = ENV.Game(**copy.deepcopy(cfg["kwargs"]))
        self.frame = 0
        orig = self.game.frame_step

        # bears with an odd index >= 5 draw four way-points per frame with random.randrange (ENV:750-754); the
        # draws are replaced by the build's counter stream (include/ftl.h ftl_rand_range) for the duration of a frame only -- reset()
        # keeps CPython's own randrange for the rocks
        import random as _rnd
        from continiousenvironment_follower_leader_amd.abi import rand_range
        qual = [b for b in range(self.game.bear_number if self.game.add_bear else 0) if b % 2 and b >= 4]

        def frame_step(action):
            self.frame += 1
            calls = [0]

            def randrange(start, stop, step=1):
                assert step == 10 and qual, (start, stop, step)
                k = calls[0]
                calls[0] += 1
                return rand_range(0, 0, self.resets, self.game.step_count, qual[k // 8], k % 8, int(start), int(stop))
            real = _rnd.randrange
            _rnd.randrange = randrange
            try:
                return orig(action)
            finally:
                _rnd.randrange = real

        self.game.frame_step = frame_step
        self.pygame.time._ticks_fn = lambda: self.frame
        # the only `random` draw inside step() is random.uniform at ENV:1156; it is replaced by the
        # build's counter-based stream keyed on (rng_seed=0, env 0, reset number, frame) -- include/ftl.h ftl_uniform01
        import random as _random
        from continiousenvironment_follower_leader_amd.abi import uniform01

        def uniform(a, b):
            return a + (b - a) * uniform01(0, 0, self.resets, self.game.step_count)
        _random.uniform = uniform

    def reset(self, seed):
        g = self.game
        g.seed(seed)
        self.frame = 0
        self.resets += 1
        with contextlib.redirect_stdout(io.StringIO()):
            # the reference's reset() runs use_sensors itself; for config D the sensor must be
            # patched BEFORE that first scan, so intercept robot creation.
            if self.cfg["post"]:
                orig_create = g._create_robots
                post = self.cfg["post"]

                def create():
                    orig_create()
                    for sname, attrs in post.items():
                        for k, v in attrs.items():
                            setattr(g.follower.sensors[sname], k, v)
                g._create_robots = create
                try:
                    obs = g.reset()
                finally:
                    g._create_robots = orig_create
            else:
                obs = g.reset()
        return obs

    def step(self, action):
        """``action`` = the decoded (speed, rotation) pair for the Box(2) configs; the raw action (an int of Discrete(5) / the rotation
        of Box(1)) for configs with an "action" kind -- handed to Game.step in the form a gym policy would produce it."""
        kind = self.cfg.get("action")
        with contextlib.redirect_stdout(io.StringIO()):
            if kind == "discrete":
                k = int(action)
                return self.game.step(np.array([[k]]) if k % 2 else k)       # both accepted forms (ENV:919-921)
            if kind == "turn":
                return self.game.step(np.array([action], dtype=np.float32))   # Box(shape=(1,), dtype=float32), ENV:368-372
            return self.game.step((float(action[0]), float(action[1])))


MISSION = {"in_progress": 0, "fail": 1, "success": 2, "finished_by_time": 3}
AGENT = {"moving": 0, "crash": 1, "low_reward": 2, "too_far_from_leader": 3, "finished": 4}
LEADER = {"moving": 0, "crash": 1, "finished": 2}


def robot_record(r):
    """[pos_x, pos_y (f32 values), direction, speed, rotation_speed] + rect + rotation_direction."""
    rect = r.rectangle
    return (np.array([r.position[0], r.position[1]], dtype=np.float32),
            np.array([float(r.direction), float(r.speed), float(r.rotation_speed),
                      float(r.desirable_speed), float(r.desirable_rotation_speed)], dtype=np.float64),
            np.array([rect.x, rect.y, rect.w, rect.h, int(r.rotation_direction),
                      int(r.desirable_rotation_direction)], dtype=np.int32))


def scenario_of(g):
    """Post-reset scenario = everything the step path needs that reset() produced
    (ENV:434-543); unaffected by the use_sensors call at ENV:541."""
    robots = [g.leader, g.follower] + list(g.game_dynamic_list)
    pos, f64, i32 = zip(*[robot_record(r) for r in robots])
    statics = [o for o in g.game_object_list if o is not g.leader and o is not g.follower]
    srect = np.array([[o.rectangle.x, o.rectangle.y, o.rectangle.w, o.rectangle.h] for o in statics],
                     dtype=np.int32).reshape(-1, 4)
    traj = np.array([[float(p[0]), float(p[1])] for p in g.leader_factual_trajectory], dtype=np.float32).reshape(-1, 2)
    # check that the f32 values survive the float() round trip exactly
    for p, q in zip(g.leader_factual_trajectory, traj):
        assert np.float32(p[0]) == q[0] and np.float32(p[1]) == q[1]
    route = np.array(g.trajectory, dtype=np.float64).reshape(-1, 2)
    return dict(robot_pos=np.stack(pos), robot_f64=np.stack(f64), robot_i32=np.stack(i32),
                static_rects=srect, route=route, init_traj=traj,
                found_target_point=np.array(bool(g.found_target_point)),
                bear_points=np.array([[float(p[0]), float(p[1])] for p in getattr(g, "cur_points_for_bear", [])],
                                     dtype=np.float64).reshape(-1, 2),
                done_at_reset=np.array(bool(g.done)))


def obs_record(g, obs, laser_names):
    rec = {"num": np.asarray(obs["numerical_features"], dtype=np.float32),
           "target": np.array([float(obs["leader_target_point"][0]), float(obs["leader_target_point"][1])])}
    for n in laser_names:
        a = np.asarray(obs[n])
        a = a.astype(np.float32) if a.dtype == np.float32 else a.astype(np.float64)
        rec["laser:" + n] = a[None, :] if a.ndim == 1 else a       # LeaderCorridor_lasers_v2 returns [N]: stored as one row
    for n, v in g.follower_sensors.items():
        if v.get("sensor_class", n) == "FollowerInfo":
            rec["finfo:" + n] = np.asarray(obs[n], dtype=np.float32)
        if v.get("sensor_class", n) in ("LaserSensor", "LeaderTrackDetector_vector", "LeaderTrackDetector_radar"):
            a = np.asarray(obs[n])
            assert a.dtype == np.float32, (n, a.dtype)
            if v.get("sensor_class", n) == "LaserSensor" and v.get("return_all_points", False):
                # K rows, K changing from step to step (SEN:112-113): stored in the batched layout [K][rows][zeros] (include/ftl.h)
                sens = g.follower.sensors[n]
                n_ang = 1 + 2 * len(np.arange(0, int(sens.available_angle / 2), sens.angle_step))
                cap = n_ang * sens.points_number * (1 if sens.return_only_distances else 2)
                blk = np.zeros(1 + cap, np.float32)
                blk[0] = len(a)
                blk[1:1 + a.size] = a.reshape(-1)
                a = blk
            rec["aux:" + n] = a.copy()
    return rec


def debug_record(g):
    """Internal state used only to localise a parity failure (never an API contract)."""
    robots = [g.leader, g.follower] + list(g.game_dynamic_list)
    pos, f64, i32 = zip(*[robot_record(r) for r in robots])
    d = dict(robot_pos=np.stack(pos), robot_f64=np.stack(f64), robot_i32=np.stack(i32),
             counters=np.array([g.step_count, len(g.leader_factual_trajectory),
                                len(g.green_zone_trajectory_points), g.cur_target_id,
                                int(g.leader_finished), int(g.is_in_box), int(g.is_on_trace),
                                int(g.follower_too_close), int(g.crash), int(g.done),
                                -1 if g.finish_position_framestimer is None else g.finish_position_framestimer],
                               dtype=np.int64),
             acc=np.array([float(g.accumulated_penalty), float(g.overall_reward)]))
    tr = g.follower.sensors.get("LeaderPositionsTracker_v2") if hasattr(g.follower, "sensors") else None
    if tr is None and hasattr(g.follower, "sensors") and type(g.follower.sensors.get("LeaderPositionsTracker")).__name__ == "LeaderPositionsTracker":
        tr = g.follower.sensors["LeaderPositionsTracker"]      # the v1 class under its own key (config T)
    if tr is not None:
        hist = np.array([[float(p[0]), float(p[1])] for p in tr.leader_positions_hist], dtype=np.float64).reshape(-1, 2)
        corr = np.array([[float(c[0][0]), float(c[0][1]), float(c[1][0]), float(c[1][1])] for c in tr.corridor],
                        dtype=np.float64).reshape(-1, 4)
        isf64 = np.array([np.asarray(p).dtype == np.float64 for p in tr.leader_positions_hist], dtype=np.uint8)
        d.update(trk=np.array([tr.saving_counter, len(hist), len(corr)], dtype=np.int64))
        d["hist"] = hist
        d["corr"] = corr
        d["hist_isf64"] = isf64
    if hasattr(g, "dynamics_index"):
        d["dyn_index"] = np.array(g.dynamics_index, dtype=np.int64)
    return d


def chase_action(g, rng, noise):
    """A simple pursuit policy (so that episodes are long and rewards non-trivial) + noise."""
    f, l = g.follower, g.leader
    # aim at a point max(min_distance*1.6, ..) behind the leader along the factual trajectory
    tgt = g.leader_factual_trajectory[max(0, len(g.leader_factual_trajectory) - 70)]
    dx, dy = float(tgt[0]) - float(f.position[0]), float(tgt[1]) - float(f.position[1])
    want = math.degrees(math.atan2(dy, dx)) % 360.0
    err = (want - float(f.direction) + 540.0) % 360.0 - 180.0
    w = max(-f.max_rotation_speed, min(f.max_rotation_speed, err * 0.3))
    dist = math.hypot(float(l.position[0]) - float(f.position[0]), float(l.position[1]) - float(f.position[1]))
    v = f.max_speed if dist > g.min_distance * 2.4 else (0.0 if dist < g.min_distance * 1.5 else 0.9 * f.max_speed)
    w += noise * rng.normal() * f.max_rotation_speed
    v *= (1.0 - 0.3 * noise * rng.random())
    w = max(-f.max_rotation_speed, min(f.max_rotation_speed, w))
    return (float(v), float(w))


def random_action(g, rng):
    f = g.follower
    v = rng.uniform(0.5, 1.0) * f.max_speed
    w = float(np.clip(rng.normal(0.0, 0.2 * f.max_rotation_speed), -f.max_rotation_speed, f.max_rotation_speed))
    return (float(v), w)


def ram_action(g, rng, statics_only=False):
    """Full speed at the nearest obstacle rect or the leader (episodes of ignore_follower_collisions must drive THROUGH things)."""
    f = g.follower
    objs = [o for o in g.game_object_list if o is not f and not (statics_only and o is g.leader)] + ([] if statics_only else list(g.game_dynamic_list))
    fx, fy = float(f.position[0]), float(f.position[1])
    tgt = min(objs, key=lambda o: math.hypot(o.rectangle.centerx - fx, o.rectangle.centery - fy) + (200 if o.rectangle.collidepoint(fx, fy) else 0))
    want = math.degrees(math.atan2(tgt.rectangle.centery - fy, tgt.rectangle.centerx - fx)) % 360.0
    err = (want - float(f.direction) + 540.0) % 360.0 - 180.0
    w = max(-f.max_rotation_speed, min(f.max_rotation_speed, err * 0.3 + 0.05 * rng.normal() * f.max_rotation_speed))
    return (float(f.max_speed), float(w))


def sensor_prev_wrapper(g):
    """The reference's own ContinuousObserveModifier_sensorPrev (utils/wrappers.py:169-221) around the Game, or None when its
    constructor cannot run on this sensor dict (it indexes sensor_config["sensor_class"] / ["pad_sectors"] unconditionally)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import continuous_grid_arctic.utils.wrappers as WRP
    hs = {v.get("max_prev_obs") for v in g.follower_sensors.values()
          if v.get("sensor_class") in ("LeaderCorridor_Prev_lasers_v2", "LeaderCorridor_Prev_lasers_v3", "LeaderCorridor_lasers_compas")}
    if len(hs) != 1:
        return None
    try:
        return WRP.ContinuousObserveModifier_sensorPrev(g, max_prev_obs=hs.pop())
    except KeyError:
        return None


def run_episode(config_name, seed, policy, n_steps, debug_every=1, stop_after_done=3):
    r = Runner(config_name)
    obs0 = r.reset(seed)
    g = r.game
    wrap = sensor_prev_wrapper(g)
    laser_names = [k for k, v in g.follower_sensors.items()
                   if v.get("sensor_class", k) in ("LeaderCorridor_Prev_lasers_v2", "LeaderCorridor_lasers_v2", "LeaderCorridor_lasers",
                                                   "LeaderCorridor_lasers_compas")]
    scen = scenario_of(g)
    out = {"scen:" + k: v for k, v in scen.items()}
    for k, v in obs_record(g, obs0, laser_names).items():
        out["reset:" + k] = v
    for k, v in debug_record(g).items():
        out["reset_dbg:" + k] = v
    if wrap is not None:      # observation() is a pure function of the obs dict (+ the sensors' laser_length)
        out["reset:wrap_sensorPrev"] = np.asarray(wrap.observation(obs0)).copy()
    rng = np.random.default_rng(1000 + seed)
    acts, rews, dones, infos, raws = [], [], [], [], []
    obs_rows = {}
    dbg_rows = {}
    after_done = 0
    for t in range(n_steps):
        if policy == "chase":
            a = chase_action(g, rng, 0.05)
        elif policy == "chase_noisy":
            a = chase_action(g, rng, 0.5)
        elif policy == "random":
            a = random_action(g, rng)
        elif policy in ("ram", "ram_rocks"):
            a = ram_action(g, rng, policy == "ram_rocks")
        elif policy == "straight":
            a = (0.225 * g.follower.max_speed / 0.25, 0.0)
        else:
            raise ValueError(policy)
        kind = r.cfg.get("action")
        if kind == "discrete":                  # nearest entry of the Discrete(5) table (ENV:362-367); random policy: any entry
            table = g.discrete_rotation_speed_to_value
            k = int(rng.integers(5)) if policy == "random" else min(table, key=lambda i: abs(table[i] - a[1]))
            raws.append(k)
            obs, rew, done, info = r.step(k)
            a = (float(g.follower.max_speed), float(table[k]))       # what ENV:922 turns it into
        elif kind == "turn":
            w = np.float32(a[1])
            raws.append(float(w))
            obs, rew, done, info = r.step(w)
            a = (0.25, float(w))                                       # np.concatenate([[0.25], action]), ENV:924-925
        else:
            obs, rew, done, info = r.step(a)
        acts.append(a)
        rews.append(float(rew))
        dones.append(bool(done))
        infos.append([MISSION[info["mission_status"]], AGENT[info["agent_status"]], LEADER[info["leader_status"]]])
        for k, v in obs_record(g, obs, laser_names).items():
            obs_rows.setdefault(k, []).append(v)
        if wrap is not None:
            obs_rows.setdefault("wrap_sensorPrev", []).append(np.asarray(wrap.observation(obs)).copy())
        if t % debug_every == 0:
            for k, v in debug_record(g).items():
                dbg_rows.setdefault(k, []).append(v)
        if done:
            after_done += 1
            if after_done >= stop_after_done:  # the reference keeps simulating after done (ENV:935-936)
                break
    out["actions"] = np.array(acts, dtype=np.float64)
    if raws:
        out["actions_raw"] = np.array(raws, dtype=np.int32 if r.cfg.get("action") == "discrete" else np.float32)
    out["reward"] = np.array(rews, dtype=np.float64)
    out["done"] = np.array(dones, dtype=np.uint8)
    out["info"] = np.array(infos, dtype=np.uint8)
    for k, v in obs_rows.items():
        out["obs:" + k] = np.stack(v)
    for k, v in dbg_rows.items():
        if k in ("hist", "corr", "hist_isf64"):
            # ragged: pad to max length
            m = max(len(x) for x in v)
            w = v[0].shape[1:] if v[0].ndim > 1 else ()
            arr = np.full((len(v), m) + tuple(w), np.nan if k != "hist_isf64" else 255,
                          dtype=np.float64 if k != "hist_isf64" else np.uint8)
            for i, x in enumerate(v):
                arr[i, :len(x)] = x
            out["dbg:" + k] = arr
        else:
            out["dbg:" + k] = np.stack(v)
    meta = dict(config=config_name, seed=seed, policy=policy, n_steps=len(acts), debug_every=debug_every,
                kwargs=json.loads(json.dumps(CONFIGS[config_name]["kwargs"], default=str)),
                post=CONFIGS[config_name]["post"], laser_names=laser_names,
                numpy=np.__version__, tick="frame k sees get_ticks()==k")
    out["meta"] = np.array(json.dumps(meta))
    return out


# (name, config, seed, policy, steps)
EPISODES = [
    ("B_s12_straight", "B", 12, "straight", 300),
    ("B_s1_chase", "B", 1, "chase", 520),
    ("B_s3_chase_noisy", "B", 3, "chase_noisy", 300),
    ("B_s5_random", "B", 5, "random", 200),
    ("B_s7_random", "B", 7, "random", 200),
    ("B_s21_chase", "B", 21, "chase", 300),
    ("A_s0_chase", "A", 0, "chase", 200),
    ("A_s4_random", "A", 4, "random", 150),
    ("Bes_s2_random", "B_es", 2, "random", 200),
    ("Bes_s6_chase", "B_es", 6, "chase_noisy", 200),
    ("B3_s8_chase", "B3", 8, "chase", 250),
    ("Bastar_s1_chase", "B_astar", 1, "chase", 150),
    ("Bastar_s6_random", "B_astar", 6, "random", 100),
    ("Bastar_s14_chase", "B_astar", 14, "chase", 60),
    ("B6_s2_chase", "B6", 2, "chase", 250),
    ("B6_s9_random", "B6", 9, "random", 120),
    ("D_s2_chase", "D", 2, "chase", 60),
    ("D_s7_random", "D", 7, "random", 60),
    ("M_s3_chase", "M", 3, "chase", 100),
    ("C_s1_chase", "C", 1, "chase", 150),
    ("C_s5_random", "C", 5, "random", 100),
    ("L_s2_chase", "L", 2, "chase", 150),
    ("L_s7_random", "L", 7, "random", 100),
    ("Lall_s2_chase", "L_all", 2, "chase", 150),
    ("Lall_s7_ram", "L_all", 7, "ram_rocks", 120),
    ("T_s3_chase", "T", 3, "chase", 200),
    ("T_s9_random", "T", 9, "random", 100),
    ("Bshort_s4_chase", "B_short", 4, "chase", 60),
    ("Bshort_s9_random", "B_short", 9, "random", 60),
    ("Bnobear_s1_chase", "B_nobear", 1, "chase", 520),
    ("Bnobear_s2_chase", "B_nobear", 2, "chase", 520),
    ("Bnobear_s5_chase", "B_nobear", 5, "chase", 520),
    ("Bpad_s4_chase", "B_pad", 4, "chase", 150),
    ("E_s3_chase", "E", 3, "chase", 700),
    ("E_s5_random", "E", 5, "random", 200),
    ("E_s8_chase", "E", 8, "chase_noisy", 400),
    ("G_s2_chase", "G", 2, "chase", 120),
    ("G_s5_random", "G", 5, "random", 120),
    ("F_s1_chase", "F", 1, "chase", 120),
    ("F_s6_random", "F", 6, "random", 60),
    ("F_s7_chase", "F", 7, "chase_noisy", 100),
    ("N_s3_chase", "N", 3, "chase", 200),
    ("N_s8_random", "N", 8, "random", 120),
    ("Bcfs_s2_chase", "B_cfs", 2, "chase", 150),
    ("Bcfs_s5_random", "B_cfs", 5, "random", 100),
    ("Bnocoll_s4_ram", "B_nocoll", 4, "ram_rocks", 300),
    ("Bnocoll_s7_ram", "B_nocoll", 7, "ram", 200),
    ("Bagg_s2_chase", "B_agg", 2, "chase", 150),
    ("Bagg_s6_random", "B_agg", 6, "random", 120),
    ("Bnoobst_s1_chase", "B_noobst", 1, "chase", 300),
    ("Bnoobst_s5_random", "B_noobst", 5, "random", 120),
    ("Bmep_s2_chase", "B_mep", 2, "chase", 520),
    ("Bmep_s11_noisy", "B_mep", 11, "chase_noisy", 200),
    ("Btraj_s3_chase", "B_traj", 3, "chase", 400),
    ("Btraj_s6_random", "B_traj", 6, "random", 150),
]


def one_episode(args):
    name, cfg, seed, policy, steps = args
    out = run_episode(cfg, seed, policy, steps)
    path = os.path.join(GOLDEN, "episode_%s.npz" % name)
    np.savez_compressed(path, **out)
    return name, int(out["done"].sum()), len(out["reward"]), float(out["reward"].sum())


def pool_worker(args):
    config_name, seeds = args
    r = Runner(config_name)
    recs = []
    for s in seeds:
        r.reset(s)
        sc = scenario_of(r.game)
        if not bool(sc["found_target_point"]) or bool(sc["done_at_reset"]):
            continue
        sc["seed"] = np.array(s)
        recs.append(sc)
    return recs


def make_pool(config_name, n_seeds, jobs, seed0=0):
    """Scenario pool for bench.py / full-size property tests: post-reset scenarios of the reference
    for python seeds seed0..seed0+n_seeds-1 (D*-failed scenarios dropped)."""
    import multiprocessing as mp
    chunks = [list(range(seed0 + i, seed0 + n_seeds, jobs)) for i in range(jobs)]
    with mp.Pool(jobs) as p:
        parts = p.map(pool_worker, [(config_name, c) for c in chunks])
    recs = sorted([r for part in parts for r in part], key=lambda r: int(r["seed"]))
    P = len(recs)
    S = recs[0]["static_rects"].shape[0]
    R = recs[0]["robot_pos"].shape[0]
    wmax = max(r["route"].shape[0] for r in recs)
    tmax = max(r["init_traj"].shape[0] for r in recs)
    out = dict(seed=np.array([int(r["seed"]) for r in recs], dtype=np.int32),
               static_rects=np.stack([r["static_rects"] for r in recs]).astype(np.int16 if True else np.int32),
               robot_pos=np.stack([r["robot_pos"] for r in recs]),
               robot_dir=np.stack([r["robot_f64"][:, 0] for r in recs]),
               robot_rect=np.stack([r["robot_i32"][:, :4] for r in recs]).astype(np.int16),
               route=np.zeros((P, wmax, 2), dtype=np.int16), route_len=np.zeros(P, dtype=np.int32),
               init_traj=np.zeros((P, tmax, 2), dtype=np.float32), init_traj_len=np.zeros(P, dtype=np.int32))
    for i, r in enumerate(recs):
        assert np.all(r["route"] == np.round(r["route"])) and np.abs(r["route"]).max() < 32000
        out["route"][i, :len(r["route"])] = r["route"]
        out["route_len"][i] = len(r["route"])
        out["init_traj"][i, :len(r["init_traj"])] = r["init_traj"]
        out["init_traj_len"][i] = len(r["init_traj"])
    meta = dict(config=config_name, n_seeds=n_seeds, seed0=seed0, kept=P, S=S, R=R,
                kwargs=json.loads(json.dumps(CONFIGS[config_name]["kwargs"], default=str)))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(GOLDEN, "pool_%s.npz" % config_name)
    np.savez_compressed(path, **out)
    print("pool", config_name, "kept", P, "of", n_seeds, "->", path, os.path.getsize(path), "bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--pool", type=int, default=0, help="also (re)generate scenario pools with this many seeds")
    ap.add_argument("--pool-config", default="B")
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--no-episodes", action="store_true")
    a = ap.parse_args()
    if not a.no_episodes:
        todo = [e for e in EPISODES if a.only is None or a.only in e[0]]
        import multiprocessing as mp
        with mp.Pool(min(a.jobs, len(todo))) as p:
            for name, ndone, n, ret in p.imap_unordered(one_episode, todo):
                print("episode %-20s steps=%4d done_steps=%d return=%.2f" % (name, n, ndone, ret), flush=True)
    if a.pool:
        make_pool(a.pool_config, a.pool, a.jobs)


if __name__ == "__main__":
    main()

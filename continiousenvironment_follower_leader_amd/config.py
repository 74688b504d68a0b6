"""Host-side mirror of ``Game.__init__`` (reference ``follow_the_leader_continuous_env.py:45-417``):
same keyword arguments, defaults, unit conversion (metres -> px, per-second -> per-frame with
``AVG_FRAMES_PER_SECOND = 100``, ENV:38, 330-357) and error behaviour, producing the frozen
``ftl_config`` the C-ABI takes.  Nothing here computes on the hot path."""
from collections import OrderedDict
from dataclasses import dataclass, field
from warnings import warn

from . import abi

AVG_FRAMES_PER_SECOND = 100  # ENV:38

# Reward dataclass defaults (utils/reward_constructor.py:4-16); Game overrides leader_movement_reward=0 (ENV:279)
REWARD_DEFAULTS = dict(reward_in_box=1.0, reward_on_track=0.1, reward_in_dev=0.5, leader_movement_reward=1.0,
                       crash_penalty=-10.0, not_on_track_penalty=-1.0, too_close_penalty=-5.0,
                       leader_stop_penalty=-1.0)

# SENSOR_CLASSNAME_TO_CLASS (utils/sensors.py:1291-1307): the classes on the accelerated path.
SUPPORTED_SENSOR_CLASSES = ("LeaderPositionsTracker_v2", "LeaderCorridor_Prev_lasers_v2", "LeaderCorridor_lasers_v2", "LeaderCorridor_lasers", "FollowerInfo",
                            "LeaderCorridor_lasers_compas", "LaserSensor", "LeaderPositionsTracker", "LeaderTrackDetector_vector",
                            "LeaderTrackDetector_radar")
# classes whose constructor (or scan) raises in the reference itself: exception type and message mirrored (SEN:493-495, 810-864, 993)
DEPRECATED_SENSOR_CLASSES = {
    "GreenBoxBorderSensor": (ValueError, "To use it, you need to uncomment the call self._get_green_zone_border_points(). Commented out because it slows down the simulation"),
    "LeaderObstacles_lasers": (ValueError, "Deprecated class, use LeaderCorridor_lasers_v2 with flags insteadreact_to_safe_corridor=False and react_to_green_zone=False"),
    "Leader_Dyn_Obstacles_lasers": (ValueError, "Deprecated class, use LeaderCorridor_lasers_v2 with flags insteadreact_to_safe_corridor=False and react_to_green_zone=False, react_to_obstacles='dynamic'"),
    # (constructs in the reference; its scan raises at the first use_sensors, i.e. inside the first reset(), SEN:993)
    "LeaderCorridor_Prev_lasers_v3": (ValueError, "The sensor ignores all points inside the corridor, including obstacles, this is an error"),
    "LaserPrevSensor": (TypeError, "This is an obsolete class, you should use LeaderCorridor_Prev_lasers_v2 with flags insteadreact_to_safe_corridor=False and react_to_green_zone=False, react_to_obstacles=True and first_laser_angle_offset=0"),
}
KNOWN_SENSOR_CLASSES = ("LaserSensor", "LeaderPositionsTracker", "LeaderPositionsTracker_v2",
                        "LeaderTrackDetector_vector", "LeaderTrackDetector_radar", "LeaderCorridor_lasers",
                        "GreenBoxBorderSensor", "LeaderCorridor_lasers_v2", "LeaderObstacles_lasers",
                        "Leader_Dyn_Obstacles_lasers", "FollowerInfo", "LaserPrevSensor",
                        "LeaderCorridor_Prev_lasers_v2", "LeaderCorridor_Prev_lasers_v3",
                        "LeaderCorridor_lasers_compas")


@dataclass
class LaserSpec:
    name: str
    count: int
    length: float
    react_corridor: bool
    react_green: bool
    react_obstacles: int          # 0 False, 1 True/"all", 2 "static", 3 "dynamic"
    history: int
    angle_offset: float
    after_tracker: bool
    pad_sectors: bool = False
    out_offset: int = 0
    lenient: bool = False       # LeaderCorridor_lasers_v2: flat [count] observation, no error on a short corridor
    ray_angles: tuple = None    # LeaderCorridor_lasers: explicit ray directions relative to the heading, deg
    in_policy_obs: bool = False  # one of the classes ContinuousObserveModifier_sensorPrev concatenates (wrappers.py:204, 214)
    compas: bool = False        # LeaderCorridor_lasers_compas: rows of 5*count (no wall | front | back | left | right), SEN:1138-1288

    @property
    def width(self):            # row width of the sensor's output block (SEN:932-958, 1226)
        if self.compas:
            return 5 * self.count
        return 4 * self.count if self.pad_sectors else self.count


@dataclass
class AuxSpec:
    """LaserSensor / LeaderTrackDetector_vector / LeaderTrackDetector_radar: a float32 block after the ray sensors' blocks."""
    name: str
    kind: int
    shape: tuple                # shape of the observation the reference returns
    after_tracker: bool
    out_offset: int = 0
    params: dict = field(default_factory=dict)

    @property
    def out_len(self):
        n = 1
        for d in self.shape:
            n *= d
        return n


@dataclass
class GameConfig:
    """Parsed constructor arguments + the ctypes struct."""
    kwargs: dict
    c: abi.Config
    lasers: list = field(default_factory=list)
    tracker_name: str = None
    sensor_order: list = field(default_factory=list)
    follower_info: list = field(default_factory=list)       # (name, speed_direction_param) of FollowerInfo sensors
    aux: list = field(default_factory=list)                 # AuxSpec of lidar / leader-track detectors, dict order
    discrete_action_space: bool = False
    constant_follower_speed: bool = False
    discrete_rotation_speed_to_value: dict = None
    pixels_to_meter: float = 50
    action_low: tuple = None
    action_high: tuple = None

    @property
    def n_robots(self):
        return 2 + self.c.n_bears

    @property
    def lasers_len(self):
        return sum(l.history * l.width for l in self.lasers) + sum(a.out_len for a in self.aux)


def _react_code(v):
    if v is False or v is None:
        return 0
    if v is True or v == "all":
        return 1
    if v == "static":
        return 2
    if v == "dynamic":
        return 3
    # the reference builds a ValueError here without raising it and then fails on an unbound name (SEN:661-664)
    raise ValueError("You need to specify which obstacles the sensor should respond to. Set react_to_obstacles "
                     "equal to one of the values: True, 'all', 'dynamic', 'static'")


def make_config(game_width=1500, game_height=1000, framerate=500, frames_per_step=10,
                random_frames_per_step=None, caption=None, trajectory=None, leader_pos_epsilon=25,
                show_leader_path_flag=True, show_leader_trajectory_flag=True, show_rectangles_flag=True,
                show_box_flag=True, show_objects_flag=True, show_sensors_flag=True,
                simulation_time_limit=None, reward_config=None, pixels_to_meter=50,
                min_distance=1, max_distance=4, max_dev=1, warm_start=500, manual_control=False,
                manual_control_input="keyboard", max_steps=5000, aggregate_reward=False,
                add_obstacles=True, add_bear=True, bear_number=3, multi_random_bears=False,
                move_bear_v4=True, obstacle_number=35, bear_behind=False, step_grid=10,
                early_stopping=None, follower_sensors=None, leader_speed_regime=None,
                leader_acceleration_regime=None, discrete_action_space=False,
                constant_follower_speed=False, path_finding_algorythm="dstar",
                multiple_end_points=False, negative_speed=False, follower_max_speed=0.5,
                leader_max_speed=0.5, follower_max_rotation_speed=57.296,
                leader_max_rotation_speed=57.296, follower_acceleration=0.005,
                leader_acceleration=0.005, bear_max_speed=1.1, follower_size=(0.5, 0.35),
                leader_size=(0.38, 0.52), bear_size=(0.5, 0.5), bridge_size=(80, 40),
                return_render_matrix=True, ignore_follower_collisions=False,
                path_finding_iterations=15000, leader_margin=1.5,
                # capacities of the fixed per-env slots of the batched state (no reference equivalent)
                traj_cap=None, corr_cap=None, route_cap=128, init_traj_cap=None, n_static=None,
                # stream of the per-frame uniform draws that replace the global `random` (ENV:1156)
                rng_seed=0, env_id_base=0,
                **kwargs):
    """Same signature and defaults as ``Game.__init__`` (ENV:45-105); unknown kwargs are swallowed like the
    reference's ``**kwargs`` (ENV:104)."""
    early_stopping = {} if early_stopping is None else early_stopping
    follower_sensors = {} if follower_sensors is None else follower_sensors
    all_kwargs = dict(locals())
    all_kwargs.pop("kwargs")

    # ---- error behaviour of the reference constructor -------------------------------------------
    if multiple_end_points and path_finding_algorythm != "dstar":  # ENV:239-243
        raise NotImplementedError("Only dstar pathfinding function supports multiple end points. "
                                  "multiple_end_points must be False or diggerent path_finding_algorythm "
                                  "must be chosen")
    if path_finding_algorythm not in ["astar", "dstar"]:  # ENV:423-425
        raise ValueError("path_finding_algorythm {} not in list:{}".format(path_finding_algorythm, ["astar", "dstar"]))
    if add_bear and bear_number <= 0:  # ENV:426-427
        raise ValueError("Add bear is true, but number of bears is not greater then 0")
    if random_frames_per_step is not None and frames_per_step is not None:
        assert len(random_frames_per_step) == 2, \
            "random frames per step должен быть задан в виде границ для генерации случайных значений. " \
            "Задано: {}".format(random_frames_per_step)  # ENV:402-404

    # ---- features of the reference that are outside the accelerated hot path ----------------------
    if manual_control:
        raise NotImplementedError("manual_control (pygame event loop, ENV:913-915) is not part of the batched step path")
    if random_frames_per_step is not None and frames_per_step is None:
        raise NotImplementedError("random_frames_per_step without frames_per_step leaves the reference's first step "
                                  "without a frame count (ENV:399-405)")
    if leader_speed_regime is not None and type(leader_speed_regime) not in (dict, OrderedDict):
        warn("leader_speed_regime должен быть dict или OrderedDict, получено: {}, будет проигнорировано".format(
            type(leader_speed_regime)))  # ENV:387-389
    if leader_acceleration_regime is not None and type(leader_acceleration_regime) not in (dict, OrderedDict):
        warn("leader_acceleration_regime должен быть dict, получено: {}, будет проигнорировано".format(
            type(leader_acceleration_regime)))
    if simulation_time_limit is not None:
        raise NotImplementedError("simulation_time_limit depends on the wall clock (ENV:1119-1123)")
    if add_bear and bear_number > abi.FTL_MAX_BEARS:
        raise NotImplementedError("at most %d bears (2 + bears robots share the 8 lanes of an env's group)" % abi.FTL_MAX_BEARS)
    if add_bear and bear_number > 5 and move_bear_v4 and (max_distance * pixels_to_meter) != int(max_distance * pixels_to_meter):
        raise ValueError("non-integer arg 1 for randrange()")      # ENV:753: random.randrange(self.max_distance, ..) with a fractional float

    def to_px(m):  # ENV:1942-1943
        return m * pixels_to_meter

    c = abi.Config()
    c.abi_version = abi.FTL_ABI_VERSION
    c.width, c.height = int(game_width), int(game_height)
    c.frames_per_step = int(frames_per_step)
    if random_frames_per_step is not None:                  # ENV:399-405, 939-940: np.random.randint(lo, hi) per step,
        warn("Одновременно заданы и random_frames_per_step и frames_per_step, будет использоваться random_frames_per_step")
        c.rand_fps_lo, c.rand_fps_hi = int(random_frames_per_step[0]), int(random_frames_per_step[1])   # drawn from the
        if not (0 < c.rand_fps_lo < c.rand_fps_hi):                                                     # per-env counter stream
            raise ValueError("random_frames_per_step must be (low, high) with 0 < low < high")           # np.random.randint: low >= high
    c.max_steps = int(max_steps)
    c.warm_start = int(warm_start)
    c.trajectory_saving_period = 5  # ENV:262
    n_obst = obstacle_number if add_obstacles else 0  # ENV:322-323
    c.n_static = (n_obst + 2 if add_obstacles else 0) if n_static is None else int(n_static)  # two bridge walls + rocks
    c.n_bears = int(bear_number) if add_bear else 0
    c.move_bear_v4 = int(bool(move_bear_v4))
    c.ignore_follower_collisions = int(bool(ignore_follower_collisions))
    c.aggregate_reward = int(bool(aggregate_reward))
    c.has_low_reward = int("low_reward" in early_stopping)
    c.low_reward = float(early_stopping.get("low_reward", 0.0))
    c.has_max_distance_coef = int("max_distance_coef" in early_stopping)
    c.max_distance_coef = float(early_stopping.get("max_distance_coef", 0.0))
    c.min_distance = to_px(min_distance)
    c.max_distance = to_px(max_distance)
    c.max_dev = to_px(max_dev)
    c.leader_pos_epsilon = leader_pos_epsilon

    rw = dict(REWARD_DEFAULTS)
    if reward_config:
        import json
        with open(reward_config, "r") as fh:  # Reward.from_json, reward_constructor.py:21-26
            d = json.load(fh)
        d.pop("name", None)
        rw.update(d)
    else:
        rw["leader_movement_reward"] = 0  # ENV:279
    for k in ("reward_in_box", "reward_on_track", "reward_in_dev", "not_on_track_penalty", "crash_penalty",
              "too_close_penalty", "leader_movement_reward"):
        setattr(c, k, float(rw[k]))

    # ---- robots (ENV:330-357, 556-566, 578-589, 704-714) -----------------------------------------
    def fill(p, min_speed, max_speed, max_rot, acc, w, h):
        p.min_speed, p.max_speed = float(min_speed), float(max_speed)
        p.max_rotation_speed = float(max_rot)
        p.max_speed_change = float(acc)
        p.max_rotation_speed_change = 20 / 100
        p.img_w, p.img_h = int(w), int(h)  # pygame.transform.scale truncates (CLS:42)

    f_max = to_px(follower_max_speed) / AVG_FRAMES_PER_SECOND
    f_min = -(to_px(follower_max_speed) / AVG_FRAMES_PER_SECOND) if negative_speed else 0
    fill(c.follower, f_min, f_max, follower_max_rotation_speed / AVG_FRAMES_PER_SECOND,
         to_px(follower_acceleration) / AVG_FRAMES_PER_SECOND,
         to_px(follower_size[1]), to_px(follower_size[0]))            # height=size[0], width=size[1] (ENV:333-334)
    l_max = to_px(leader_max_speed) / AVG_FRAMES_PER_SECOND
    fill(c.leader, 0, l_max, leader_max_rotation_speed / AVG_FRAMES_PER_SECOND,
         to_px(leader_acceleration) / AVG_FRAMES_PER_SECOND,
         to_px(leader_size[0]), to_px(leader_size[1]))                # width=size[0], height=size[1] (ENV:352-353)
    fill(c.bear, 0, bear_max_speed * l_max, leader_max_rotation_speed / AVG_FRAMES_PER_SECOND,
         to_px(0.005), to_px(bear_size[1]), to_px(bear_size[0]))      # ENV:706-711

    # ---- sensors (CLS:239-253 registry protocol, dict order matters: CLS:269-286) ------------------
    lasers, tracker_name, order, follower_info, aux = [], None, [], [], []
    seen_tracker = False
    trk1_generate_corridor = True
    for name, spec in follower_sensors.items():
        spec = dict(spec)
        cls = spec.pop("sensor_class", None)
        if cls is None:
            if name in KNOWN_SENSOR_CLASSES:
                cls = name
            else:
                raise ValueError(f"Sensor class is undefined: {name}")  # CLS:249
        if cls not in KNOWN_SENSOR_CLASSES:
            raise KeyError(cls)  # SENSOR_CLASSNAME_TO_CLASS[...] lookup, CLS:245
        if cls in DEPRECATED_SENSOR_CLASSES:
            exc, msg = DEPRECATED_SENSOR_CLASSES[cls]
            raise exc(msg)
        if cls not in SUPPORTED_SENSOR_CLASSES:
            raise NotImplementedError(f"sensor class {cls} is outside the accelerated hot path "
                                      f"(supported: {SUPPORTED_SENSOR_CLASSES})")
        order.append((name, cls))
        if cls == "LeaderPositionsTracker_v2":
            if name not in ("LeaderPositionsTracker_v2", "LeaderPositionsTracker"):
                # use_sensors finds the tracker by these literal keys (CLS:257-267); under either of them the v2 class
                # is scanned once up front and once more at its dict position (its type name is not the one skipped
                # at CLS:270) -- the shipped training configs register it as "LeaderPositionsTracker"
                raise NotImplementedError("the tracker must be registered under the key 'LeaderPositionsTracker_v2' or "
                                          "'LeaderPositionsTracker' (CLS:257-267 look it up by name)")
            if spec.get("eat_close_points", True):
                # inherited default eat_close_points=True is never applied by the v2 scan (SEN:243-327)
                pass
            if not spec.get("generate_corridor", True):
                raise NotImplementedError("generate_corridor=False leaves `leader_corridor` unbound for the ray sensors")
            if c.has_tracker:
                raise NotImplementedError("one tracker per follower (with both registered the v2 history silently replaces the v1 one, CLS:257-267)")
            c.has_tracker = 2
            c.tracker_saving_period = int(spec.get("saving_period", 5))
            c.tracker_start_behind = int(bool(spec.get("start_corridor_behind_follower", False)))
            c.corridor_length = float(spec["corridor_length"])   # required keyword-only args (SEN:238)
            c.corridor_width = float(spec["corridor_width"])
            tracker_name = name
            seen_tracker = True
        elif cls == "LeaderPositionsTracker":        # SEN:148-229, deprecated v1: looked up by this literal key and scanned ONCE per step
            if name != "LeaderPositionsTracker":     # (CLS:257-261; its own dict entry is skipped, CLS:269-270, so it never shows up in the obs)
                raise NotImplementedError("the v1 tracker must be registered under the key 'LeaderPositionsTracker' (CLS:257 looks it up by name)")
            if c.has_tracker:
                raise NotImplementedError("one tracker per follower (with both registered the v2 history silently replaces the v1 one, CLS:257-267)")
            if not spec.get("generate_corridor", True) and any(True for _ in ()):
                pass
            c.has_tracker = 1
            c.tracker_saving_period = int(spec.get("saving_period", 5))
            c.trk1_eat_close_points = int(bool(spec.get("eat_close_points", True)))
            c.corridor_length, c.corridor_width = 0.0, float(max_dev * pixels_to_meter)      # half-width = env.max_dev (SEN:189)
            trk1_generate_corridor = bool(spec.get("generate_corridor", True))
            tracker_name = name
            seen_tracker = False                     # there is no second scan: every sensor sees the state after the one up-front scan
        elif cls == "LaserSensor":                   # SEN:18-145
            aa = min(360, spec.get("available_angle", 360))
            step = spec.get("angle_step", 10)
            border, cur, n_ang = int(aa / 2), 0, 1
            while cur < border:                      # SEN:95-101
                cur += step
                n_ang += 2
            only_d = bool(spec.get("return_only_distances", False))
            all_pts = bool(spec.get("return_all_points", False))
            rng_px = spec.get("sensor_range", 5) * pixels_to_meter
            npts = int(spec.get("points_number", 20))
            # return_all_points (SEN:112-113, 131-134): the scan returns every marching point up to the first hit of every ray -- K rows, K
            # changing from call to call; the batched block is [K][K points or distances][zeros] (the facade cuts it back to K rows)
            shape = ((1 + n_ang * npts * (1 if only_d else 2),) if all_pts else ((n_ang,) if only_d else (n_ang, 2)))
            aux.append(AuxSpec(name=name, kind=abi.AUX_LIDAR, shape=shape, after_tracker=seen_tracker,
                               params=dict(n_angles=n_ang, points_number=npts, return_only_distances=int(only_d), return_all_points=int(all_pts),
                                           range_px=float(rng_px), in_range_px=float(rng_px + 3 * pixels_to_meter), angle_step=float(step),
                                           border_angle=border)))
        elif cls in ("LeaderTrackDetector_vector", "LeaderTrackDetector_radar"):       # SEN:342-487
            if name in ("LeaderTrackDetector_vector", "LeaderTrackDetector_radar") and \
                    "LeaderPositionsTracker" not in follower_sensors and "LeaderPositionsTracker_v2" not in follower_sensors:
                raise ValueError("Sensor {} requires sensor LeaderPositionsTracker for tracking leader movement.".format(name))   # CLS:240-243
            L = int(spec.get("position_sequence_length", 100))
            if cls == "LeaderTrackDetector_vector":
                det = spec.get("detectable_positions", "new")
                if det not in ("new", "old"):
                    raise UnboundLocalError("local variable 'vecs' referenced before assignment")      # SEN:368-379
                aux.append(AuxSpec(name=name, kind=abi.AUX_TRACK_VECTOR, shape=(L, 2), after_tracker=seen_tracker,
                                   params=dict(seq_len=L, detectable=("new", "old").index(det))))
            else:
                det = spec.get("detectable_positions", "old")
                if det not in ("new", "old", "near"):
                    raise UnboundLocalError("local variable 'chosen_dots' referenced before assignment")   # SEN:452-462
                ns = int(spec.get("radar_sectors_number", 180))
                aux.append(AuxSpec(name=name, kind=abi.AUX_TRACK_RADAR, shape=(ns,), after_tracker=seen_tracker,
                                   params=dict(seq_len=L, detectable=("new", "old", "near").index(det), radar_sectors=ns)))
        elif cls == "FollowerInfo":                  # SEN:822-845: [speed / max_speed, direction / 360], host-side from the state
            follower_info.append((name, int(spec.get("speed_direction_param", 2))))
        elif cls == "LeaderCorridor_lasers":         # SEN:571-702: 3 or 5 front rays, optionally 2 rear ones, current edges only
            front, back = int(spec.get("front_lasers_count", 3)), int(spec.get("back_lasers_count", 0))
            assert front in [3, 5]                   # SEN:596-597
            assert back in [0, 2]
            angles = (-40.0, 0.0, 40.0) + ((-90.0, 90.0) if front == 5 else ()) + ((-150.0, 150.0) if back == 2 else ())
            lasers.append(LaserSpec(name=name, count=front + back, length=float(spec.get("laser_length", 100)),
                                    react_corridor=bool(spec.get("react_to_safe_corridor", True)),
                                    react_green=bool(spec.get("react_to_green_zone", False)),
                                    react_obstacles=_react_code(spec.get("react_to_obstacles", False)),
                                    history=1, angle_offset=0.0, after_tracker=seen_tracker, pad_sectors=False, lenient=True,
                                    ray_angles=angles))
        elif cls == "LeaderCorridor_lasers_v2":      # SEN:736-807: the ray cast on the current edges only, ray 0 straight ahead
            n = int(spec.get("lasers_count", 12))
            if n not in (12, 24, 20, 36):
                raise ValueError("Invalid number of laser beams, should be 12,24,20 or 36")  # SEN:761-762
            lasers.append(LaserSpec(name=name, count=n, length=float(spec.get("laser_length", 100)),
                                    react_corridor=bool(spec.get("react_to_safe_corridor", True)),
                                    react_green=bool(spec.get("react_to_green_zone", False)),
                                    react_obstacles=_react_code(spec.get("react_to_obstacles", False)),
                                    history=1, angle_offset=0.0, after_tracker=seen_tracker, pad_sectors=False, lenient=True))
        elif cls == "LeaderCorridor_lasers_compas":  # SEN:1138-1288: Prev_lasers_v2's constructor + a flag check
            n = int(spec.get("lasers_count", 12))
            if n not in (12, 24, 20, 36):
                raise ValueError("Invalid number of laser beams, should be 12,24,20 or 36")  # SEN:761-762
            hist = spec.get("max_prev_obs", 0)
            assert hist > 0  # SEN:876
            if not spec.get("react_to_safe_corridor", True) or not spec.get("react_to_green_zone", False) or spec.get("react_to_obstacles", False):
                raise ValueError("Unsupported set of flags for LeaderCorridor_lasers_compas class, now implemented"
                                 "only option for flags: "
                                 "react_to_safe_corridor=True, react_to_green_zone=True, react_to_obstacles=False")   # SEN:1148-1151
            lasers.append(LaserSpec(name=name, count=n, length=float(spec.get("laser_length", 100)), react_corridor=True, react_green=True,
                                    react_obstacles=0, history=int(hist), angle_offset=float(spec.get("first_laser_angle_offset", -45)),
                                    after_tracker=seen_tracker, pad_sectors=False, in_policy_obs=True, compas=True))
        else:
            n = int(spec.get("lasers_count", 12))
            if n not in (12, 24, 20, 36) and not spec.get("_allow_any_lasers_count", False):
                raise ValueError("Invalid number of laser beams, should be 12,24,20 or 36")  # SEN:761-762
            hist = spec.get("max_prev_obs", 0)
            assert hist > 0  # SEN:876
            lasers.append(LaserSpec(name=name, count=n, length=float(spec.get("laser_length", 100)),
                                    react_corridor=bool(spec.get("react_to_safe_corridor", True)),
                                    react_green=bool(spec.get("react_to_green_zone", False)),
                                    react_obstacles=_react_code(spec.get("react_to_obstacles", False)),
                                    history=int(hist),
                                    angle_offset=float(spec.get("first_laser_angle_offset", -45)),
                                    after_tracker=seen_tracker, pad_sectors=bool(spec.get("pad_sectors", True)),
                                    in_policy_obs=True))
    if lasers and not c.has_tracker:
        raise NotImplementedError("ray sensors need a leader-positions tracker (reference: NameError on "
                                  "`leader_corridor`, CLS:280)")
    if lasers and c.has_tracker == 1 and not trk1_generate_corridor:
        raise NotImplementedError("generate_corridor=False leaves `leader_corridor` unbound for the ray sensors")
    if any(a.kind != abi.AUX_LIDAR for a in aux) and not c.has_tracker:
        raise NameError("name 'leader_positions_hist' is not defined")      # CLS:272: a detector without any tracker
    if len(aux) > abi.FTL_MAX_AUX:
        raise NotImplementedError(f"at most {abi.FTL_MAX_AUX} lidar / detector sensors")
    if len(lasers) > abi.FTL_MAX_LASERS:
        raise NotImplementedError(f"at most {abi.FTL_MAX_LASERS} ray sensors")
    c.n_lasers = len(lasers)
    off = 0
    for k, l in enumerate(lasers):
        l.out_offset = off
        lc = c.lasers[k]
        lc.count, lc.length, lc.history = l.count, l.length, l.history
        lc.react_corridor, lc.react_green, lc.react_obstacles = int(l.react_corridor), int(l.react_green), l.react_obstacles
        lc.angle_offset, lc.after_tracker, lc.out_offset = l.angle_offset, int(l.after_tracker), off
        lc.pad_sectors = int(l.pad_sectors)
        lc.lenient = int(l.lenient)
        lc.in_policy_obs = int(l.in_policy_obs)
        lc.compas = int(l.compas)
        lc.explicit_angles = int(l.ray_angles is not None)
        for i, a in enumerate(l.ray_angles or ()):
            lc.ray_angles[i] = a
        off += l.history * l.width
    c.n_aux = len(aux)
    for j, a in enumerate(aux):
        a.out_offset = off
        ac = c.aux[j]
        ac.kind, ac.after_tracker, ac.out_offset, ac.out_len = a.kind, int(a.after_tracker), off, a.out_len
        for k, v in a.params.items():
            setattr(ac, k, v)
        off += a.out_len
    c.trk1_eat_radius = float(max(to_px(follower_size[1]), to_px(follower_size[0])))      # max(host.width, host.height), SEN:213

    # ---- leader regimes (ENV:382-397): int(key) -> value in dict insertion order ---------------------------------
    c.n_speed_regime, c.n_acc_regime = -1, -1
    if type(leader_speed_regime) in (dict, OrderedDict):
        if len(leader_speed_regime) > abi.FTL_MAX_REGIME:
            raise NotImplementedError("at most %d leader_speed_regime entries" % abi.FTL_MAX_REGIME)
        c.n_speed_regime = len(leader_speed_regime)
        for i, (k, v) in enumerate(leader_speed_regime.items()):
            c.speed_key[i] = int(k)
            if type(v) in (tuple, list):                       # ENV:1155-1156: uniform(v[0], v[1]) every frame
                c.speed_is_range[i], c.speed_lo[i], c.speed_hi[i] = 1, float(v[0]), float(v[1])
            else:
                c.speed_is_range[i], c.speed_lo[i], c.speed_hi[i] = 0, float(v), float(v)
    if type(leader_acceleration_regime) in (dict, OrderedDict):
        if len(leader_acceleration_regime) > abi.FTL_MAX_REGIME:
            raise NotImplementedError("at most %d leader_acceleration_regime entries" % abi.FTL_MAX_REGIME)
        c.n_acc_regime = len(leader_acceleration_regime)
        for i, (k, v) in enumerate(leader_acceleration_regime.items()):
            c.acc_key[i], c.acc_val[i] = int(k), float(v)
    c.rng_seed = int(rng_seed) & ((1 << 64) - 1)
    c.env_id_base = int(env_id_base)

    # ---- capacities ---------------------------------------------------------------------------------
    # leader_factual_trajectory: initial int(dist/(5*v)) points, dist < 0.9*max_distance, then one point per
    # trajectory_saving_period frames up to max_steps (+ the frames of the step that crosses it).
    init_pts = int(0.9 * c.max_distance / (5 * l_max)) + 2
    c.init_traj_cap = int(init_traj_cap) if init_traj_cap else ((init_pts + 7) // 8) * 8
    need = c.init_traj_cap + (c.max_steps + 2 * max(c.frames_per_step, c.rand_fps_hi)) // 5 + 8
    c.traj_cap = int(traj_cap) if traj_cap else ((need + 63) // 64) * 64
    def pow2(v):
        return 1 << max(3, (int(v) - 1).bit_length())

    if corr_cap:
        c.corr_cap = pow2(corr_cap)       # the tracker rings are indexed with a mask
    elif c.has_tracker:
        # The corridor is trimmed by LENGTH (corridor_length), so its point count is length / spacing.  A point is saved every
        # tracker_saving_period scans = period/2 steps = period/2 * frames_per_step frames of leader motion (two scans per step);
        # seeded points are period*5*v apart.  A leader that always runs at its full speed gets 2.5x head-room on that count.  Speed /
        # acceleration regimes can slow it down without a bound (a multiplier drawn from [0, 0.5]): those configs keep the flat 4x
        # over the seeded spacing that the soak runs have been through.  Overflow is detected (FTL_ERR_CORR_OVERFLOW), never silent.
        # The ray kernel keeps a float32 copy of the ring in LDS, where 1.5 KB decide about a wavefront per SIMD: hence
        # no flat 4x for everybody.
        if c.n_speed_regime >= 0 or c.n_acc_regime >= 0:
            c.corr_cap = pow2(max(32, int(4 * c.corridor_length / max(c.tracker_saving_period * 5 * l_max, 1e-9))))
        else:
            fps_min = c.rand_fps_lo if c.rand_fps_hi > 0 else c.frames_per_step
            gap = c.tracker_saving_period * min(5.0, fps_min / 2.0) * l_max
            c.corr_cap = pow2(max(32, int(2.5 * c.corridor_length / max(gap, 1e-9))))
    else:
        c.corr_cap = 16
    c.route_cap = int(route_cap)
    # v1 tracker: one point per saving_period scans (one scan per step) + the follower's start position, never trimmed
    steps_max = c.max_steps // max(1, min(c.frames_per_step, c.rand_fps_lo or c.frames_per_step)) + 2
    c.hist1_cap = ((steps_max // max(c.tracker_saving_period, 1) + 8 + 7) // 8) * 8 if c.has_tracker == 1 else 8
    if c.has_tracker == 1 and not corr_cap:
        c.corr_cap = pow2(c.hist1_cap)      # the corridor gets one pair per saved point and is never trimmed either

    cfg = GameConfig(kwargs=all_kwargs, c=c, lasers=lasers, tracker_name=tracker_name, sensor_order=order, follower_info=follower_info, aux=aux,
                     discrete_action_space=bool(discrete_action_space),
                     constant_follower_speed=bool(constant_follower_speed), pixels_to_meter=pixels_to_meter)
    max_rot = c.follower.max_rotation_speed
    if discrete_action_space:  # ENV:360-367
        cfg.discrete_rotation_speed_to_value = {0: -max_rot, 1: -max_rot / 2, 2: 0, 3: max_rot / 2, 4: max_rot}
    elif constant_follower_speed:  # ENV:368-372
        cfg.action_low, cfg.action_high = (-max_rot,), (max_rot,)
    else:  # ENV:373-378
        cfg.action_low, cfg.action_high = (c.follower.min_speed, -max_rot), (c.follower.max_speed, max_rot)
    return cfg

"""ctypes mirror of ``include/ftl.h`` (the C-ABI of the HIP library).

Field order and types must match the header exactly; ``tests/test_abi.py`` checks the struct sizes
against ``ftl_sizeof_*`` exported by the library."""
import ctypes as C

FTL_ABI_VERSION = 4
FTL_MAX_BEARS = 6
FTL_MAX_LASERS = 4
FTL_MAX_AUX = 8
AUX_LIDAR, AUX_TRACK_VECTOR, AUX_TRACK_RADAR = 1, 2, 3
FTL_MAX_REGIME = 16
FTL_OBS_NUM = 10

FTL_OK = 0
FTL_E_INVALID, FTL_E_UNSUPPORTED, FTL_E_DEVICE, FTL_E_STATE = -1, -2, -3, -4

MISSION = ("in_progress", "fail", "success", "finished_by_time")          # ENV:951-955, 963, 1083, 1130
AGENT = ("moving", "crash", "low_reward", "too_far_from_leader", "finished")
LEADER = ("moving", "crash", "finished")

FTL_ERR_TRAJ_OVERFLOW, FTL_ERR_CORR_OVERFLOW, FTL_ERR_EMPTY_CORRIDOR, FTL_ERR_TRACKER_SEED, FTL_ERR_HIST1_OVERFLOW = 1, 2, 4, 8, 16
FTL_ERR_LIDAR_OVERFLOW = 32
FTL_ERR_BAD_ACTION = 64
FTL_ACTION_BOX2, FTL_ACTION_DISCRETE, FTL_ACTION_TURN = 0, 1, 2
FTL_STEP_AUTO_RESET = 1
FTL_N_METRICS = 8
FTL_METRICS_CLEAR = 1
(M_EPISODES, M_RETURN_SUM, M_FRAMES_SUM, M_SUCCESS, M_CRASH, M_LOW_REWARD, M_TOO_FAR, M_TIMEOUT) = range(8)

# env_int indices
(EI_SCEN, EI_TARGET_ID, EI_LEADER_FINISHED, EI_DONE, EI_CRASH, EI_IN_BOX, EI_ON_TRACE, EI_TOO_CLOSE,
 EI_STEP_COUNT, EI_FINISH_TIMER, EI_TRAJ_LEN, EI_TRK_COUNTER, EI_CORR_LO, EI_CORR_HI, EI_SEED_END,
 EI_SNAP_COUNT, EI_DYN_INDEX0, EI_DYN_INDEX1, EI_DYN_INDEX2, EI_DYN_INDEX3, EI_DYN_INDEX4, EI_DYN_INDEX5, EI_ERROR, EI_EPISODES,
 EI_GREEN_COUNT, EI_GREEN_LEN, EI_SCAN_OK, EI_SNAP_HEAD, EI_HINT, EI_GREEN_TINY, EI_RESETS, EI_ACC_CONSUMED,
 EI_HINT_X, EI_HINT_Y, EI_CLR_GREEN, EI_CLR_ALL, EI_FPS, EI_HW0_LO, EI_HW0_HI, EI_HIST1_LEN, EI_ERROR_STICKY, EI_COUNT) = range(42)
ED_ACC_PENALTY, ED_OVERALL_REWARD, ED_SPARE0, ED_SPARE1, ED_BEAR_POINTS = range(5)
ED_GREEN_W = ED_BEAR_POINTS + 2 * FTL_MAX_BEARS
ED_CUR_MULT, ED_CUR_ACC, ED_CUM_SPEED = ED_GREEN_W + 1, ED_GREEN_W + 2, ED_GREEN_W + 3
ED_COUNT = ED_CUM_SPEED + 1
RD_DIRECTION, RD_SPEED, RD_ROT_SPEED, RD_DES_SPEED, RD_DES_ROT_SPEED, RD_COUNT = range(6)
RI_X, RI_Y, RI_W, RI_H, RI_ROT_DIR, RI_DES_ROT_DIR, RI_SPARE0, RI_SPARE1, RI_COUNT = range(9)


def rand_frames(rng_seed, env_id, resets, step_count, lo, hi):
    """Twin of ftl_rand_frames (include/ftl.h): the stand-in for np.random.randint(lo, hi) at ENV:405/940."""
    v = lo + int(uniform01(rng_seed, env_id, resets, step_count | (1 << 40)) * (hi - lo))
    return v if v < hi else hi - 1


def rand_range(rng_seed, env_id, resets, frame, bear, k, start, stop):
    """Twin of ftl_rand_range (include/ftl.h): the stand-in for random.randrange(start, stop, 10) at ENV:753-754."""
    n = (stop - start + 9) // 10
    v = int(uniform01(rng_seed, env_id, resets, frame | (1 << 41) | (bear << 44) | (k << 48)) * n)
    return start + 10 * (v if v < n else n - 1)


class RobotParams(C.Structure):
    _fields_ = [("min_speed", C.c_double), ("max_speed", C.c_double), ("max_rotation_speed", C.c_double),
                ("max_speed_change", C.c_double), ("max_rotation_speed_change", C.c_double),
                ("img_w", C.c_int32), ("img_h", C.c_int32), ("_pad", C.c_int32 * 2)]


class LaserCfg(C.Structure):
    _fields_ = [("count", C.c_int32), ("react_corridor", C.c_int32), ("react_green", C.c_int32),
                ("react_obstacles", C.c_int32), ("history", C.c_int32), ("after_tracker", C.c_int32),
                ("out_offset", C.c_int32), ("pad_sectors", C.c_int32), ("lenient", C.c_int32), ("in_policy_obs", C.c_int32),
                ("length", C.c_double), ("angle_offset", C.c_double),
                ("explicit_angles", C.c_int32), ("compas", C.c_int32), ("ray_angles", C.c_double * 8)]


class AuxCfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("after_tracker", C.c_int32), ("out_offset", C.c_int32), ("out_len", C.c_int32),
                ("n_angles", C.c_int32), ("points_number", C.c_int32), ("return_all_points", C.c_int32),
                ("return_only_distances", C.c_int32), ("range_px", C.c_double), ("in_range_px", C.c_double),
                ("angle_step", C.c_double), ("border_angle", C.c_int32), ("seq_len", C.c_int32),
                ("detectable", C.c_int32), ("radar_sectors", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("frames_per_step", C.c_int32), ("max_steps", C.c_int32), ("warm_start", C.c_int32),
                ("trajectory_saving_period", C.c_int32), ("n_static", C.c_int32), ("n_bears", C.c_int32),
                ("move_bear_v4", C.c_int32), ("ignore_follower_collisions", C.c_int32),
                ("aggregate_reward", C.c_int32), ("has_low_reward", C.c_int32),
                ("has_max_distance_coef", C.c_int32), ("has_tracker", C.c_int32),
                ("tracker_saving_period", C.c_int32), ("tracker_start_behind", C.c_int32),
                ("n_lasers", C.c_int32), ("traj_cap", C.c_int32), ("corr_cap", C.c_int32),
                ("route_cap", C.c_int32), ("init_traj_cap", C.c_int32), ("hist1_cap", C.c_int32), ("trk1_eat_close_points", C.c_int32),
                ("low_reward", C.c_double), ("max_distance_coef", C.c_double),
                ("min_distance", C.c_double), ("max_distance", C.c_double), ("max_dev", C.c_double),
                ("leader_pos_epsilon", C.c_double), ("corridor_length", C.c_double),
                ("corridor_width", C.c_double),
                ("reward_in_box", C.c_double), ("reward_on_track", C.c_double), ("reward_in_dev", C.c_double),
                ("not_on_track_penalty", C.c_double), ("crash_penalty", C.c_double),
                ("too_close_penalty", C.c_double), ("leader_movement_reward", C.c_double),
                ("leader", RobotParams), ("follower", RobotParams), ("bear", RobotParams),
                ("lasers", LaserCfg * FTL_MAX_LASERS),
                ("n_speed_regime", C.c_int32), ("n_acc_regime", C.c_int32),
                ("speed_key", C.c_int32 * FTL_MAX_REGIME), ("speed_is_range", C.c_int32 * FTL_MAX_REGIME),
                ("acc_key", C.c_int32 * FTL_MAX_REGIME), ("env_id_base", C.c_int32),
                ("rand_fps_lo", C.c_int32), ("rand_fps_hi", C.c_int32), ("_pad1", C.c_int32),
                ("speed_lo", C.c_double * FTL_MAX_REGIME), ("speed_hi", C.c_double * FTL_MAX_REGIME),
                ("acc_val", C.c_double * FTL_MAX_REGIME), ("rng_seed", C.c_uint64),
                ("trk1_eat_radius", C.c_double), ("n_aux", C.c_int32), ("_pad2", C.c_int32), ("aux", AuxCfg * FTL_MAX_AUX)]


_M64 = (1 << 64) - 1


def mix64(x):
    x &= _M64
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & _M64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & _M64
    x ^= x >> 31
    return x


def uniform01(rng_seed, env_id, resets, frame):
    """Python twin of ftl_uniform01() in include/ftl.h (the stream that replaces random.uniform at ENV:1156)."""
    key = mix64(rng_seed + 0x9E3779B97F4A7C15 * (env_id + 1)) ^ mix64(0xD1B54A32D192ED03 * (resets + 1))
    return (mix64(key + 0x9E3779B97F4A7C15 * (frame + 1)) >> 11) * (1.0 / 9007199254740992.0)


class Scenarios(C.Structure):
    _fields_ = [("n_scenarios", C.c_int32), ("_pad", C.c_int32),
                ("static_rects", C.c_void_p), ("robot_pos", C.c_void_p), ("robot_dir", C.c_void_p),
                ("robot_rect", C.c_void_p), ("route", C.c_void_p), ("route_len", C.c_void_p),
                ("init_traj", C.c_void_p), ("init_traj_len", C.c_void_p)]


class ScenParams(C.Structure):
    """ftl_scen_params: what the reset-time scenario generator needs of Game(**kwargs)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("step_grid", C.c_int32), ("obstacle_number", C.c_int32),
                ("add_obstacles", C.c_int32), ("add_bear", C.c_int32), ("bear_number", C.c_int32), ("bear_behind", C.c_int32),
                ("multiple_end_points", C.c_int32), ("path_finding_iterations", C.c_int32),
                ("bridge_gap", C.c_int32), ("bridge_width", C.c_int32),
                ("trajectory_saving_period", C.c_int32), ("planner", C.c_int32),
                ("min_distance", C.c_double), ("max_distance", C.c_double),
                ("leader_pos_epsilon", C.c_double), ("leader_margin", C.c_double),
                ("leader_w", C.c_double), ("leader_h", C.c_double), ("leader_max_speed", C.c_double),
                ("fixed_route", C.c_void_p), ("fixed_route_len", C.c_int32), ("_pad", C.c_int32)]


SCEN_FOUND, SCEN_DONE_AT_RESET, SCEN_ROUTE_OVERFLOW, SCEN_TRAJ_OVERFLOW, SCEN_REF_RAISES = 1, 2, 4, 8, 16


class Outputs(C.Structure):
    _fields_ = [("obs_num", C.c_void_p), ("lasers", C.c_void_p), ("target", C.c_void_p),
                ("reward", C.c_void_p), ("done", C.c_void_p), ("status", C.c_void_p), ("policy_obs", C.c_void_p)]

# ftl_tune keys (include/ftl.h)
FTL_TUNE_COSCHEDULED_ENVS, FTL_TUNE_REGROUP_EVERY, FTL_TUNE_TWO_STREAMS = 0, 1, 2

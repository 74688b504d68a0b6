/*
 * ftl.h -- C-ABI of the MI355X-native batched `Game.step()` for the 2-D
 * continuous_grid_arctic follow-the-leader environment.
 *
 * The reference has no FFI layer: the path sits behind the gym API of
 * `class Game(gym.Env)` (reference src/continuous_grid_arctic/
 * follow_the_leader_continuous_env.py, "ENV" below) plus the sensor plugin
 * registry (utils/sensors.py "SEN", utils/classes.py "CLS").  The entry points
 * below are what a ctypes binding inside the reference's `Game` would call in
 * place of its Python hot loop; each one cites the reference interface it
 * replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain pointers and sizes only (no torch types); every function
 * returns 0 on success or a negative FTL_E_* code and stores a message
 * retrievable with ftl_last_error(); a handle is bound to one (process,
 * device, stream-at-call-time) and is not thread-safe -- the same contract as
 * the reference (one env object per process, Python exceptions instead of
 * codes).  All buffer arguments of ftl_reset/ftl_step are DEVICE pointers
 * owned by the caller (PyTorch-ROCm tensors); the library only borrows them
 * for the duration of the call.
 */
#ifndef FTL_H
#define FTL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTL_ABI_VERSION 4
#define FTL_MAX_BEARS 6   /* robots per env = 2 + bears <= 8 (one lane each in a group of the frame kernel); bears 5, 7, .. of
                             move_bear_v4 draw their way-points from `random` every frame (ENV:750-754): ftl_rand_range below */
#define FTL_MAX_LASERS 4
#define FTL_MAX_AUX 8     /* lidar / leader-track detectors per env (ftl_aux_cfg) */
#define FTL_MAX_REGIME 16 /* entries of leader_speed_regime / leader_acceleration_regime */
#define FTL_OBS_NUM 10    /* numerical_features, ENV:1793-1802 */
#define FTL_TRAJ_BLOCK 32 /* trajectory points per bounding-box block (state field "traj_bb"; traj_cap is a multiple) */

/* error codes */
#define FTL_OK 0
#define FTL_E_INVALID (-1)   /* bad argument / config rejected (reference: ValueError, ENV:419-427, SEN:761-762) */
#define FTL_E_UNSUPPORTED (-2) /* reference feature outside the hot-path scope (NotImplementedError) */
#define FTL_E_DEVICE (-3)    /* HIP runtime error */
#define FTL_E_STATE (-4)     /* call order: state not bound / scenarios not loaded */

/* status codes written to `status[n][3]` = info dict of ENV:951-955 */
enum { FTL_MISSION_IN_PROGRESS = 0, FTL_MISSION_FAIL = 1, FTL_MISSION_SUCCESS = 2, FTL_MISSION_FINISHED_BY_TIME = 3 };
enum { FTL_AGENT_MOVING = 0, FTL_AGENT_CRASH = 1, FTL_AGENT_LOW_REWARD = 2, FTL_AGENT_TOO_FAR = 3, FTL_AGENT_FINISHED = 4 };
enum { FTL_LEADER_MOVING = 0, FTL_LEADER_CRASH = 1, FTL_LEADER_FINISHED = 2 };

/* per-env error bits (env_int[FTL_EI_ERROR]); a reference run would have raised here */
#define FTL_ERR_TRAJ_OVERFLOW 1u      /* leader_factual_trajectory longer than traj_cap */
#define FTL_ERR_CORR_OVERFLOW 2u      /* tracker history longer than corr_cap */
#define FTL_ERR_EMPTY_CORRIDOR 4u     /* SEN:893/962: scan with len(corridor) <= 1 (reference: UnboundLocalError) */
#define FTL_ERR_TRACKER_SEED 8u       /* SEN:264-297: fewer than 2 seed points / popleft on empty corridor */
#define FTL_ERR_HIST1_OVERFLOW 16u    /* v1 tracker history longer than hist1_cap */
#define FTL_ERR_LIDAR_OVERFLOW 32u    /* more than 128 objects within range of a LaserSensor: the extra ones were ignored */
#define FTL_ERR_BAD_ACTION 64u        /* ftl_step_encoded: a Discrete(5) action outside 0..4 (reference: KeyError at ENV:922); stepped as action 2 */

/* robot kinematic limits, px/frame and deg/frame (ENV:330-357, 556-566, 704-714; CLS:59-105) */
typedef struct ftl_robot_params {
    double min_speed, max_speed;
    double max_rotation_speed;
    double max_speed_change;          /* "acceleration" */
    double max_rotation_speed_change; /* 20/100 everywhere in the reference */
    int32_t img_w, img_h;             /* size of the scaled sprite = un-rotated hitbox (CLS:42) */
    int32_t _pad[2];
} ftl_robot_params;

/* one LeaderCorridor_Prev_lasers_v2 instance (SEN:742-769, 873-881) */
typedef struct ftl_laser_cfg {
    int32_t count;            /* lasers_count */
    int32_t react_corridor;   /* react_to_safe_corridor */
    int32_t react_green;      /* react_to_green_zone */
    int32_t react_obstacles;  /* 0 False, 1 True/"all", 2 "static", 3 "dynamic" (SEN:651-660) */
    int32_t history;          /* max_prev_obs (rows of the output) */
    int32_t after_tracker;    /* 1: scanned after the tracker's 2nd scan of the step (dict order, CLS:269-286) */
    int32_t out_offset;       /* filled by the library: offset of this sensor's [history][width] block in `lasers` */
    int32_t pad_sectors;      /* SEN:932-953: rows are [front|right|behind|left], 4*count wide, zeros outside a ray's sector */
    int32_t lenient;          /* 1: LeaderCorridor_lasers_v2 (SEN:736-807) -- one row of the current edges, and a corridor of <= 1 points
                                 reads laser_length on every ray instead of raising (no FTL_ERR_EMPTY_CORRIDOR) */
    int32_t in_policy_obs;    /* 1: the sensor is one of the classes ContinuousObserveModifier_sensorPrev concatenates
                                 (LeaderCorridor_Prev_lasers_v2/_v3 and LeaderCorridor_lasers_compas, utils/wrappers.py:204, 214) */
    double length;            /* laser_length, px */
    double angle_offset;      /* first_laser_angle_offset, deg */
    int32_t explicit_angles;  /* 1: LeaderCorridor_lasers (SEN:571-702) -- ray i points at direction + ray_angles[i] instead of a full circle */
    int32_t compas;           /* 1: LeaderCorridor_lasers_compas (SEN:1138-1288) -- corridor walls only, kept in float64; rows are 5*count wide:
                                 [no wall hit | front | back | left | right] by the orientation of the nearest wall (ftl_aux_kernel) */
    double ray_angles[8];     /* deg: -40, 0, 40 [, -90, 90] [, -150, 150] (SEN:609-632) */
} ftl_laser_cfg;

/* The sensors of the registry (SEN:1291-1307) that are not ray casts against segments: their float32 outputs are further
 * blocks of ftl_outputs.lasers, after the ray sensors' blocks, in dict order. */
enum { FTL_AUX_LIDAR = 1,         /* LaserSensor (SEN:18-145): point-in-rect marching along available_angle / angle_step rays */
       FTL_AUX_TRACK_VECTOR = 2,  /* LeaderTrackDetector_vector (SEN:342-387): vectors follower -> the newest / oldest tracked leader positions */
       FTL_AUX_TRACK_RADAR = 3 }; /* LeaderTrackDetector_radar (SEN:390-487): nearest tracked position per sector of the front half plane */
typedef struct ftl_aux_cfg {
    int32_t kind;
    int32_t after_tracker;        /* 1: scanned after the v2 tracker's 2nd scan of the step (dict order, CLS:269-286) */
    int32_t out_offset, out_len;  /* filled by the library: block inside ftl_outputs.lasers, f32 elements */
    /* lidar */
    int32_t n_angles;             /* 1 + 2 * (number of angle_step increments until border_angle is reached), SEN:88-101 */
    int32_t points_number;
    int32_t return_all_points;    /* the scan returns EVERY marching point up to and including the first hit of every ray (all points_number of a ray
                                     without a hit), rays in order, as the reference's list does (SEN:112-113, 131-134) -- an array whose length K
                                     changes from call to call: out = [K as a float][K points (x, y) or K distances][zeros] in a block of
                                     1 + n_angles * points_number * (2 or 1) floats */
    int32_t return_only_distances; /* out = [n][1] norms instead of [n][2] offsets (SEN:131-134) */
    double  range_px;             /* sensor_range * PIXELS_TO_METER */
    double  in_range_px;          /* range_px + 3 * PIXELS_TO_METER: objects farther than this (distance_to_rect) are ignored, SEN:78-79 */
    double  angle_step;
    int32_t border_angle;         /* int(available_angle / 2) */
    /* detectors */
    int32_t seq_len;              /* position_sequence_length */
    int32_t detectable;           /* 0 "new", 1 "old", 2 "near" (radar only) */
    int32_t radar_sectors;
} ftl_aux_cfg;

/* Game(**kwargs) after unit conversion (ENV:45-105, 283-357) */
typedef struct ftl_config {
    int32_t abi_version;
    int32_t width, height;               /* game_width, game_height */
    int32_t frames_per_step;
    int32_t max_steps;                   /* in frames (ENV:1127-1134) */
    int32_t warm_start;                  /* in frames */
    int32_t trajectory_saving_period;    /* 5, ENV:262 */
    int32_t n_static;                    /* walls + rocks */
    int32_t n_bears;
    int32_t move_bear_v4;
    int32_t ignore_follower_collisions;
    int32_t aggregate_reward;
    int32_t has_low_reward, has_max_distance_coef; /* early_stopping keys, ENV:1088-1107 */
    int32_t has_tracker;                 /* 2: LeaderPositionsTracker_v2 present; 1: the deprecated LeaderPositionsTracker (SEN:148-229: scanned once
                                            per step, corridor half-width max_dev, never trimmed, history thinned by eat_close_points); 0: none */
    int32_t tracker_saving_period;
    int32_t tracker_start_behind;        /* start_corridor_behind_follower */
    int32_t n_lasers;
    int32_t traj_cap;                    /* capacity of leader_factual_trajectory per env (points) */
    int32_t corr_cap;                    /* capacity of tracker history / corridor ring per env */
    int32_t route_cap;                   /* capacity of the planned route per scenario (waypoints) */
    int32_t init_traj_cap;               /* capacity of the initial trajectory per scenario */
    int32_t hist1_cap;                   /* capacity of the v1 tracker's position history per env (points) */
    int32_t trk1_eat_close_points;       /* v1 tracker: eat_close_points */
    double low_reward, max_distance_coef;
    double min_distance, max_distance, max_dev; /* px */
    double leader_pos_epsilon;
    double corridor_length, corridor_width;
    /* Reward dataclass, reward_constructor.py:4-16 with leader_movement_reward=0 (ENV:279) */
    double reward_in_box, reward_on_track, reward_in_dev, not_on_track_penalty;
    double crash_penalty, too_close_penalty, leader_movement_reward;
    ftl_robot_params leader, follower, bear;
    ftl_laser_cfg lasers[FTL_MAX_LASERS];
    /* leader_speed_regime (ENV:382-386, 1143-1157): entries in dict insertion order; the LAST entry with key <=
     * step_count (frames) wins; a [lo, hi] entry draws uniform(lo, hi) EVERY frame.  n_speed_regime < 0: None. */
    int32_t n_speed_regime;
    int32_t n_acc_regime;                /* leader_acceleration_regime (ENV:390-394, 1159-1174); < 0: None */
    int32_t speed_key[FTL_MAX_REGIME];
    int32_t speed_is_range[FTL_MAX_REGIME];
    int32_t acc_key[FTL_MAX_REGIME];
    int32_t env_id_base;                 /* global index of env 0 of this handle (multi-GPU shards draw distinct streams) */
    int32_t rand_fps_lo, rand_fps_hi;    /* random_frames_per_step bounds [lo, hi) (ENV:402-405, 939-940); hi == 0: fixed frames_per_step */
    int32_t _pad1;
    double speed_lo[FTL_MAX_REGIME], speed_hi[FTL_MAX_REGIME];
    double acc_val[FTL_MAX_REGIME];
    uint64_t rng_seed;                   /* seed of the counter-based streams that replace the global `random` (ftl_uniform01) */
    double trk1_eat_radius;              /* v1 tracker: max(follower.width, follower.height) in px (SEN:213) */
    int32_t n_aux, _pad2;
    ftl_aux_cfg aux[FTL_MAX_AUX];
} ftl_config;

/* Counter-based uniform stream that stands in for `random.uniform` at ENV:1156: the draw of
 * frame `frame` of the `resets`-th episode of global env `env_id` is a pure function of (rng_seed, env_id, resets,
 * frame), so the oracle, the device and the golden generator agree without sharing generator state. */
static inline uint64_t ftl_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
static inline double ftl_uniform01(uint64_t rng_seed, uint64_t env_id, uint64_t resets, uint64_t frame) {
    uint64_t key = ftl_mix64(rng_seed + 0x9E3779B97F4A7C15ULL * (env_id + 1)) ^ ftl_mix64(0xD1B54A32D192ED03ULL * (resets + 1));
    return (double)(ftl_mix64(key + 0x9E3779B97F4A7C15ULL * (frame + 1)) >> 11) * (1.0 / 9007199254740992.0);
}

/* np.random.randint(lo, hi) of ENV:405/940 on the same counter stream, in a key range of its own (bit 40 of the frame key):
 * the draw made after the step that ended at frame `step_count` of episode `resets` (0, 0: the constructor's draw). */
static inline int32_t ftl_rand_frames(uint64_t rng_seed, uint64_t env_id, uint64_t resets, uint64_t step_count, int32_t lo, int32_t hi) {
    double u = ftl_uniform01(rng_seed, env_id, resets, step_count | (1ULL << 40));
    int32_t v = lo + (int32_t)(u * (double)(hi - lo));
    return v < hi ? v : hi - 1;
}

/* random.randrange(start, stop, 10) of ENV:753-754 (the way-points of bears with an odd index >= 5 under move_bear_v4: four (x, y)
 * pairs per bear and frame, of which the pair at dynamics_index is used) on the same counter stream, key range bit 41: draw `k`
 * (0..7 = x, y of pair 0..3) of bear `bear` in frame `frame`.  CPython: start + 10 * _randbelow(ceil((stop - start) / 10)). */
static inline int32_t ftl_rand_range(uint64_t rng_seed, uint64_t env_id, uint64_t resets, uint64_t frame, int32_t bear, int32_t k,
                                     int32_t start, int32_t stop) {
    const int32_t n = (stop - start + 9) / 10;
    double u = ftl_uniform01(rng_seed, env_id, resets, frame | (1ULL << 41) | ((uint64_t)bear << 44) | ((uint64_t)k << 48));
    int32_t v = (int32_t)(u * (double)n);
    return start + 10 * (v < n ? v : n - 1);
}

/* Scenario pool = output of the reference's reset() (ENV:434-543) for P episodes, device arrays.
 * Robots are ordered leader, follower, bear0.. (R = 2 + n_bears). */
typedef struct ftl_scenarios {
    int32_t n_scenarios;
    int32_t _pad;
    const int32_t* static_rects;   /* [P][n_static][4]  x,y,w,h  (integer pygame.Rect) */
    const float*   robot_pos;      /* [P][R][2]  f32 positions (CLS:47) */
    const double*  robot_dir;      /* [P][R]     start directions, deg */
    const int32_t* robot_rect;     /* [P][R][4] */
    const double*  route;          /* [P][route_cap][2]  planned route waypoints (ENV:1547-1550) */
    const int32_t* route_len;      /* [P] */
    const float*   init_traj;      /* [P][init_traj_cap][2]  initial leader_factual_trajectory (ENV:533-539) */
    const int32_t* init_traj_len;  /* [P] */
} ftl_scenarios;

/* ---- reset-time scenario generation (host side, no GPU involved) ------------------------------
 * The scenario part of Game.reset(): robots (ENV:545-611), bridge walls + rocks by rejection sampling (ENV:613-677),
 * finish point (ENV:1614-1630), grid route (ENV:1493-1612 on utils/dstar.py:84-210), bears (ENV:687-720, 761-770),
 * initial leader trajectory (ENV:533-539).  The draws come from a bit-compatible twin of CPython's `random`
 * (MT19937, seed(int), randrange) so that seed s yields the scenario of `game.seed(s); game.reset()`.
 * planner 1 ("astar") follows utils/astar.py:50-166 literally -- f = g + squared distance, CPython's heapq order on ties, the 1000-iteration
 * cap that returns the path to the last expanded node, the two legs through the bridge (ENV:1670-1700); found_target_point stays False as
 * in the reference (ENV:1537 is D*-only), so FTL_SCEN_FOUND is set whenever the route has at least two points.
 * The (dstar) route is a shortest 8-connected path on the reference's cost model (1 / sqrt 2 per move, inflated obstacle
 * cells); among equal-cost paths the reference's choice depends on CPython set iteration order over object ids and is
 * not reproducible -- the generator breaks such ties by insertion order (documented as unpinned in DESIGN.md). */
typedef struct ftl_scen_params {
    int32_t width, height;                /* game_width, game_height */
    int32_t step_grid, obstacle_number;   /* obstacle_number is 0 when add_obstacles is False (ENV:323-324) */
    int32_t add_obstacles, add_bear, bear_number, bear_behind;
    int32_t multiple_end_points, path_finding_iterations;
    int32_t bridge_gap, bridge_width;     /* bridge_size[0], bridge_size[1] (ENV:617-620) */
    int32_t trajectory_saving_period;
    int32_t planner;                      /* path_finding_algorythm: 0 "dstar" (ENV:1493-1612), 1 "astar" (ENV:1632-1711 on utils/astar.py);
                                             2: the caller-supplied `trajectory=` of the constructor (ENV:229, 469-470): no finish point is drawn, no
                                             planner runs, every scenario gets fixed_route */
    double  min_distance, max_distance;   /* pixels */
    double  leader_pos_epsilon, leader_margin;
    double  leader_w, leader_h;           /* the float pixel sizes the reference keeps on the robot (ENV:352-353, CLS:104-105) */
    double  leader_max_speed;             /* px/frame */
    const double* fixed_route;            /* planner 2: [fixed_route_len][2] way-points (HOST pointer), else NULL */
    int32_t fixed_route_len, _pad;
} ftl_scen_params;

/* per-scenario status bits written by ftl_generate_scenarios */
#define FTL_SCEN_FOUND        1u   /* found_target_point (ENV:1537): the route reaches the finish point */
#define FTL_SCEN_DONE_AT_RESET 2u  /* empty route (ENV:508-510) */
#define FTL_SCEN_ROUTE_OVERFLOW 4u /* route longer than cfg->route_cap: truncated, do not use */
#define FTL_SCEN_TRAJ_OVERFLOW 8u  /* initial trajectory longer than cfg->init_traj_cap: truncated, do not use */
#define FTL_SCEN_REF_RAISES  16u   /* the reference's reset() would raise here (one-point route, ENV:513) */

/* Fill `out` (HOST arrays shaped like ftl_scenarios with P = n; cfg gives n_static, R, route_cap, init_traj_cap) with the
 * scenarios of python seeds seeds[0..n); status[i] gets the FTL_SCEN_* bits of scenario i.  n_threads <= 0: all cores. */
int ftl_generate_scenarios(const ftl_config* cfg, const ftl_scen_params* sp, const int64_t* seeds, int32_t n,
                           int32_t n_threads, const ftl_scenarios* out, uint8_t* status);

/* struct sizes as this library was compiled (binding self-check) */
size_t ftl_sizeof_config(void);
size_t ftl_sizeof_scenarios(void);
size_t ftl_sizeof_outputs(void);
size_t ftl_sizeof_scen_params(void);


/* step()/reset() outputs = (obs, reward, done, info) of ENV:945 for n envs, device arrays */
typedef struct ftl_outputs {
    float*   obs_num;    /* [n][10]            numerical_features (ENV:1793-1802) */
    float*   lasers;     /* [n][lasers_len]    per ray sensor k a [history_k][width_k] block at lasers[k].out_offset, width_k = count_k
                            (4*count_k with pad_sectors, 5*count_k for compas); then per aux sensor a block of aux[j].out_len at
                            aux[j].out_offset */
    double*  target;     /* [n][2]             leader_target_point (ENV:1803-1806) */
    double*  reward;     /* [n]                last-frame reward (ENV:935-936, 1136-1141) */
    uint8_t* done;       /* [n] */
    uint8_t* status;     /* [n][3]             mission / agent / leader status codes */
    float*   policy_obs; /* optional (may be NULL): [n][H][sum_k width_k] = ContinuousObserveModifier_sensorPrev.observation
                            (utils/wrappers.py:200-221): per sensor with in_policy_obs set clip(x / laser_length, 0, 1),
                            concatenated along axis 1; those sensors must share one history H (wrappers.py:183-188) */
} ftl_outputs;

typedef struct ftl_handle ftl_handle;

/* flags of ftl_step */
#define FTL_STEP_AUTO_RESET 1u  /* envs that finish are re-initialised from scenario (scen_idx+n_envs) % P inside the
                                   same launch; outputs keep the terminal reward/done/status, obs are the new episode's */

/* Game.__init__ (ENV:45-417): validate + freeze the config. device < 0 is rejected (there is no CPU path). */
int ftl_create(const ftl_config* cfg, int32_t n_envs, int32_t device, ftl_handle** out);
void ftl_destroy(ftl_handle* h);

/* number of f32 elements per env in ftl_outputs.lasers */
int32_t ftl_lasers_len(const ftl_handle* h);
/* copy of the frozen config (out_offset of every laser filled in) */
int ftl_get_config(const ftl_handle* h, ftl_config* out);

/* Per-env mutable state lives in ONE caller-owned device buffer (a torch uint8 tensor).  A fresh buffer MUST be
 * zero-initialised: the reset path relies on zeroed FTL_EI_FPS / FTL_EI_RESETS / FTL_EI_ACC_CONSUMED / FTL_EI_EPISODES /
 * FTL_EI_ERROR_STICKY words and on a zeroed "ep_stats" field (they survive reset() like the attributes of the reference's
 * Game object that reset() does not touch).  A buffer that already holds the state of an earlier run may be bound again. */
size_t ftl_state_bytes(const ftl_handle* h);
int ftl_bind_state(ftl_handle* h, void* dev_state, size_t bytes);

/* State introspection for parity tests: byte offset / element count / dtype code / row stride of a named field
 * ("rb_pos","rb_dbl","rb_int","env_int","env_dbl","fol_cs","snap_rects","snap_win" -- the fields of the per-env record -- and
 * "traj","traj_bb","hist","corr","corr32","ep_stats","hist1").  dtype: 0 i32, 1 f32, 2 f64.  Element j of env e sits at byte
 * offset + e * stride + j * sizeof(dtype), j < per_env: the record fields share one stride (the record size, a multiple of 128), the
 * others are dense [n_envs][per_env] arrays. */
int ftl_state_field(const ftl_handle* h, const char* name, size_t* offset, size_t* per_env, int32_t* dtype, size_t* stride);

/* reset() part 1 (ENV:461-492): hand over the scenario pool produced by reset-time generation. */
int ftl_load_scenarios(ftl_handle* h, const ftl_scenarios* pool);

/* The pool entries the in-kernel auto-reset (FTL_STEP_AUTO_RESET) draws from: a finished env that ran scenario s restarts from
 * base + ((s mod count) + stride) mod count.  ftl_load_scenarios sets the window to the whole pool with stride n_envs (base 0, count
 * n_scenarios), which is the walk documented at FTL_STEP_AUTO_RESET.  stride <= 0 keeps n_envs; a stride that shares a factor with count
 * visits only part of the window (n_envs a multiple of count: the same scenario again and again, which turns the few worlds that start
 * the follower inside a rock into one-step episodes forever) -- pick one that is coprime to count.  A caller that refills one half of a double-sized pool while the envs draw from the other
 * half (the reference builds a fresh world on every reset(), ENV:461-492; scenario.ScenarioRing) moves the window between steps; entries
 * outside the window stay valid for the episodes that are still running on them, so a half may be overwritten once every episode that
 * started before the window left it has ended (at most max_steps / frames_per_step + 1 steps). */
int ftl_set_reset_window(ftl_handle* h, int32_t base, int32_t count, int32_t stride);

/* Scheduling hints of a handle.  Results never depend on them (tests/test_gpu_api.py: regrouping, split path, pipelined parts).
 * FTL_TUNE_COSCHEDULED_ENVS: the number of envs that are stepped on this device at the same time, this handle's included -- a caller that
 *   runs the batch as several handles on several streams (two handles of 32,768 envs each: DESIGN.md section 6, PipelinedVecGame) says
 *   65,536 here.  The handle sorts its envs by expected cost when THAT many envs need more than one round of frame-kernel wavefronts
 *   (by itself it only knows its own batch, which it leaves unsorted when one round holds it).
 * FTL_TUNE_REGROUP_EVERY: rebuild the cost order every k-th step (default 4).
 * FTL_TUNE_TWO_STREAMS: 0 / 1 -- the handle's own two-stream mode (its two halves on two streams, joined every step: the default for
 *   configs with random_frames_per_step); a caller that overlaps whole handles switches it off.
 * The environment switches FTL_NO_REGROUP / FTL_REGROUP_EVERY / FTL_SPLIT (diagnostics) win over the hints. */
enum { FTL_TUNE_COSCHEDULED_ENVS = 0, FTL_TUNE_REGROUP_EVERY = 1, FTL_TUNE_TWO_STREAMS = 2 };
int ftl_tune(ftl_handle* h, int32_t what, int32_t value);

/* reset() (ENV:494-543): place env e at scenario scen_idx[e] for every e with mask[e] != 0 (mask NULL = all),
 * run the initial use_sensors (ENV:541) and write the first observation.  reward/done/status are zeroed. */
int ftl_reset(ftl_handle* h, const int32_t* scen_idx, const uint8_t* mask, const ftl_outputs* out, void* stream);

/* step(action) (ENV:908-945) for all envs: action[n][2] = (speed px/frame, signed rotation deg/frame) as f64. */
int ftl_step(ftl_handle* h, const double* action, const ftl_outputs* out, uint32_t flags, void* stream);

/* step(action) for the two other action spaces of the constructor (ENV:358-378), decoded on the device exactly as ENV:909-925 does:
 *   FTL_ACTION_BOX2      action = f64 [n][2]  (speed, signed rotation)                                   -- the same as ftl_step
 *   FTL_ACTION_DISCRETE  action = int32 [n]   index k of Discrete(5) -> (follower.max_speed, discrete_rotation_speed_to_value[k]) with the
 *                                 table {-max_rot, -max_rot/2, 0, max_rot/2, max_rot} of ENV:362-367 (discrete_action_space=True)
 *   FTL_ACTION_TURN      action = f64 [n]     Box(1) rotation -> (0.25, rotation): np.concatenate([[0.25], action]) of ENV:924-925
 *                                 (constant_follower_speed=True; the speed command of ENV:910-911 is overwritten by ENV:927) */
enum { FTL_ACTION_BOX2 = 0, FTL_ACTION_DISCRETE = 1, FTL_ACTION_TURN = 2 };
int ftl_step_encoded(ftl_handle* h, const void* action, int32_t encoding, const ftl_outputs* out, uint32_t flags, void* stream);

/* ---- episode metrics + error report (ENV:941-944 reports overall_reward / step_count at done) --------
 * Every env slot accumulates, at the step in which an episode ends (done set by this step; under FTL_STEP_AUTO_RESET
 * before the slot is re-initialised), the vector below in its "ep_stats" state field (f64[FTL_N_METRICS] per env).
 * ftl_episode_metrics sums those records over the envs of the handle in a fixed order (bit-reproducible) into
 * dev_metrics[FTL_N_METRICS] (DEVICE pointer) -- the 64-byte vector a multi-GPU job all-reduces -- and reports the
 * sticky error words: dev_errors[0] = number of envs whose FTL_EI_ERROR_STICKY is non-zero, dev_errors[1] = OR of them
 * (DEVICE pointer, may be NULL).  FTL_METRICS_CLEAR zeroes the per-env records and sticky words afterwards. */
#define FTL_N_METRICS 8
enum { FTL_M_EPISODES = 0,   /* finished episodes */
       FTL_M_RETURN_SUM,     /* sum of overall_reward at done (ENV:943) */
       FTL_M_FRAMES_SUM,     /* sum of step_count at done, in frames (ENV:944) */
       FTL_M_SUCCESS,        /* mission_status == success at done */
       FTL_M_CRASH,          /* agent_status == crash */
       FTL_M_LOW_REWARD,     /* agent_status == low_reward */
       FTL_M_TOO_FAR,        /* agent_status == too_far_from_leader */
       FTL_M_TIMEOUT };      /* mission_status == finished_by_time */
#define FTL_METRICS_CLEAR 1u
int ftl_episode_metrics(ftl_handle* h, double* dev_metrics, int32_t* dev_errors, uint32_t flags, void* stream);

/* ---- measurement hook (bench.py): per-kernel durations from HIP events on the launch stream ------------------------------
 * While enabled every ftl_step records events around its launches (frame kernel, ray kernel, ftl_aux_kernel, the two regroup kernels);
 * ftl_kernel_times synchronises and returns the SUM of the durations in milliseconds since the last call as
 * ms[4] = {frames (+ the v1 tracker's kernel), rays, aux (row-f3 sensors; ~0 without them), regroup} and the number of steps they cover.  At most 512 steps are held; not available in the
 * two-stream mode (returns FTL_E_UNSUPPORTED).  Off by default: the events cost a few microseconds per step. */
int ftl_kernel_timing(ftl_handle* h, int32_t enable);
int ftl_kernel_times(ftl_handle* h, double* ms, int32_t* n_steps);

const char* ftl_last_error(void);

/* indices into the "env_int" state field */
enum {
    FTL_EI_SCEN = 0, FTL_EI_TARGET_ID, FTL_EI_LEADER_FINISHED, FTL_EI_DONE, FTL_EI_CRASH, FTL_EI_IN_BOX,
    FTL_EI_ON_TRACE, FTL_EI_TOO_CLOSE, FTL_EI_STEP_COUNT, FTL_EI_FINISH_TIMER, FTL_EI_TRAJ_LEN,
    FTL_EI_TRK_COUNTER, FTL_EI_CORR_LO, FTL_EI_CORR_HI, FTL_EI_SEED_END, FTL_EI_SNAP_COUNT,
    FTL_EI_DYN_INDEX0, FTL_EI_DYN_INDEX1, FTL_EI_DYN_INDEX2, FTL_EI_DYN_INDEX3, FTL_EI_DYN_INDEX4, FTL_EI_DYN_INDEX5,
    FTL_EI_ERROR, FTL_EI_EPISODES, FTL_EI_GREEN_COUNT, FTL_EI_GREEN_LEN,
    FTL_EI_SCAN_OK,   /* bit g set: the ray sensors of dict-order group g (before / after the tracker's 2nd scan) scanned this step */
    FTL_EI_SNAP_HEAD, /* ring slot the next snapshot goes to (= snap_count mod max_prev_obs, kept incrementally) */
    FTL_EI_HINT,      /* index of a trajectory point that was close to the follower last frame (search hint only) */
    FTL_EI_GREEN_TINY, /* the green window may hold a segment so short that f64 sums of segment lengths are no longer exact */
    FTL_EI_RESETS,     /* number of resets of this env slot so far (keys the RNG stream of the episode) */
    FTL_EI_ACC_CONSUMED, /* bit i: entry i of leader_acceleration_regime was consumed -- the reference deletes the key for good (ENV:1170) */
    /* search caches of _check_agent_position (float bit patterns; they never change a result, only which points are looked at):
       coordinates of trajectory point FTL_EI_HINT, and lower bounds on the follower's distance to every green point /
       to every trajectory point */
    FTL_EI_HINT_X, FTL_EI_HINT_Y, FTL_EI_CLR_GREEN, FTL_EI_CLR_ALL,
    FTL_EI_FPS,        /* frames of the NEXT step of this env under random_frames_per_step (drawn at the end of a step, ENV:939-940;
                          kept across resets like the reference's attribute; 0 = not drawn yet) */
    FTL_EI_HW0_LO, FTL_EI_HW0_HI, /* tracker history window after the FIRST tracker scan of the step (what a detector that precedes the tracker
                          in dict order sees); after the second one it is FTL_EI_CORR_LO / FTL_EI_CORR_HI */
    FTL_EI_HIST1_LEN,   /* v1 tracker: points in the "hist1" field */
    FTL_EI_ERROR_STICKY, /* OR of every FTL_ERR_* bit this env slot ever raised: survives reset / auto-reset (FTL_EI_ERROR is per
                          episode); cleared by ftl_episode_metrics(FTL_METRICS_CLEAR) */
    FTL_EI_COUNT
};
/* indices into the "env_dbl" state field; bear waypoints follow at FTL_ED_BEAR_POINTS + 2*b */
enum { FTL_ED_ACC_PENALTY = 0, FTL_ED_OVERALL_REWARD, FTL_ED_SPARE0, FTL_ED_SPARE1, FTL_ED_BEAR_POINTS,
       FTL_ED_GREEN_W = FTL_ED_BEAR_POINTS + 2 * FTL_MAX_BEARS, /* running length of the green-zone window (search acceleration) */
       FTL_ED_CUR_MULT,  /* cur_speed_multiplier (ENV:412, 449, 1150-1156) */
       FTL_ED_CUR_ACC, FTL_ED_CUM_SPEED, /* cur_leader_acceleration, cur_leader_cumulative_speed (ENV:591-592, 1167-1172) */
       FTL_ED_COUNT };
/* per robot: rb_dbl[5] and rb_int[8] */
enum { FTL_RD_DIRECTION = 0, FTL_RD_SPEED, FTL_RD_ROT_SPEED, FTL_RD_DES_SPEED, FTL_RD_DES_ROT_SPEED, FTL_RD_COUNT };
enum { FTL_RI_X = 0, FTL_RI_Y, FTL_RI_W, FTL_RI_H, FTL_RI_ROT_DIR, FTL_RI_DES_ROT_DIR, FTL_RI_SPARE0, FTL_RI_SPARE1, FTL_RI_COUNT };

#ifdef __cplusplus
}
#endif
#endif /* FTL_H */

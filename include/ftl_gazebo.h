/*
 * ftl_gazebo.h -- C-ABI of the follower-relative ("Gazebo") variant of the tracker / ray-sensor maths:
 * reference src/arctic_gym/gazebo_utils/gazebo_tracker.py ("GZ" below), classes GazeboLeaderPositionsTracker_v2 (GZ:13-172) and
 * GazeboCorridor_Prev_lasers_v2 (GZ:175-297), driven by src/arctic_gym/arctic_env/arctic_env.py:62-90, 190-211.
 *
 * Everything lives in the follower's frame: the follower sits at (0, 0); every call shifts the stored leader history and corridor by
 * the follower's displacement `delta` since the last call (GZ:46-79), the obstacle edges come from lidar-derived point pairs
 * instead of pygame rectangles (GZ:190-198).  The constants the reference hard-codes inside scan() (GZ:34-43: saving_period 3,
 * corridor half-width 2, corridor length 25, 10 seed points starting 10 behind the follower, ray offset -45 deg at GZ:212) are
 * hard-coded here too.
 *
 * Same conventions as ftl.h: plain pointers and sizes, 0 / negative FTL_E_* codes + ftl_last_error(), DEVICE pointers for the
 * per-call arrays, one handle per (process, device), state in ONE caller-owned zero-initialised device buffer.
 */
#ifndef FTL_GAZEBO_H
#define FTL_GAZEBO_H

#include "ftl.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FTL_GZ_MAX_LASERS 2    /* arctic_env.py keeps two: `laser` (12 rays, all edges) and `laser_aux` (36 rays, obstacles only) */
#define FTL_GZ_HIST_CAP 64     /* leader history / corridor points per env: the path is trimmed to length 25 with points >= 1 apart */

/* one GazeboCorridor_Prev_lasers_v2 (constructor = LeaderCorridor_Prev_lasers_v2's, SEN:742-769, 873-881) */
typedef struct ftl_gz_laser_cfg {
    int32_t count, history;                 /* lasers_count, max_prev_obs */
    int32_t react_corridor, react_green, react_obstacles, pad_sectors;
    double  length;                         /* laser_length */
} ftl_gz_laser_cfg;

typedef struct ftl_gz_config {
    int32_t n_lasers;
    int32_t max_pts;                        /* capacity of cur_object_points_1 / _2 per env and call */
    ftl_gz_laser_cfg lasers[FTL_GZ_MAX_LASERS];
} ftl_gz_config;

typedef struct ftl_gz_handle ftl_gz_handle;

int ftl_gz_create(const ftl_gz_config* cfg, int32_t n_envs, int32_t device, ftl_gz_handle** out);
void ftl_gz_destroy(ftl_gz_handle* h);
size_t ftl_gz_state_bytes(const ftl_gz_handle* h);
int ftl_gz_bind_state(ftl_gz_handle* h, void* dev_state, size_t bytes);
/* elements of f32 output per env: sum over sensors of history * (count, or 4 * count with pad_sectors) */
int32_t ftl_gz_lasers_len(const ftl_gz_handle* h);
/* tracker.reset() + every laser's reset() (SEN:223-226, 964-968) for the envs with mask[e] != 0 (mask NULL = all) */
int ftl_gz_reset(ftl_gz_handle* h, const uint8_t* mask, void* stream);
/* One tracker.scan (GZ:17-172) followed by every laser's scan (GZ:203-297), as arctic_env.py:190-211 calls them.
 *   leader_pos [n][2] f64  leader position in the follower's frame;  yaw [n] f64  follower_orientation[2], radians;
 *   delta [n][2] f64  (delta_x, delta_y);  pts1, pts2 [n][max_pts][2] f64 + n_pts [n]  cur_object_points_1 / _2;
 *   lasers [n][ftl_gz_lasers_len] f32 out: per sensor a [history][width] block, oldest row first. */
int ftl_gz_step(ftl_gz_handle* h, const double* leader_pos, const double* yaw, const double* delta, const double* pts1,
                const double* pts2, const int32_t* n_pts, float* lasers, void* stream);
/* parity introspection: byte offset of the per-env records inside the state buffer.
 * "gz_int" i32[8] = {saving_counter, hist_len, corr_len, error, ...}; "gz_hist" f64[FTL_GZ_HIST_CAP][2]; "gz_corr" f64[FTL_GZ_HIST_CAP][4] */
int ftl_gz_state_field(const ftl_gz_handle* h, const char* name, size_t* offset, size_t* per_env, int32_t* dtype);

#ifdef __cplusplus
}
#endif
#endif /* FTL_GAZEBO_H */

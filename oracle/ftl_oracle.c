/*
 * ftl_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C, one-env-at-a-time restatement of the reference's Game.step() hot path, written to
 * be read side by side with the reference (every function cites the reference file:line it
 * follows; ENV = src/continuous_grid_arctic/follow_the_leader_continuous_env.py, CLS =
 * utils/classes.py, SEN = utils/sensors.py, MISC = utils/misc.py, RWD = utils/reward_constructor.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this file,
 * and only as the checker / the reported CPU baseline.  The product (the HIP library behind
 * include/ftl.h) never routes through it.
 *
 * Pinning: checked against the golden episodes under tests/golden/ that were produced by running
 * the UNMODIFIED reference here (tests/golden/gen/make_golden.py).  The reference's own arithmetic
 * (numpy 2.2.6 / scipy 1.15.3 / glibc) is pinned by those vectors; the pygame.Rect /
 * transform.rotate / clock semantics (restated in tests/golden/gen/standins/pygame, SURVEY.md
 * Appendix B) are third-party code absent from this image => "parity unpinned" at that boundary.
 *
 * Numeric notes (established empirically against the libraries in this image):
 *   - scipy distance.euclidean(f32, f32)  = (float) sqrt((double)dx*dx + (double)dy*dy), dx,dy f32
 *     differences (BLAS snrm2);  with any f64/int operand = (double) sqrtl((long double)dx*dx +
 *     (long double)dy*dy) (BLAS dnrm2, x87).
 *   - np.sum = numpy pairwise summation over ALL n elements (8 accumulators, blocks of 128).
 *   - np.linalg.norm(1-D f64) = sqrt(fma(v1, v1, v0*v0)); (1-D f32) = sqrtf(v0*v0 + v1*v1) in f32.
 *   - np.sin / np.cos / np.radians on f64 == glibc sin / cos, x*(pi/180).
 * Build with -ffp-contract=off (no implicit FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ftl.h"

#define PI_D 3.141592653589793
static const double DEG2RAD = PI_D / 180.0; /* Python math.radians / np.radians */
static const double RAD2DEG = 180.0 / PI_D; /* Python math.degrees */

typedef struct {
    float px, py;                 /* GameObject.position, np.float32[2] (CLS:47) */
    double direction, speed, rot_speed, des_speed, des_rot_speed;
    int rot_dir, des_rot_dir;
    int rx, ry, rw, rh;           /* pygame.Rect */
    const ftl_robot_params* p;
} robot_t;

typedef struct { float a[2], b[2]; } seg_t; /* one obstacle line, f32[2][2] (SEN:672) */

typedef struct {
    seg_t* segs;
    int n;
    int valid; /* 0 = the initial zeros item of SEN:964-968 */
} snapshot_t;

typedef struct ftlo_env {
    ftl_config cfg;
    int R;
    robot_t rb[2 + FTL_MAX_BEARS]; /* 0 leader, 1 follower, 2.. bears */
    int32_t* srect;                /* [n_static][4] */
    double* route; int route_len;
    int cur_target_id; double cur_target[2];
    int leader_finished, done, crash, is_in_box, is_on_trace, too_close;
    int step_count, finish_timer /* -1 == None */;
    double accumulated_penalty, overall_reward;
    float* traj; int traj_len, traj_cap;
    int green_count;
    double bear_pt[FTL_MAX_BEARS][2]; int dyn_index[FTL_MAX_BEARS];
    /* LeaderPositionsTracker_v2 state (SEN:156-176) */
    double* hist; uint8_t* hist_f64; int hist_len, hist_cap; /* deque of points; dtype flag per point */
    double* corr; int corr_len;                               /* deque of [right.xy, left.xy] */
    int trk_counter;
    /* LeaderCorridor_Prev_lasers_v2 history per sensor (SEN:964-968) */
    snapshot_t* snaps[FTL_MAX_LASERS];
    uint32_t error;
    int lasers_len;
    /* leader regimes (ENV:412, 449, 591-592, 1143-1174) */
    double cur_mult, cur_acc, cum_speed;
    uint32_t acc_consumed;      /* keys the reference deleted from leader_acceleration_regime (never restored) */
    int fps;                    /* self.frames_per_step under random_frames_per_step (ENV:405, 939-940); 0 = constructor draw pending */
    uint64_t resets;            /* resets of this Game object so far: keys the uniform stream of the episode */
} ftlo_env;

/* ------------------------------------------------------------------ helpers */

static int laser_width(const ftl_laser_cfg* L) { return L->compas ? 5 * L->count : (L->pad_sectors ? 4 * L->count : L->count); } /* SEN:932-958, 1226 */

/* scipy.spatial.distance.euclidean on two float32 points */
static double euclid_f32(float ax, float ay, float bx, float by) {
    float dx = ax - bx, dy = ay - by;
    return (double)(float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}
/* ... when at least one operand is float64 / int (dnrm2, x87 extended accumulation) */
static double euclid_f64(double ax, double ay, double bx, double by) {
    long double dx = (long double)(ax - bx), dy = (long double)(ay - by);
    return (double)sqrtl(dx * dx + dy * dy);
}

/* MISC:6-13 */
static double angle_correction(double a) {
    if (a >= 360.0) return a - 360.0;
    if (a < 0.0) return 360.0 + a;
    return a;
}
/* MISC:16-26, relative position already in f64 */
static double angle_to_point(double cx, double cy, double tx, double ty) {
    double rx = tx - cx, ry = ty - cy, res;
    if (rx > 0.0) res = atan(ry / rx) * RAD2DEG;
    else if (rx < 0.0) res = atan(ry / rx) * RAD2DEG + 180.0;
    else res = 0.0;
    return angle_correction(res);
}

/* numpy pairwise summation (numpy/_core/src/umath/loops_utils.h.src), f64 and f32 flavours */
static double pairwise_f64(const double* a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8]; int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return pairwise_f64(a, n2) + pairwise_f64(a + n2, n - n2);
}
static float pairwise_f32(const float* a, int n) {
    if (n < 8) { float r = 0.0f; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        float r[8]; int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return pairwise_f32(a, n2) + pairwise_f32(a + n2, n - n2);
}

/* pygame.transform.rotate bounding box, transform.c surf_rotate (parity unpinned) */
static void rotate_size(int w, int h, double angle_deg, int* nw, int* nh) {
    double a = (double)(float)angle_deg; /* parsed as C float */
    if (fmod(a, 90.0) == 0.0) {
        long q = (long)(a / 90.0) % 4; if (q < 0) q += 4;
        if (q % 2) { *nw = h; *nh = w; } else { *nw = w; *nh = h; }
        return;
    }
    double r = a * .01745329251994329, s = sin(r), c = cos(r);
    double cx = c * w, cy = c * h, sx = s * w, sy = s * h;
    double mx = fmax(fmax(fmax(fabs(cx + sy), fabs(cx - sy)), fabs(-cx + sy)), fabs(-cx - sy));
    double my = fmax(fmax(fmax(fabs(sx + cy), fabs(sx - cy)), fabs(-sx + cy)), fabs(-sx - cy));
    *nw = (int)mx; *nh = (int)my;
}

/* ------------------------------------------------------------------ robots (CLS:109-215) */

static void command_turn(robot_t* r, double des, int dir) { /* CLS:109-117 */
    r->des_rot_speed = (des <= r->p->max_rotation_speed) ? des : r->p->max_rotation_speed;
    r->des_rot_dir = dir;
}
static void command_forward(robot_t* r, double s) { /* CLS:119-127 */
    if (s > r->p->max_speed) s = r->p->max_speed;
    if (s < r->p->min_speed) s = r->p->min_speed;
    r->des_speed = s;
}
static void turn_processing(robot_t* r) { /* CLS:134-153 */
    double change;
    if (r->rot_dir == 0) r->rot_dir = r->des_rot_dir;
    if (r->rot_dir == r->des_rot_dir) {
        double needed = fabs(r->rot_speed - r->des_rot_speed);
        change = (needed <= r->p->max_rotation_speed_change) ? needed : r->p->max_rotation_speed_change;
        if (r->des_rot_speed < r->rot_speed) change = -1 * change;
    } else {
        double needed = fabs(r->des_rot_speed + r->rot_speed);
        change = -((needed <= r->p->max_rotation_speed_change) ? needed : r->p->max_rotation_speed_change);
    }
    double nr = r->rot_speed + change;
    if (nr < 0) r->rot_dir = -1 * r->rot_dir;
    r->rot_speed = fabs(nr);
}
static void speed_processing(robot_t* r) { /* CLS:155-163 */
    double needed = fabs(r->speed - r->des_speed);
    double change = (r->p->max_speed_change <= needed) ? r->p->max_speed_change : needed;
    if (r->speed > r->des_speed) change = -1 * change;
    r->speed = r->speed + change;
}
static void robot_move(robot_t* r) { /* CLS:165-182 */
    turn_processing(r);
    speed_processing(r);
    if (r->rot_speed != 0) {
        r->direction = angle_correction(r->direction + r->rot_dir * r->rot_speed);
        int nw, nh, cx = r->rx + (r->rw >> 1), cy = r->ry + (r->rh >> 1);
        rotate_size(r->p->img_w, r->p->img_h, -r->direction, &nw, &nh);
        r->rw = nw; r->rh = nh;
        r->rx = cx - (nw >> 1); r->ry = cy - (nh >> 1);
    }
    double th = r->direction * DEG2RAD;
    float mx = (float)(cos(th) * r->speed), my = (float)(sin(th) * r->speed);
    r->px += mx; r->py += my;
    double dx = (double)r->px - (double)(r->rx + (r->rw >> 1));
    double dy = (double)r->py - (double)(r->ry + (r->rh >> 1));
    r->rx += (int)dx; r->ry += (int)dy; /* Rect.move_ip truncates toward zero */
}
/* CLS:184-215; speed < 0 encodes speed=None (use the distance) */
static void move_to_the_point(robot_t* r, double tx, double ty, int has_speed, double speed) {
    double new_speed = has_speed ? speed : euclid_f64((double)r->px, (double)r->py, tx, ty);
    int desirable = (int)angle_to_point((double)r->px, (double)r->py, tx, ty);
    int cur = (int)r->direction;
    int delta, dir;
    if (desirable - cur > 0) {
        if (desirable - cur > 180) { delta = cur + (360 - desirable); dir = -1; }
        else { delta = desirable - cur; dir = 1; }
    } else {
        if (cur - desirable > 180) { dir = 1; delta = (360 - cur) + desirable; }
        else { dir = -1; delta = cur - desirable; }
    }
    command_turn(r, (double)delta, dir);
    command_forward(r, new_speed);
    robot_move(r);
}

static int rects_collide(int ax, int ay, int aw, int ah, int bx, int by, int bw, int bh) { /* pygame colliderect */
    if (aw == 0 || ah == 0 || bw == 0 || bh == 0) return 0;
    return ax < bx + bw && ay < by + bh && ax + aw > bx && ay + ah > by;
}

/* ENV:1176-1194 */
static int collision_check(const ftlo_env* e, int who) {
    const robot_t* t = &e->rb[who];
    int hit = 0;
    for (int k = 0; k < 2 && !hit; k++) { /* game_object_list starts with leader, follower */
        if (k == who) continue;
        const robot_t* o = &e->rb[k];
        hit = rects_collide(t->rx, t->ry, t->rw, t->rh, o->rx, o->ry, o->rw, o->rh);
    }
    for (int s = 0; s < e->cfg.n_static && !hit; s++) {
        const int32_t* q = e->srect + 4 * s;
        hit = rects_collide(t->rx, t->ry, t->rw, t->rh, q[0], q[1], q[2], q[3]);
    }
    if (who != 0) /* the leader does not collide with dynamic obstacles (ENV:1180-1186) */
        for (int b = 0; b < e->cfg.n_bears && !hit; b++) {
            const robot_t* o = &e->rb[2 + b];
            hit = rects_collide(t->rx, t->ry, t->rw, t->rh, o->rx, o->ry, o->rw, o->rh);
        }
    if (hit) return 1;
    if ((double)t->px > (double)e->cfg.width || (double)t->py > (double)e->cfg.height) return 1;
    if (t->px < 0.0f || t->py < 0.0f) return 1;
    return 0;
}

/* ENV:1828-1843: returns G = number of green points; they are traj[n-2], traj[n-3], ..., traj[n-1-G] */
static int trajectory_in_box(const ftlo_env* e) {
    double acc = 0.0; int g = 0;
    for (int k = e->traj_len - 2; k >= 0; k--) {
        const float* cur = e->traj + 2 * k; const float* prev = e->traj + 2 * (k + 1);
        acc += euclid_f32(prev[0], prev[1], cur[0], cur[1]);
        if (acc <= e->cfg.max_distance) g++; else break;
    }
    return g;
}

/* ENV:1951-1960 on float32 points: first index of the minimal f32 squared distance */
static int closest_point(float px, float py, const float* pts, int n, int reversed) {
    int best = 0; float bestv = 0.0f;
    for (int i = 0; i < n; i++) {
        const float* q = reversed ? pts - 2 * i : pts + 2 * i;
        float dx = q[0] - px, dy = q[1] - py;
        float d2 = dx * dx + dy * dy;
        if (i == 0 || d2 < bestv) { bestv = d2; best = i; }
    }
    return best;
}

/* ENV:1906-1937 */
static void check_agent_position(ftlo_env* e, int G) {
    const robot_t* f = &e->rb[1]; const robot_t* l = &e->rb[0];
    if (G > 2) {
        const float* first = e->traj + 2 * (e->traj_len - 2); /* green[0] */
        int id = closest_point(f->px, f->py, first, G, 1);
        const float* c = first - 2 * id;
        double d = euclid_f32(f->px, f->py, c[0], c[1]);
        if (d <= e->cfg.leader_pos_epsilon) { e->is_on_trace = 1; e->is_in_box = 1; }
        else if (d <= e->cfg.max_dev) { e->is_in_box = 1; e->is_on_trace = 0; }
        else {
            int id2 = closest_point(f->px, f->py, e->traj, e->traj_len, 0);
            const float* c2 = e->traj + 2 * id2;
            if (euclid_f32(f->px, f->py, c2[0], c2[1]) <= e->cfg.leader_pos_epsilon) { e->is_on_trace = 1; e->is_in_box = 0; }
        }
    }
    e->too_close = euclid_f32(l->px, l->py, f->px, f->py) <= e->cfg.min_distance;
}

/* MISC:47-53 applied to the integer vector [k, 0]: (cos(theta)*k, sin(theta)*k) */
static void rot_k0(double k, double angle_deg, double* ox, double* oy) {
    double th = angle_deg * DEG2RAD;
    *ox = cos(th) * k; *oy = sin(th) * k;
}

/* ENV:819-837 */
static void choose_points_for_bear_stat(ftlo_env* e, int idx, double* ox, double* oy) {
    const robot_t* b = &e->rb[2 + idx]; const robot_t* l = &e->rb[0];
    if (euclid_f64((double)b->px, (double)b->py, e->bear_pt[idx][0], e->bear_pt[idx][1]) < e->cfg.leader_pos_epsilon) {
        e->dyn_index[idx] += 1;
        if (e->dyn_index[idx] > 1) e->dyn_index[idx] = 0;
    }
    double k = 100.0 * (idx + 1), vx, vy;
    rot_k0(k, l->direction + (e->dyn_index[idx] == 0 ? -130.0 : 130.0), &vx, &vy);
    *ox = (double)l->px + vx; *oy = (double)l->py + vy;
}
/* ENV:722-758 (index <= 3 only; larger indices draw from `random`) */
static void move_bear_v4(ftlo_env* e, int idx, double* ox, double* oy) {
    const robot_t* b = &e->rb[2 + idx]; const robot_t* l = &e->rb[0];
    if (euclid_f64((double)b->px, (double)b->py, e->bear_pt[idx][0], e->bear_pt[idx][1]) < e->cfg.leader_pos_epsilon)
        e->dyn_index[idx] += 1;
    if (e->dyn_index[idx] > 3) e->dyn_index[idx] = 0;
    if (idx >= 4) {   /* ENV:750-754: four random points per frame, the one at dynamics_index is used; random.randrange -> ftl_rand_range */
        const ftl_config* c = &e->cfg;
        const int lo = (int)c->max_distance, k = 2 * e->dyn_index[idx];
        *ox = (double)ftl_rand_range(c->rng_seed, (uint64_t)c->env_id_base, e->resets, (uint64_t)e->step_count, idx, k, lo, c->width - lo);
        *oy = (double)ftl_rand_range(c->rng_seed, (uint64_t)c->env_id_base, e->resets, (uint64_t)e->step_count, idx, k + 1, lo, c->height - lo);
        return;
    }
    /* p1=(150,+140) p2=(150,-140) p3=(250,-160) p4=(250,+160) */
    static const int order[4][4] = { {1, 2, 4, 3}, {4, 3, 1, 2}, {2, 4, 3, 1}, {3, 1, 2, 4} };
    int p = order[idx][e->dyn_index[idx]];
    double lvl = (p <= 2) ? 150.0 : 250.0;
    double off = (p == 1) ? 140.0 : (p == 2) ? -140.0 : (p == 3) ? -160.0 : 160.0;
    double vx, vy;
    rot_k0(lvl, l->direction + off, &vx, &vy);
    *ox = (double)l->px + vx; *oy = (double)l->py + vy;
}

/* ENV:1869-1904 */
static double reward_computation(const ftlo_env* e) {
    const ftl_config* c = &e->cfg;
    double res = 0;
    res += c->leader_movement_reward; /* stop_signal is never set */
    if (e->too_close) res += c->too_close_penalty;
    else {
        if (e->is_in_box && e->is_on_trace) res += c->reward_in_box;
        else if (e->is_in_box) res += c->reward_in_dev;
        else if (e->is_on_trace) res += c->reward_on_track;
        else if (e->step_count > c->warm_start) res += c->not_on_track_penalty;
    }
    if (e->crash) res += c->crash_penalty;
    return res;
}

/* ENV:1143-1157.  `random.uniform(lo, hi)` = lo + (hi-lo)*random() is drawn from the counter-based stream of
 * include/ftl.h (the golden generator patches random.uniform with the same function). */
/* self.frames_per_step: fixed, or the last np.random.randint(lo, hi) draw (ENV:405, 939-940) from the counter stream */
static int cur_fps(const ftlo_env* e) { return e->cfg.rand_fps_hi > 0 ? e->fps : e->cfg.frames_per_step; }

static double process_leader_speed_regime(ftlo_env* e) {
    const ftl_config* c = &e->cfg;
    int sel = -1;
    for (int i = 0; i < c->n_speed_regime; i++) if (c->speed_key[i] <= e->step_count) sel = i;   /* dict order, last match wins */
    if (sel >= 0) {
        if (c->speed_is_range[sel]) {
            double u = ftl_uniform01(c->rng_seed, (uint64_t)c->env_id_base, e->resets, (uint64_t)e->step_count);
            e->cur_mult = c->speed_lo[sel] + (c->speed_hi[sel] - c->speed_lo[sel]) * u;
        } else e->cur_mult = c->speed_lo[sel];
    }
    return e->rb[0].p->max_speed * e->cur_mult;
}
/* ENV:1159-1174 */
static double process_leader_acceleration_regime(ftlo_env* e) {
    const ftl_config* c = &e->cfg;
    for (int i = 0; i < c->n_acc_regime; i++)
        if (!((e->acc_consumed >> i) & 1u) && c->acc_key[i] <= e->step_count) {
            e->cur_acc = c->acc_val[i]; e->cum_speed = e->cur_acc;
            e->acc_consumed |= 1u << i;                      /* del self.leader_acceleration_regime[cur_key] */
        }
    e->cum_speed += e->cur_acc;
    return e->cum_speed * e->rb[0].p->max_speed;
}

/* ENV:947-1141; returns the frame reward, writes info */
static double frame_step(ftlo_env* e, uint8_t info[3]) {
    const ftl_config* c = &e->cfg;
    robot_t* leader = &e->rb[0]; robot_t* follower = &e->rb[1];
    e->is_in_box = 0; e->is_on_trace = 0;
    info[0] = FTL_MISSION_IN_PROGRESS; info[1] = FTL_AGENT_MOVING; info[2] = FTL_LEADER_MOVING;

    robot_move(follower);                                              /* ENV:957 */
    if (!c->ignore_follower_collisions && collision_check(e, 1)) {     /* ENV:960-964 */
        e->crash = 1; e->done = 1; info[0] = FTL_MISSION_FAIL; info[1] = FTL_AGENT_CRASH;
    }
    int G = trajectory_in_box(e);                                      /* ENV:968-969 */
    e->green_count = G;
    check_agent_position(e, G);                                        /* ENV:973 */

    if (euclid_f64((double)leader->px, (double)leader->py, e->cur_target[0], e->cur_target[1]) < c->leader_pos_epsilon) {
        e->cur_target_id += 1;                                         /* ENV:978-983 */
        if (e->cur_target_id >= e->route_len) e->leader_finished = 1;
        else { e->cur_target[0] = e->route[2 * e->cur_target_id]; e->cur_target[1] = e->route[2 * e->cur_target_id + 1]; }
    }
    for (int b = 0; b < c->n_bears; b++) {                             /* ENV:987-995 */
        double tx, ty;
        if (c->move_bear_v4 && (b % 2)) move_bear_v4(e, b, &tx, &ty);
        else choose_points_for_bear_stat(e, b, &tx, &ty);
        e->bear_pt[b][0] = tx; e->bear_pt[b][1] = ty;
        move_to_the_point(&e->rb[2 + b], tx, ty, 0, 0.0);
    }
    if (!e->leader_finished) {                                         /* ENV:1048-1058 */
        double speed = (c->n_speed_regime >= 0) ? process_leader_speed_regime(e) : leader->p->max_speed;
        double acceleration = (c->n_acc_regime >= 0) ? process_leader_acceleration_regime(e) / cur_fps(e) : 0;
        move_to_the_point(leader, e->cur_target[0], e->cur_target[1], 1, speed + acceleration);
    } else {                                                           /* ENV:1062-1065 */
        command_forward(leader, 0); command_turn(leader, 0, 0); info[2] = FTL_LEADER_FINISHED;
    }
    if (collision_check(e, 0)) { e->done = 1; info[0] = FTL_MISSION_FAIL; info[2] = FTL_LEADER_CRASH; } /* ENV:1068-1072 */

    /* ENV:1074-1075 with the deterministic tick of SURVEY Appendix B.4: frame k sees get_ticks()==k */
    if ((e->step_count + 1) % c->trajectory_saving_period == 0) {
        if (e->traj_len < e->traj_cap) { e->traj[2 * e->traj_len] = leader->px; e->traj[2 * e->traj_len + 1] = leader->py; e->traj_len++; }
        else e->error |= FTL_ERR_TRAJ_OVERFLOW;
    }
    if (e->leader_finished && e->is_in_box) {                          /* ENV:1077-1087 */
        if (e->finish_timer < 0) e->finish_timer = 0;
        else {
            e->finish_timer += 1;
            if (e->finish_timer > cur_fps(e) * 20) {
                info[0] = FTL_MISSION_SUCCESS; info[2] = FTL_LEADER_FINISHED; info[1] = FTL_AGENT_FINISHED; e->done = 1;
            }
        }
    }
    if (e->step_count > c->warm_start) {                               /* ENV:1088-1107 */
        if (c->has_low_reward && e->accumulated_penalty < c->low_reward) {
            info[0] = FTL_MISSION_FAIL; info[2] = FTL_LEADER_MOVING; info[1] = FTL_AGENT_LOW_REWARD; e->crash = 1; e->done = 1;
        }
        if (c->has_max_distance_coef) {
            /* np.linalg.norm(f32 - f32) is float32; the Python-float threshold is weak => compared in f32 */
            float dx = follower->px - leader->px, dy = follower->py - leader->py;
            float nrm = sqrtf(dx * dx + dy * dy);
            if (nrm > (float)(c->max_distance * c->max_distance_coef)) {
                info[0] = FTL_MISSION_FAIL; info[2] = FTL_LEADER_MOVING; info[1] = FTL_AGENT_TOO_FAR; e->crash = 1; e->done = 1;
            }
        }
    }
    double res = reward_computation(e);                                /* ENV:1109-1115 */
    if (res < 0) e->accumulated_penalty += res; else e->accumulated_penalty = 0;
    e->overall_reward += res;
    e->step_count += 1;                                                /* ENV:1127-1134 */
    if (e->step_count > c->max_steps) {
        info[0] = FTL_MISSION_FINISHED_BY_TIME; info[2] = FTL_LEADER_MOVING; info[1] = FTL_AGENT_MOVING; e->done = 1;
    }
    return c->aggregate_reward ? e->overall_reward : res;
}

/* ------------------------------------------------------------------ LeaderPositionsTracker_v2 (SEN:243-327) */

static void hist_push(ftlo_env* e, double x, double y, int isf64) {
    if (e->hist_len >= e->hist_cap) { e->error |= FTL_ERR_CORR_OVERFLOW; return; }
    e->hist[2 * e->hist_len] = x; e->hist[2 * e->hist_len + 1] = y; e->hist_f64[e->hist_len] = (uint8_t)isf64; e->hist_len++;
}
static void corr_push(ftlo_env* e, const double r[2], const double l[2]) {
    if (e->corr_len >= e->hist_cap) { e->error |= FTL_ERR_CORR_OVERFLOW; return; }
    double* q = e->corr + 4 * e->corr_len; q[0] = r[0]; q[1] = r[1]; q[2] = l[0]; q[3] = l[1]; e->corr_len++;
}
static void deque_popleft(ftlo_env* e) {
    memmove(e->hist, e->hist + 2, sizeof(double) * 2 * (size_t)(e->hist_len - 1));
    memmove(e->hist_f64, e->hist_f64 + 1, (size_t)(e->hist_len - 1));
    e->hist_len--;
    if (e->corr_len > 0) { memmove(e->corr, e->corr + 4, sizeof(double) * 4 * (size_t)(e->corr_len - 1)); e->corr_len--; }
    else e->error |= FTL_ERR_TRACKER_SEED; /* reference: IndexError pop from an empty deque */
}
/* SEN:288-290 / 295-297: np.sum(np.linalg.norm(diff(np.array(hist)), axis=1)); array dtype is f64 as soon as
 * one f64 point is present, else f32.  Returned as double (exact for the f32 case). */
static double hist_path_length(const ftlo_env* e) {
    int m = e->hist_len, any64 = 0;
    if (m < 2) return 0.0;
    for (int i = 0; i < m; i++) any64 |= e->hist_f64[i];
    if (any64) {
        double* d = (double*)malloc(sizeof(double) * (size_t)(m - 1));
        for (int i = 0; i < m - 1; i++) {
            double dx = e->hist[2 * i] - e->hist[2 * i + 2], dy = e->hist[2 * i + 1] - e->hist[2 * i + 3];
            d[i] = sqrt(dx * dx + dy * dy);
        }
        double s = pairwise_f64(d, m - 1); free(d); return s;
    }
    float* d = (float*)malloc(sizeof(float) * (size_t)(m - 1));
    for (int i = 0; i < m - 1; i++) {
        float dx = (float)e->hist[2 * i] - (float)e->hist[2 * i + 2], dy = (float)e->hist[2 * i + 1] - (float)e->hist[2 * i + 3];
        d[i] = sqrtf(dx * dx + dy * dy);
    }
    float s = pairwise_f32(d, m - 1); free(d); return (double)s;
}
/* SEN:302-317: border pair from the vector hist[i1]-hist[i0], anchored at hist[ia] */
static void border_pair(ftlo_env* e, int i1, int i0, int ia) {
    double vx, vy;
    const double* p1 = e->hist + 2 * i1; const double* p0 = e->hist + 2 * i0;
    if (e->hist_f64[i1] || e->hist_f64[i0]) {
        vx = p1[0] - p0[0]; vy = p1[1] - p0[1];
        double nrm = sqrt(fma(vy, vy, vx * vx));          /* np.linalg.norm 1-D f64: sqrt(ddot) */
        double sc = e->cfg.corridor_width / nrm;
        vx *= sc; vy *= sc;
    } else {
        float fx = (float)p1[0] - (float)p0[0], fy = (float)p1[1] - (float)p0[1];
        float nrm = sqrtf(fx * fx + fy * fy);
        float sc = (float)e->cfg.corridor_width / nrm;      /* python int / np.float32 -> float32 */
        fx *= sc; fy *= sc; vx = (double)fx; vy = (double)fy;
    }
    /* rotateVector(v, +90) / (v, -90), MISC:47-53 */
    double c90 = cos(90.0 * DEG2RAD), s90 = sin(90.0 * DEG2RAD);
    double cm90 = cos(-90.0 * DEG2RAD), sm90 = sin(-90.0 * DEG2RAD);
    const double* a = e->hist + 2 * ia;
    double r[2] = { (c90 * vx + (-s90) * vy) + a[0], (s90 * vx + c90 * vy) + a[1] };
    double l[2] = { (cm90 * vx + (-sm90) * vy) + a[0], (sm90 * vx + cm90 * vy) + a[1] };
    corr_push(e, r, l);
}
static void tracker_scan(ftlo_env* e) {
    const ftl_config* c = &e->cfg;
    const robot_t* leader = &e->rb[0]; const robot_t* f = &e->rb[1];
    if (e->trk_counter % c->tracker_saving_period == 0) {
        if (e->hist_len > 0 && e->hist[2 * (e->hist_len - 1)] == (double)leader->px &&
            e->hist[2 * (e->hist_len - 1) + 1] == (double)leader->py)
            return; /* SEN:247-251: no counter increment */
        if (e->hist_len == 0 && e->trk_counter == 0) {
            double sx, sy;
            if (c->tracker_start_behind) {                      /* SEN:257-272 */
                double th = angle_correction(f->direction + 180.0) * DEG2RAD;
                sx = 50 * cos(th) + (double)f->px; sy = 50 * sin(th) + (double)f->py;
            } else { sx = (double)f->px; sy = (double)f->py; }
            /* without start_behind both operands are f32 (SEN:275-277) */
            double dist = c->tracker_start_behind ? euclid_f64(sx, sy, (double)leader->px, (double)leader->py)
                                                  : euclid_f32(f->px, f->py, leader->px, leader->py);
            int n = (int)(dist / ((double)(c->tracker_saving_period * 5) * leader->p->max_speed));
            if (n < 2) { e->error |= FTL_ERR_TRACKER_SEED; e->trk_counter += 1; return; }
            int f64pts = c->tracker_start_behind; /* np.linspace(f32, f32) stays f32 under numpy 2 */
            if (f64pts) {
                double stepx = ((double)leader->px - sx) / (n - 1), stepy = ((double)leader->py - sy) / (n - 1);
                for (int i = 0; i < n; i++) {
                    double x = (stepx == 0) ? ((double)i / (n - 1)) * ((double)leader->px - sx) + sx : (double)i * stepx + sx;
                    double y = (stepy == 0) ? ((double)i / (n - 1)) * ((double)leader->py - sy) + sy : (double)i * stepy + sy;
                    if (i == n - 1) { x = (double)leader->px; y = (double)leader->py; }
                    hist_push(e, x, y, 1);
                }
            } else {
                float fsx = f->px, fsy = f->py;
                float stepx = (leader->px - fsx) / (float)(n - 1), stepy = (leader->py - fsy) / (float)(n - 1);
                for (int i = 0; i < n; i++) {
                    float x = (stepx == 0) ? ((float)i / (float)(n - 1)) * (leader->px - fsx) + fsx : (float)i * stepx + fsx;
                    float y = (stepy == 0) ? ((float)i / (float)(n - 1)) * (leader->py - fsy) + fsy : (float)i * stepy + fsy;
                    if (i == n - 1) { x = leader->px; y = leader->py; }
                    hist_push(e, (double)x, (double)y, 0);
                }
            }
        } else hist_push(e, (double)leader->px, (double)leader->py, 0); /* SEN:286 */

        double path = hist_path_length(e);                        /* SEN:288-297 */
        while (path > c->corridor_length) { deque_popleft(e); path = hist_path_length(e); }

        if (e->hist_len > 1) {                                   /* SEN:299-317 */
            int m = e->hist_len;
            if (e->trk_counter == 0)
                for (int i = m - 1; i > 0; i--) border_pair(e, i, i - 1, m - i - 1);
            border_pair(e, m - 1, m - 2, m - 2);
        }
    }
    e->trk_counter += 1;
}

/* ------------------------------------------------------------------ ray sensor (SEN:608-673, 883-968) */

static void push_seg(seg_t* s, int* n, double ax, double ay, double bx, double by) {
    s[*n].a[0] = (float)ax; s[*n].a[1] = (float)ay; s[*n].b[0] = (float)bx; s[*n].b[1] = (float)by; (*n)++;
}
static void rect_edges(seg_t* s, int* n, int x, int y, int w, int h) { /* SEN:668-671 */
    double l = x, t = y, r = x + w, b = y + h;
    push_seg(s, n, l, b, r, b); /* bottomleft, bottomright */
    push_seg(s, n, r, t, r, b); /* topright, bottomright */
    push_seg(s, n, r, t, l, t); /* topright, topleft */
    push_seg(s, n, l, b, l, t); /* bottomleft, topleft */
}
/* SEN:642-673 */
static seg_t* collect_obstacle_edges(const ftlo_env* e, const ftl_laser_cfg* L, int* out_n) {
    int C = e->corr_len, cap = 2 * (C > 0 ? C : 1) + 2 + 4 * (1 + e->cfg.n_static + e->cfg.n_bears) + 4;
    seg_t* s = (seg_t*)malloc(sizeof(seg_t) * (size_t)cap);
    int n = 0;
    if (L->react_corridor)
        for (int i = 0; i < C - 1; i++) {
            const double* p = e->corr + 4 * i; const double* q = e->corr + 4 * (i + 1);
            push_seg(s, &n, p[0], p[1], q[0], q[1]);
            push_seg(s, &n, p[2], p[3], q[2], q[3]);
        }
    if (L->react_green) {
        const double* p = e->corr; const double* q = e->corr + 4 * (C - 1);
        push_seg(s, &n, p[0], p[1], p[2], p[3]);
        push_seg(s, &n, q[0], q[1], q[2], q[3]);
    }
    if (L->react_obstacles) {
        int use_static = (L->react_obstacles == 1 || L->react_obstacles == 2);
        int use_dynamic = (L->react_obstacles == 1 || L->react_obstacles == 3);
        if (use_static) { /* game_object_list: leader, (follower skipped), walls, rocks */
            rect_edges(s, &n, e->rb[0].rx, e->rb[0].ry, e->rb[0].rw, e->rb[0].rh);
            for (int k = 0; k < e->cfg.n_static; k++) rect_edges(s, &n, e->srect[4 * k], e->srect[4 * k + 1], e->srect[4 * k + 2], e->srect[4 * k + 3]);
        }
        if (use_dynamic)
            for (int b = 0; b < e->cfg.n_bears; b++) rect_edges(s, &n, e->rb[2 + b].rx, e->rb[2 + b].ry, e->rb[2 + b].rw, e->rb[2 + b].rh);
    }
    *out_n = n;
    return s;
}

/* SEN:608-614 with the dtype flow of SURVEY A.6: A,B f32 segment ends, C f32 origin, D f64 ray end */
static int seg_hit(const seg_t* s, float cx, float cy, double dx, double dy) {
    float ax = s->a[0], ay = s->a[1], bx = s->b[0], by = s->b[1];
    int t1 = (dy - (double)ay) * (double)(cx - ax) > (double)(cy - ay) * (dx - (double)ax);   /* ccw(A,C,D) */
    int t2 = (dy - (double)by) * (double)(cx - bx) > (double)(cy - by) * (dx - (double)bx);   /* ccw(B,C,D) */
    int t3 = (cy - ay) * (bx - ax) > (by - ay) * (cx - ax);                                   /* ccw(A,B,C) all f32 */
    int t4 = (dy - (double)ay) * (double)(bx - ax) > (double)(by - ay) * (dx - (double)ax);   /* ccw(A,B,D) */
    return (t1 != t2) && (t3 != t4);
}
/* SEN:626-640: squared-free distance from the origin to the intersection point */
static double seg_hit_distance(const seg_t* s, float cx, float cy, double ex, double ey, double* ox, double* oy) {
    float dax = s->b[0] - s->a[0], day = s->b[1] - s->a[1];
    double dbx = ex - (double)cx, dby = ey - (double)cy;
    float dpx = s->a[0] - cx, dpy = s->a[1] - cy;
    float dapx = -day, dapy = dax;
    double denom = (double)dapx * dbx + (double)dapy * dby;
    float num = dapx * dpx + dapy * dpy;
    double t = (double)num / denom;
    double x = t * dbx + (double)cx, y = t * dby + (double)cy;
    *ox = x; *oy = y;
    double qx = x - (double)cx, qy = y - (double)cy;
    return sqrt(qx * qx + qy * qy);
}

static void laser_scan(ftlo_env* e, int k, float* out) {
    const ftl_laser_cfg* L = &e->cfg.lasers[k];
    const robot_t* f = &e->rb[1];
    int N = L->count, H = L->history, W = laser_width(L);
    double period = 360.0 / N;
    if (e->corr_len <= 1) {          /* SEN:893/962: Prev_lasers_v2 raises UnboundLocalError; lasers_v2 (SEN:779, 803-806) reads laser_length */
        if (!L->lenient) e->error |= FTL_ERR_EMPTY_CORRIDOR;
        for (int i = 0; i < H * W; i++) out[i] = (float)L->length;
        return;
    }
    if (L->pad_sectors) for (int i = 0; i < H * W; i++) out[i] = 0.0f;   /* np.zeros sector rows, SEN:933-936 */
    /* SEN:896-897 */
    snapshot_t* hs = e->snaps[k];
    free(hs[0].segs);
    memmove(hs, hs + 1, sizeof(snapshot_t) * (size_t)(H - 1));
    hs[H - 1].segs = collect_obstacle_edges(e, L, &hs[H - 1].n);
    hs[H - 1].valid = 1;
    for (int i = 0; i < N; i++) {
        /* SEN:888-891; LeaderCorridor_lasers (SEN:609-632): direction + a fixed angle per ray */
        double th = (L->explicit_angles ? (f->direction + L->ray_angles[i]) : ((f->direction + L->angle_offset) + i * period)) * DEG2RAD;
        double ex = (double)f->px + cos(th) * L->length, ey = (double)f->py + sin(th) * L->length;
        for (int j = 0; j < H; j++) {
            double bx = ex, by = ey; int found = 0; double best = 0;
            if (hs[j].valid)
                for (int m = 0; m < hs[j].n; m++)
                    if (seg_hit(&hs[j].segs[m], f->px, f->py, ex, ey)) {
                        double x, y, d = seg_hit_distance(&hs[j].segs[m], f->px, f->py, ex, ey, &x, &y);
                        if (!found || d < best) { best = d; bx = x; by = y; found = 1; } /* argmin: first minimum */
                    }
            double qx = bx - (double)f->px, qy = by - (double)f->py;
            float val = (float)sqrt(fma(qy, qy, qx * qx));                         /* SEN:930 np.linalg.norm 1-D */
            int col = i;
            if (L->pad_sectors) {                                                   /* SEN:938-953 */
                double lis = (double)N / 4;
                int sector = ((double)i < lis) ? 0 : ((double)i < 2 * lis) ? 1 : ((double)i < 3 * lis) ? 2 : 3;
                col = sector * N + i;
            }
            out[j * W + col] = val;
        }
    }
}


/* ------------------------------------------------------------------ LeaderPositionsTracker v1 (SEN:148-229) */
/* hist = list of float32 points (stored in e->hist as doubles with hist_f64 = 0); corridor pairs are appended and never
 * popped; eat_close_points deletes history points within max(width, height) of the follower after every scan. */
static void tracker1_scan(ftlo_env* e) {
    const ftl_config* c = &e->cfg;
    const robot_t* leader = &e->rb[0]; const robot_t* f = &e->rb[1];
    if (e->trk_counter % c->tracker_saving_period == 0) {
        if (e->hist_len > 0 && e->hist[2 * (e->hist_len - 1)] == (double)leader->px && e->hist[2 * (e->hist_len - 1) + 1] == (double)leader->py)
            return;                                                       /* SEN:178-183: nothing at all happens */
        if (e->hist_len == 0) hist_push(e, (double)f->px, (double)f->py, 0);   /* SEN:185-186 */
        hist_push(e, (double)leader->px, (double)leader->py, 0);           /* SEN:187 */
        if (e->hist_len > 1) {                                            /* SEN:188-195 */
            const double* p1 = e->hist + 2 * (e->hist_len - 1); const double* p0 = e->hist + 2 * (e->hist_len - 2);
            float fx = (float)p1[0] - (float)p0[0], fy = (float)p1[1] - (float)p0[1];
            float nrm = sqrtf(fx * fx + fy * fy);
            float sc = (float)c->corridor_width / nrm;                    /* env.max_dev (python float) / np.float32 -> float32 */
            fx *= sc; fy *= sc;
            double vx = (double)fx, vy = (double)fy;
            double c90 = cos(90.0 * DEG2RAD), s90 = sin(90.0 * DEG2RAD), cm90 = cos(-90.0 * DEG2RAD), sm90 = sin(-90.0 * DEG2RAD);
            double r[2] = { (c90 * vx + (-s90) * vy) + p0[0], (s90 * vx + c90 * vy) + p0[1] };
            double l[2] = { (cm90 * vx + (-sm90) * vy) + p0[0], (sm90 * vx + cm90 * vy) + p0[1] };
            corr_push(e, r, l);
        }
    }
    e->trk_counter += 1;
    if (c->trk1_eat_close_points && e->hist_len > 0) {                    /* SEN:211-216 */
        int w = 0;
        const float thr = (float)c->trk1_eat_radius;                      /* float32 norms <= python float: compared in float32 */
        for (int i = 0; i < e->hist_len; i++) {
            float dx = (float)e->hist[2 * i] - f->px, dy = (float)e->hist[2 * i + 1] - f->py;
            float nrm = sqrtf(dx * dx + dy * dy);
            if (nrm <= thr) continue;
            e->hist[2 * w] = e->hist[2 * i]; e->hist[2 * w + 1] = e->hist[2 * i + 1]; e->hist_f64[w] = 0; w++;
        }
        e->hist_len = w;
    }
}

/* ------------------------------------------------------------------ LeaderCorridor_lasers_compas (SEN:1138-1288) */
/* Corridor walls only, kept in float64 (no float32 cast as at SEN:672); wall list order = front, back, left walls, right walls
 * (SEN:1166); the nearest hit's wall orientation selects one of four N-wide blocks after the "no wall" block. */
typedef struct { double a[2], b[2]; int cls; } wall_t;    /* cls: 0 front, 1 back, 2 left, 3 right */
typedef struct { wall_t* w; int n; int valid; } wsnap_t;

static int wall_hit(const wall_t* s, float cxf, float cyf, double dx, double dy) {   /* SEN:608-614, every operand float64 */
    double ax = s->a[0], ay = s->a[1], bx = s->b[0], by = s->b[1], cx = (double)cxf, cy = (double)cyf;
    int t1 = (dy - ay) * (cx - ax) > (cy - ay) * (dx - ax);   /* ccw(A,C,D) */
    int t2 = (dy - by) * (cx - bx) > (cy - by) * (dx - bx);   /* ccw(B,C,D) */
    int t3 = (cy - ay) * (bx - ax) > (by - ay) * (cx - ax);   /* ccw(A,B,C) */
    int t4 = (dy - ay) * (bx - ax) > (by - ay) * (dx - ax);   /* ccw(A,B,D) */
    return (t1 != t2) && (t3 != t4);
}
static double wall_hit_point(const wall_t* s, float cxf, float cyf, double ex, double ey, double* ox, double* oy) {   /* SEN:626-640 */
    double cx = (double)cxf, cy = (double)cyf;
    double dax = s->b[0] - s->a[0], day = s->b[1] - s->a[1];
    double dbx = ex - cx, dby = ey - cy;
    double dpx = s->a[0] - cx, dpy = s->a[1] - cy;
    double dapx = -day, dapy = dax;
    double denom = dapx * dbx + dapy * dby;
    double num = dapx * dpx + dapy * dpy;
    double t = num / denom;
    double x = t * dbx + cx, y = t * dby + cy;
    *ox = x; *oy = y;
    double qx = x - cx, qy = y - cy;
    return sqrt(qx * qx + qy * qy);                       /* np.linalg.norm(x - position, axis=1) */
}
static void compas_scan(ftlo_env* e, int k, float* out) {
    const ftl_laser_cfg* L = &e->cfg.lasers[k];
    const robot_t* f = &e->rb[1];
    const int N = L->count, H = L->history, W = 5 * N;
    const double period = 360.0 / N;
    if (e->corr_len <= 1) {          /* SEN:1192/1244: all_obs_arr is unbound -> UnboundLocalError */
        e->error |= FTL_ERR_EMPTY_CORRIDOR;
        for (int i = 0; i < H * W; i++) out[i] = (i % W) < N ? (float)L->length : 0.0f;
        return;
    }
    wsnap_t* hs = (wsnap_t*)e->snaps[k];
    free(hs[0].w);
    memmove(hs, hs + 1, sizeof(wsnap_t) * (size_t)(H - 1));
    {   /* collect_obstacle_edges, SEN:1153-1176 */
        const int C = e->corr_len;
        wall_t* w = (wall_t*)malloc(sizeof(wall_t) * (size_t)(2 * C + 2));
        int n = 0;
        const double* first = e->corr; const double* last = e->corr + 4 * (C - 1);
        w[n].a[0] = last[0]; w[n].a[1] = last[1]; w[n].b[0] = last[2]; w[n].b[1] = last[3]; w[n].cls = 0; n++;      /* front: corridor[-1] */
        w[n].a[0] = first[0]; w[n].a[1] = first[1]; w[n].b[0] = first[2]; w[n].b[1] = first[3]; w[n].cls = 1; n++;  /* back: corridor[0] */
        for (int i = 0; i < C - 1; i++) { const double* p = e->corr + 4 * i; const double* q = p + 4;
            w[n].a[0] = p[2]; w[n].a[1] = p[3]; w[n].b[0] = q[2]; w[n].b[1] = q[3]; w[n].cls = 2; n++; }             /* left walls */
        for (int i = 0; i < C - 1; i++) { const double* p = e->corr + 4 * i; const double* q = p + 4;
            w[n].a[0] = p[0]; w[n].a[1] = p[1]; w[n].b[0] = q[0]; w[n].b[1] = q[1]; w[n].cls = 3; n++; }             /* right walls */
        hs[H - 1].w = w; hs[H - 1].n = n; hs[H - 1].valid = 1;
    }
    for (int i = 0; i < H * W; i++) out[i] = 0.0f;
    for (int i = 0; i < N; i++) {
        double th = ((f->direction + L->angle_offset) + i * period) * DEG2RAD;
        double ex = (double)f->px + cos(th) * L->length, ey = (double)f->py + sin(th) * L->length;
        for (int j = 0; j < H; j++) {
            double bx = ex, by = ey; int found = 0, cls = -1; double best = 0;
            if (hs[j].valid)
                for (int m = 0; m < hs[j].n; m++)
                    if (wall_hit(&hs[j].w[m], f->px, f->py, ex, ey)) {
                        double x, y, d = wall_hit_point(&hs[j].w[m], f->px, f->py, ex, ey, &x, &y);
                        if (!found || d < best) { best = d; bx = x; by = y; cls = hs[j].w[m].cls; found = 1; }   /* argmin: first minimum */
                    }
            double qx = bx - (double)f->px, qy = by - (double)f->py;
            float val = (float)sqrt(fma(qy, qy, qx * qx));                         /* np.linalg.norm 1-D, SEN:1229-1243 */
            out[j * W + (found ? (1 + cls) * N : 0) + i] = val;                    /* the "no wall" slot stays 0 on a hit */
        }
    }
}

/* ------------------------------------------------------------------ LaserSensor (SEN:18-145) */
static void lidar_scan(const ftlo_env* e, const ftl_aux_cfg* A, float* out) {
    const robot_t* f = &e->rb[1];
    /* objects_in_range, SEN:72-79: every object but the follower whose distance_to_rect (MISC:29-44: nearest of the 4 corners and
     * 4 edge mid-points, scipy euclidean on a float32 point and an int tuple = float64 dnrm2) is within range + 3 PIXELS_TO_METER */
    int nobj = 1 + e->cfg.n_static + e->cfg.n_bears, nin = 0;
    int (*rc)[4] = (int (*)[4])malloc(sizeof(int[4]) * (size_t)nobj);
    for (int o = 0; o < nobj; o++) {
        int x, y, w, h;
        if (o == 0) { x = e->rb[0].rx; y = e->rb[0].ry; w = e->rb[0].rw; h = e->rb[0].rh; }
        else if (o <= e->cfg.n_static) { const int32_t* q = e->srect + 4 * (o - 1); x = q[0]; y = q[1]; w = q[2]; h = q[3]; }
        else { const robot_t* b = &e->rb[2 + (o - 1 - e->cfg.n_static)]; x = b->rx; y = b->ry; w = b->rw; h = b->rh; }
        const int px[8] = { x, x, x + w, x + w, x + (w >> 1), x, x + (w >> 1), x + w };
        const int py[8] = { y, y + h, y, y + h, y, y + (h >> 1), y + h, y + (h >> 1) };
        double dmin = INFINITY;
        for (int k = 0; k < 8; k++) { double d = euclid_f64((double)f->px, (double)f->py, (double)px[k], (double)py[k]); if (d < dmin) dmin = d; }
        if (dmin <= A->in_range_px) { rc[nin][0] = x; rc[nin][1] = y; rc[nin][2] = w; rc[nin][3] = h; nin++; }
    }
    const float x1 = f->px, y1 = f->py;
    /* return_all_points (SEN:112-113): every marching point up to and including the first hit goes to the list, nothing else; the block is
     * [count][points or distances][zeros] */
    const int wd = A->return_only_distances ? 1 : 2;
    int K = 0;
    if (A->return_all_points) for (int i = 0; i < 1 + A->n_angles * A->points_number * wd; i++) out[i] = 0.0f;
    for (int a = 0; a < A->n_angles; a++) {
        /* SEN:88-101: -direction, then +- k * angle_step (angle_correction on those) */
        double angle = -f->direction;
        if (a > 0) { double k = (double)((a + 1) / 2) * A->angle_step; angle = angle_correction(a & 1 ? -f->direction + k : -f->direction - k); }
        const double rad = angle * DEG2RAD;
        /* np.float32 + python float -> float32 (NEP 50): the float64 offset is rounded to float32 first */
        const float x2 = x1 + (float)(A->range_px * cos(rad)), y2 = y1 - (float)(A->range_px * sin(rad));
        float ptx = x2, pty = y2;
        for (int i = 0; i < A->points_number; i++) {
            const double u = (double)i / (double)A->points_number;
            const float cx = x2 * (float)u + x1 * (float)(1.0 - u), cy = y2 * (float)u + y1 * (float)(1.0 - u);
            if (A->return_all_points) {
                const float ddx = cx - x1, ddy = cy - y1;
                if (wd == 1) out[1 + K] = sqrtf(ddx * ddx + ddy * ddy); else { out[1 + 2 * K] = ddx; out[2 + 2 * K] = ddy; }
                K++;
            }
            int hit = 0;
            for (int o = 0; o < nin && !hit; o++)         /* Rect.collidepoint: x <= px < x + w and y <= py < y + h */
                hit = (float)rc[o][0] <= cx && cx < (float)(rc[o][0] + rc[o][2]) && (float)rc[o][1] <= cy && cy < (float)(rc[o][1] + rc[o][3]);
            if (hit) { ptx = cx; pty = cy; break; }
        }
        const float dx = ptx - x1, dy = pty - y1;         /* sensed_points - position, float32 */
        if (A->return_all_points) continue;
        if (A->return_only_distances) out[a] = sqrtf(dx * dx + dy * dy);
        else { out[2 * a] = dx; out[2 * a + 1] = dy; }
    }
    if (A->return_all_points) out[0] = (float)K;
    free(rc);
}

/* ------------------------------------------------------------------ LeaderTrackDetector_vector / _radar (SEN:342-487) */
/* the slice of the tracked positions a detector looks at: "new" = the last seq_len points, "old" = the first, "near" = all */
static void track_slice(const ftlo_env* e, const ftl_aux_cfg* A, int* lo, int* hi) {
    int n = e->hist_len;
    if (A->detectable == 0) { *lo = n - A->seq_len > 0 ? n - A->seq_len : 0; *hi = n; }
    else if (A->detectable == 1) { *lo = 0; *hi = n < A->seq_len ? n : A->seq_len; }
    else { *lo = 0; *hi = n; }
}
static void track_vector_scan(const ftlo_env* e, const ftl_aux_cfg* A, float* out) {
    const robot_t* f = &e->rb[1];
    for (int i = 0; i < 2 * A->seq_len; i++) out[i] = 0.0f;
    int lo, hi; track_slice(e, A, &lo, &hi);
    /* np.array(slice) - position: float32 - float32 when every point of the slice is float32, else float64; either way the
     * float32 store equals the correctly rounded difference of the two values */
    for (int i = lo; i < hi; i++) {
        out[2 * (i - lo)] = (float)(e->hist[2 * i] - (double)f->px);
        out[2 * (i - lo) + 1] = (float)(e->hist[2 * i + 1] - (double)f->py);
    }
}
static void track_radar_scan(const ftlo_env* e, const ftl_aux_cfg* A, float* out) {
    const robot_t* f = &e->rb[1];
    const int S = A->radar_sectors;
    for (int i = 0; i < S; i++) out[i] = 0.0f;
    if (e->hist_len == 0) return;
    int lo, hi; track_slice(e, A, &lo, &hi);
    /* followerDirVec / followerRightVec = rotateVector([1, 0], angle): (cos, sin) of the angle, SEN:424-428 */
    const double dth = f->direction * DEG2RAD;
    double rdir = f->direction + 90; if (rdir >= 360) rdir -= 360;
    const double rth = rdir * DEG2RAD;
    const double dvx = cos(dth) * 1 + (-sin(dth)) * 0, dvy = sin(dth) * 1 + cos(dth) * 0;
    const double rvx = cos(rth) * 1 + (-sin(rth)) * 0, rvy = sin(rth) * 1 + cos(rth) * 0;
    const double nd = sqrt(fma(dvy, dvy, dvx * dvx)), nr = sqrt(fma(rvy, rvy, rvx * rvx));     /* np.linalg.norm 1-D float64 */
    int any64 = 0;
    for (int i = lo; i < hi; i++) any64 |= e->hist_f64[i];
    const double sa = PI_D / S;
    double* best = (double*)malloc(sizeof(double) * (size_t)S);
    for (int s = 0; s < S; s++) best[s] = INFINITY;
    for (int i = lo; i < hi; i++) {
        double vx, vy, dist;
        if (any64) { vx = e->hist[2 * i] - (double)f->px; vy = e->hist[2 * i + 1] - (double)f->py; dist = sqrt(vx * vx + vy * vy); }
        else { float fx = (float)e->hist[2 * i] - f->px, fy = (float)e->hist[2 * i + 1] - f->py; vx = fx; vy = fy; dist = (double)sqrtf(fx * fx + fy * fy); }
        const double ad = acos((vx * dvx + vy * dvy) / (dist * nd));       /* calculateAngle, MISC:56-63 */
        double ar = acos((vx * rvx + vy * rvy) / (dist * nr));
        if (ad > PI_D / 2) ar = -ar;
        for (int s = 0; s < S; s++)
            if (ar >= sa * s && ar < sa * (s + 1) && dist < best[s]) best[s] = dist;     /* np.min over the sector */
    }
    for (int s = 0; s < S; s++) if (best[s] != INFINITY) out[s] = (float)best[s];
    free(best);
}

/* CLS:255-288 */
static void use_sensors(ftlo_env* e, float* lasers_out) {
    const ftl_config* c = &e->cfg;
    if (c->has_tracker == 1) tracker1_scan(e);                         /* CLS:257-261: the v1 tracker, once per step */
    if (c->has_tracker == 2) tracker_scan(e);                          /* CLS:263-267 */
    for (int g = 0; g < 2; g++) {
        if (g == 1 && c->has_tracker == 2) tracker_scan(e);            /* CLS:285-286: the v2 tracker again, at its dict position */
        for (int k = 0; k < c->n_lasers; k++) if (c->lasers[k].after_tracker == g) {
            if (c->lasers[k].compas) compas_scan(e, k, lasers_out + c->lasers[k].out_offset);
            else laser_scan(e, k, lasers_out + c->lasers[k].out_offset);
        }
        for (int j = 0; j < c->n_aux; j++) if (c->aux[j].after_tracker == g) {
            float* out = lasers_out + c->aux[j].out_offset;
            if (c->aux[j].kind == FTL_AUX_LIDAR) lidar_scan(e, &c->aux[j], out);
            else if (c->aux[j].kind == FTL_AUX_TRACK_VECTOR) track_vector_scan(e, &c->aux[j], out);
            else if (c->aux[j].kind == FTL_AUX_TRACK_RADAR) track_radar_scan(e, &c->aux[j], out);
        }
    }
}

/* ENV:1789-1810 */
static void get_obs(const ftlo_env* e, float* obs_num, double* target) {
    const robot_t* l = &e->rb[0]; const robot_t* f = &e->rb[1];
    obs_num[0] = l->px; obs_num[1] = l->py; obs_num[2] = (float)l->speed; obs_num[3] = (float)l->direction; obs_num[4] = (float)l->rot_speed;
    obs_num[5] = f->px; obs_num[6] = f->py; obs_num[7] = (float)f->speed; obs_num[8] = (float)f->direction; obs_num[9] = (float)f->rot_speed;
    if (e->route_len > 0 && e->cur_target[0] == e->route[2 * (e->route_len - 1)] && e->cur_target[1] == e->route[2 * (e->route_len - 1) + 1] && e->route_len > 1) {
        target[0] = e->route[2 * (e->route_len - 2)]; target[1] = e->route[2 * (e->route_len - 2) + 1];
    } else { target[0] = e->cur_target[0]; target[1] = e->cur_target[1]; }
}

/* ------------------------------------------------------------------ public (ctypes) API */

int ftlo_lasers_len(const ftl_config* c) { int n = 0; for (int k = 0; k < c->n_lasers; k++) n += c->lasers[k].history * laser_width(&c->lasers[k]); return n; }

ftlo_env* ftlo_create(const ftl_config* cfg) {
    ftlo_env* e = (ftlo_env*)calloc(1, sizeof(ftlo_env));
    e->cfg = *cfg;
    int off = 0;
    for (int k = 0; k < cfg->n_lasers; k++) { e->cfg.lasers[k].out_offset = off; off += cfg->lasers[k].history * laser_width(&cfg->lasers[k]); }
    for (int j = 0; j < cfg->n_aux; j++) {       /* lidar / detector blocks follow the ray sensors' blocks */
        ftl_aux_cfg* a = &e->cfg.aux[j];
        a->out_len = a->kind == FTL_AUX_LIDAR ? (a->return_all_points ? 1 + a->n_angles * a->points_number * (a->return_only_distances ? 1 : 2) : a->n_angles * (a->return_only_distances ? 1 : 2))
                   : a->kind == FTL_AUX_TRACK_VECTOR ? 2 * a->seq_len : a->radar_sectors;
        a->out_offset = off; off += a->out_len;
    }
    e->lasers_len = off;
    e->R = 2 + cfg->n_bears;
    e->srect = (int32_t*)calloc((size_t)(cfg->n_static > 0 ? cfg->n_static : 1) * 4, sizeof(int32_t));
    e->route = (double*)calloc((size_t)cfg->route_cap * 2, sizeof(double));
    e->traj_cap = cfg->traj_cap;
    e->traj = (float*)calloc((size_t)cfg->traj_cap * 2, sizeof(float));
    e->hist_cap = cfg->corr_cap;
    e->hist = (double*)calloc((size_t)cfg->corr_cap * 2, sizeof(double));
    e->hist_f64 = (uint8_t*)calloc((size_t)cfg->corr_cap, 1);
    e->corr = (double*)calloc((size_t)cfg->corr_cap * 4, sizeof(double));
    for (int k = 0; k < cfg->n_lasers; k++) e->snaps[k] = (snapshot_t*)calloc((size_t)cfg->lasers[k].history, sizeof(snapshot_t) > sizeof(wsnap_t) ? sizeof(snapshot_t) : sizeof(wsnap_t));
    e->rb[0].p = &e->cfg.leader; e->rb[1].p = &e->cfg.follower;
    for (int b = 0; b < FTL_MAX_BEARS; b++) e->rb[2 + b].p = &e->cfg.bear;
    return e;
}

void ftlo_destroy(ftlo_env* e) {
    if (!e) return;
    for (int k = 0; k < e->cfg.n_lasers; k++) { for (int j = 0; j < e->cfg.lasers[k].history; j++) free(e->snaps[k][j].segs); free(e->snaps[k]); }
    free(e->srect); free(e->route); free(e->traj); free(e->hist); free(e->hist_f64); free(e->corr); free(e);
}

/* reset(): load the scenario produced by the reference's reset-time generation (ENV:461-539), then the tail of
 * reset() itself: flags (ENV:494-514), fresh sensors (ENV:578-589 creates a new RobotWithSensors), use_sensors
 * (ENV:541) and the first observation (ENV:543). */
int ftlo_reset(ftlo_env* e, const int32_t* static_rects, const float* robot_pos, const double* robot_dir,
               const int32_t* robot_rect, const double* route, int route_len, const float* init_traj, int init_traj_len,
               float* obs_num, float* lasers, double* target) {
    const ftl_config* c = &e->cfg;
    if (route_len > c->route_cap || init_traj_len > c->traj_cap) return FTL_E_INVALID;
    memcpy(e->srect, static_rects, sizeof(int32_t) * 4 * (size_t)c->n_static);
    for (int r = 0; r < e->R; r++) {
        robot_t* q = &e->rb[r];
        q->px = robot_pos[2 * r]; q->py = robot_pos[2 * r + 1];
        q->direction = robot_dir[r]; q->speed = 0; q->rot_speed = 0; q->des_speed = 0; q->des_rot_speed = 0;
        q->rot_dir = 0; q->des_rot_dir = 0;
        q->rx = robot_rect[4 * r]; q->ry = robot_rect[4 * r + 1]; q->rw = robot_rect[4 * r + 2]; q->rh = robot_rect[4 * r + 3];
    }
    memcpy(e->route, route, sizeof(double) * 2 * (size_t)route_len); e->route_len = route_len;
    memcpy(e->traj, init_traj, sizeof(float) * 2 * (size_t)init_traj_len); e->traj_len = init_traj_len;
    e->step_count = 0; e->accumulated_penalty = 0; e->overall_reward = 0;        /* ENV:447-448, 504 */
    e->done = 0; e->crash = 0; e->is_in_box = 0; e->is_on_trace = 0; e->too_close = 0; /* ENV:500-503 */
    e->cur_target_id = 1; e->leader_finished = 0; e->finish_timer = -1;          /* ENV:506-514, 542 */
    e->green_count = 0; e->error = 0;
    e->cur_mult = 1; e->cur_acc = 0; e->cum_speed = 0;                               /* ENV:449, 591-592 */
    if (c->rand_fps_hi > 0 && e->fps == 0)                                       /* the constructor's draw, ENV:405 */
        e->fps = ftl_rand_frames(c->rng_seed, (uint64_t)c->env_id_base, 0, 0, c->rand_fps_lo, c->rand_fps_hi);
    e->resets += 1;
    if (route_len == 0) { e->done = 1; e->cur_target[0] = (double)e->rb[0].px; e->cur_target[1] = (double)e->rb[0].py; }
    else if (route_len > 1) { e->cur_target[0] = route[2]; e->cur_target[1] = route[3]; }
    else return FTL_E_INVALID; /* reference: IndexError at ENV:513 */
    for (int b = 0; b < c->n_bears; b++) { /* ENV:717-718: every bear starts with the LAST bear_start_position */
        e->bear_pt[b][0] = (double)(float)(e->rb[0].px - 150.0f); e->bear_pt[b][1] = (double)(float)(e->rb[0].py - 150.0f);
        e->dyn_index[b] = 0;
    }
    e->hist_len = 0; e->corr_len = 0; e->trk_counter = 0;
    for (int k = 0; k < c->n_lasers; k++)
        for (int j = 0; j < c->lasers[k].history; j++) { free(e->snaps[k][j].segs); e->snaps[k][j].segs = NULL; e->snaps[k][j].n = 0; e->snaps[k][j].valid = 0; }
    use_sensors(e, lasers);
    get_obs(e, obs_num, target);
    return 0;
}

/* step(action): ENV:908-945 with a continuous 2-component action given as f64 */
int ftlo_step(ftlo_env* e, double a0, double a1, float* obs_num, float* lasers, double* target, double* reward,
              uint8_t* done, uint8_t* status) {
    robot_t* f = &e->rb[1];
    command_forward(f, a0);                                            /* ENV:927 */
    if (a1 < 0) command_turn(f, fabs(a1), -1);                         /* ENV:928-933 */
    else if (a1 > 0) command_turn(f, a1, 1);
    else command_turn(f, 0, 0);
    double rew = 0; uint8_t info[3] = {0, 0, 0};
    const int nf = cur_fps(e);
    for (int k = 0; k < nf; k++) rew = frame_step(e, info);            /* ENV:935-936 */
    use_sensors(e, lasers);                                            /* ENV:937 */
    get_obs(e, obs_num, target);                                       /* ENV:938 */
    if (e->cfg.rand_fps_hi > 0)                                        /* ENV:939-940 */
        e->fps = ftl_rand_frames(e->cfg.rng_seed, (uint64_t)e->cfg.env_id_base, (uint64_t)e->resets, (uint64_t)e->step_count, e->cfg.rand_fps_lo, e->cfg.rand_fps_hi);
    *reward = rew; *done = (uint8_t)e->done; status[0] = info[0]; status[1] = info[1]; status[2] = info[2];
    return 0;
}

/* ---- debug getters used by the tests to localise a mismatch ---- */
void ftlo_get_robots(const ftlo_env* e, float* pos, double* dbl, int32_t* ints) {
    for (int r = 0; r < e->R; r++) {
        const robot_t* q = &e->rb[r];
        pos[2 * r] = q->px; pos[2 * r + 1] = q->py;
        dbl[5 * r] = q->direction; dbl[5 * r + 1] = q->speed; dbl[5 * r + 2] = q->rot_speed; dbl[5 * r + 3] = q->des_speed; dbl[5 * r + 4] = q->des_rot_speed;
        ints[6 * r] = q->rx; ints[6 * r + 1] = q->ry; ints[6 * r + 2] = q->rw; ints[6 * r + 3] = q->rh; ints[6 * r + 4] = q->rot_dir; ints[6 * r + 5] = q->des_rot_dir;
    }
}
void ftlo_get_counters(const ftlo_env* e, int64_t* c, double* acc) {
    c[0] = e->step_count; c[1] = e->traj_len; c[2] = e->green_count; c[3] = e->cur_target_id; c[4] = e->leader_finished;
    c[5] = e->is_in_box; c[6] = e->is_on_trace; c[7] = e->too_close; c[8] = e->crash; c[9] = e->done; c[10] = e->finish_timer;
    c[11] = e->trk_counter; c[12] = e->hist_len; c[13] = e->corr_len; c[14] = (int64_t)e->error;
    for (int b = 0; b < FTL_MAX_BEARS; b++) c[15 + b] = e->dyn_index[b];
    acc[0] = e->accumulated_penalty; acc[1] = e->overall_reward;
}
void ftlo_set_env_id(ftlo_env* e, int env_id) { e->cfg.env_id_base = env_id; }
int ftlo_get_tracker(const ftlo_env* e, double* hist, double* corr, uint8_t* isf64) {
    memcpy(hist, e->hist, sizeof(double) * 2 * (size_t)e->hist_len);
    memcpy(corr, e->corr, sizeof(double) * 4 * (size_t)e->corr_len);
    memcpy(isf64, e->hist_f64, (size_t)e->hist_len);
    return e->hist_len;
}
int ftlo_get_traj(const ftlo_env* e, float* out, int cap) {
    int n = e->traj_len < cap ? e->traj_len : cap;
    memcpy(out, e->traj, sizeof(float) * 2 * (size_t)n);
    return e->traj_len;
}

/* Batched stepping over independent envs (used by bench.py's cpu_baseline leg; OpenMP over envs). */
int ftlo_step_batch(ftlo_env** envs, int n, const double* actions, float* obs_num, float* lasers, double* target,
                    double* reward, uint8_t* done, uint8_t* status, int nthreads) {
    int L = n > 0 ? envs[0]->lasers_len : 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
#endif
    for (int i = 0; i < n; i++)
        ftlo_step(envs[i], actions[2 * i], actions[2 * i + 1], obs_num + (size_t)FTL_OBS_NUM * i, lasers + (size_t)L * i,
                  target + 2 * (size_t)i, reward + i, done + i, status + 3 * (size_t)i);
    (void)nthreads;
    return 0;
}

// ftl_device.hpp -- CDNA4 (gfx950) device code of the batched Game.step(): shared scalar math + the ray kernel.
//
// A step() is two launches on the same stream, split where the live state is smallest:
//
//   ftl_frames_group_kernel (ftl_frames_group.hpp) -- frames_per_step x Game.frame_step (follow_the_leader_continuous_env.py:947-1141),
//       the two LeaderPositionsTracker_v2 scans of use_sensors (classes.py:263-267, 285-286; sensors.py:243-327), the history snapshot
//       of the ray sensors (sensors.py:896-897) and _get_obs (1789-1810): G = 4 or 8 lanes per env, 64 / G envs per wavefront, robot r of
//       an env on lane r of its group, list-shaped work strided over the group.
//   ftl_rays_kernel (below) -- LeaderCorridor_Prev_lasers_v2.scan (sensors.py:883-962) and its relatives for every ray sensor: one
//       wavefront per env; obstacle sources culled into a small LDS table (near rects with their FACING edges, references into a float32
//       copy of the corridor ring, green caps); one segment per lane works out the arc of rays that can reach it, every sensor turns
//       the arc into (segment, ray) candidates, and the candidates of a chunk go through the reference's intersection test 64 at a
//       time; nearest squared distance per (ray, snapshot) by a 64-bit LDS atomic min; rows written sensor by sensor.
//   ftl_aux_kernel / ftl_tracker1_kernel (ftl_aux.hpp), ftl_gz_kernel (ftl_gazebo.hpp) -- the sensors and the tracker variants outside
//       the headline configs; launched only when a config has them.
// No MFMA: there is no dense contraction anywhere on this path.
//
// Numerics follow oracle/ftl_oracle.c operation by operation (same dtype flow, explicit fma only where numpy/BLAS
// fuse); the translation unit is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ftl.h"

#define FTL_WAVE 64
#define FTL_HMAX 12         // compile-time cap on max_prev_obs (the shipped training configs use 10)
#define FTL_MAX_RAYS 1024   // rays per env over all ray sensors (the host rejects more than 1023)
#ifndef FTL_WIDE_ARC
#define FTL_WIDE_ARC 3       // segments facing more candidate rays than this have their pairs written by the whole wavefront, one ray per lane
#endif
#define FTL_PAIR_CAP (64 * FTL_WIDE_ARC)     // entries of phase 3's candidate list (u16: lane of the segment << 10 | ray): one sensor's worth of a
                                             // chunk at most.  Kept small on purpose: 512 bytes of LDS more cost the ray kernel a wavefront per CU
                                             // and 6 % of its speed (FTL_DEBUG_LDS_PAD_RAYS)
#ifndef FTL_RAYS_WPE
#define FTL_RAYS_WPE 6      // 80 VGPRs (none spilled in the one-stream instantiations); 24 wavefronts per CU need <= 6.8 KB of LDS per env
#endif

// per-sensor record of the ray kernel's phase 3 (built on the host in ftl_create)
struct FtlRaySensor {
    int32_t count, rbase;          // rays; index of ray 0 among the rays of the sensors scanned in the same pass (before / after the tracker)
    float reach2;                  // (laser_length + 2)^2: segments whose closest approach is beyond it cannot be hit
    float inv_step, inv_count;     // count / (2 pi), 1 / count
    float off_u;                   // first_laser_angle_offset in units of the ray spacing
    float slack;                   // widening of the candidate arc, in ray spacings (>= 0.01 rad)
    uint32_t flags;                // bits 0-3: reacts to SEG_STATIC / DYNAMIC / CORRIDOR / GREEN; 16: explicit ray angles; 32: scanned after the tracker; 64: not a ray-kernel sensor (compas)
};
struct FtlDevParams {
    ftl_config cfg;
    int32_t n_envs, R, lasers_len, total_rays, hmax, lds_rays;
    int32_t fr_rec_off, fr_rec_stride, fr_pend_off, fr_env_off, fr_defer, fr_lds;   // frame kernel: LDS offsets of the frame records / pending items / slot -> env table
                                      // (+ the item counter), "the later frames' position searches wait for the end of the step", total dynamic LDS
    int32_t corr_lds_cap;             // corridor points the ray kernel stages in LDS (a power of two <= cfg.corr_cap; a longer window is read in place)
    int32_t pol_off[FTL_MAX_LASERS], pol_width, pol_h;   // fused sensorPrev output: column offset per sensor (-1: not part of it), row width, common history
    // per-env state (views into the caller-owned state buffer).  The small fields every step reads and writes -- env_int, fol_cs, rb_pos,
    // rb_dbl, snap_win, snap_rects, env_dbl, rb_int, in this order -- sit in ONE record of rec_stride bytes per env (a multiple of 128: the
    // group that owns an env touches whole cache lines of its own, whatever the slot -> env permutation did to its neighbours; as six arrays
    // they cost 1.6x the algorithmic HBM traffic in partial lines); these pointers address the field inside record 0: rec_field() below.
    // The ray kernel's inputs come first, so it reads the head of the record only.  The long fields (traj .. hist1) are [n_envs][per_env] arrays.
    float* rb_pos; double* rb_dbl; int32_t* rb_int; int32_t* env_int; double* env_dbl;
    int32_t rec_stride, _pad_rec;
    float* traj; double* hist; double* corr; int32_t* snap_rects; int32_t* snap_win;
    float* traj_bb;                   // [n_envs][traj_cap / FTL_TRAJ_BLOCK][4]: xmin, ymin, xmax, ymax of each block of trajectory points
    double* ep_stats;                 // [n_envs][FTL_N_METRICS]: metrics of the episodes that ended in this env slot (include/ftl.h)
    float* corr32;                    // [n_envs][corr_cap][4]: the corridor ring once more, in float32 -- what the ray kernel works with (sensors.py:672 casts
                                      // the edges to float32); written with the float64 ring, half the bytes to read per scan
    float* hist1;                     // [n_envs][hist1_cap][2]: position history of the v1 tracker (sensors.py:148-229)
    double* fol_cs;                   // [n_envs][2]: cos, sin of the follower's direction as the frame kernel leaves it (one sincos per lane
                                      // there serves 16 envs; here it would be one per ray)
    // ray directions relative to the follower's heading, host-computed with glibc: (cos, sin) of (first_laser_angle_offset + i * 360 / N)
    // -- or of ray_angles[i] -- in degrees, indexed by the ray's position over ALL ray sensors in config order
    double ray_rot[FTL_MAX_RAYS][2];
    FtlRaySensor ray_sens[FTL_MAX_LASERS];   // phase 3 of the ray kernel: what it needs of each sensor, packed (two scalar loads)
    uint32_t inv_nrect_dyn;           // ceil(65536 / (R - 1)): source index / objects per snapshot without an integer division
    int32_t miss_const;               // 1: a ray without a hit reads float32(laser_length) exactly for every sensor (checked on the host: the
                                      // value float64 |end - origin| lies within 4e-13 of laser_length, far from a float32 rounding boundary)
    // env regrouping (library-owned; null = envs stay bound to their wavefronts): slot -> env, cost class of the next step,
    // rank inside the block histogram, per-block key histograms
    int32_t* perm; uint8_t* keys; uint16_t* rank; int32_t* bh;
    ftl_scenarios scen;
};
// per-call arguments (passed by value in the kernarg segment)
struct FtlCall {
    const double* action; const int32_t* scen_idx; const uint8_t* mask;
    ftl_outputs out;
    uint32_t flags; int32_t mode;      // mode 0 = step, 1 = reset
    int32_t action_kind;               // FTL_ACTION_*: how `action` is encoded (ftl_step_encoded)
    int32_t win_base, win_count, win_stride;   // pool entries the auto-reset draws from and the step of its walk (ftl_set_reset_window)
    int32_t part, parts, epw;          // this launch covers the slot groups (epw consecutive slots = one frame-kernel wavefront)
                                       // part, part + parts, part + 2*parts, ... of the slot -> env permutation
};

// field `f0` (its address inside record 0) of env `env`
template <typename T>
__device__ __forceinline__ T* rec_field(T* f0, const FtlDevParams& P, size_t env) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(f0) + env * (size_t)P.rec_stride);
}

namespace ftl {

static constexpr double kDeg2Rad = 3.141592653589793 / 180.0;
static constexpr double kRad2Deg = 180.0 / 3.141592653589793;

// ---------------------------------------------------------------- cross-lane helpers
// ---------------------------------------------------------------- scalar math shared by all phases
// sin & cos of a float64 angle with |x| < ~1e5 rad (every angle on this path is below 15 rad): three-term Cody-Waite
// reduction by pi/2 carrying the rounding tail, then the fdlibm minimax kernels on [-pi/4, pi/4] (< 1 ulp).  The
// generic ocml sincos carries a Payne-Hanek path for huge arguments that costs registers and instructions here.
__device__ __forceinline__ void sincos_bounded(double x, double& s, double& c) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    double fn = rint(x * invpio2);
    int n = (int)fn;
    double t = x - fn * pio2_1;
    double w = fn * pio2_2;
    double r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    // __kernel_sin(y0, y1, 1) / __kernel_cos(y0, y1)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = y0 * y0;
    double v = z * y0;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    double ww = z * z;
    double rc = z * (C1 + z * (C2 + z * C3)) + (ww * ww) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    double wc = 1.0 - hz;
    double kc = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    int q = n & 3;
    double sv = (q & 1) ? kc : ks;
    double cv = (q & 1) ? ks : kc;
    s = (q & 2) ? -sv : sv;
    c = ((q + 1) & 2) ? -cv : cv;
}

// scipy distance.euclidean on two float32 points (snrm2: f32 differences, f64 accumulate, f32 result)
__device__ __forceinline__ double euclid_f32(float ax, float ay, float bx, float by) {
    float dx = ax - bx, dy = ay - by;
    return (double)(float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}
// ... with a float64 / int operand (dnrm2).  The reference accumulates in x87 extended precision; plain f64 differs
// from it by at most one ulp, and every use on this path is a threshold comparison (DESIGN.md "numerics").
__device__ __forceinline__ double euclid_f64(double ax, double ay, double bx, double by) {
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}
// Threshold tests on those distances.  sqrt is monotone, so away from the threshold the comparison can be made on the
// squared distance; inside a relative band around thr^2 (wider than every rounding involved) the exact form runs.
// Both give the same boolean as the reference expression -- only the instruction count differs.
__device__ __forceinline__ bool euclid_f32_le(float ax, float ay, float bx, float by, double thr) {   // euclid_f32(..) <= thr
    float dx = ax - bx, dy = ay - by;
    double x = (double)dx * (double)dx + (double)dy * (double)dy, t2 = thr * thr;
    if (x < t2 * (1.0 - 1e-6)) return true;
    if (x > t2 * (1.0 + 1e-6)) return false;
    return (double)(float)sqrt(x) <= thr;
}
__device__ __forceinline__ bool euclid_f64_lt(double ax, double ay, double bx, double by, double thr) { // euclid_f64(..) < thr
    double dx = ax - bx, dy = ay - by;
    double x = dx * dx + dy * dy, t2 = thr * thr;
    if (x < t2 * (1.0 - 1e-12)) return true;
    if (x > t2 * (1.0 + 1e-12)) return false;
    return sqrt(x) < thr;
}
__device__ __forceinline__ double angle_correction(double a) {   // misc.py:6-13
    if (a >= 360.0) return a - 360.0;
    if (a < 0.0) return 360.0 + a;
    return a;
}
__device__ __forceinline__ double angle_to_point(double cx, double cy, double tx, double ty) {   // misc.py:16-26
    double rx = tx - cx, ry = ty - cy, res;
    if (rx > 0.0) res = atan(ry / rx) * kRad2Deg;
    else if (rx < 0.0) res = atan(ry / rx) * kRad2Deg + 180.0;
    else res = 0.0;
    return angle_correction(res);
}
// pygame.transform.rotate bounding box (transform.c surf_rotate)
__device__ __forceinline__ void rotate_size(int w, int h, double angle_deg, int& nw, int& nh) {
    double a = (double)(float)angle_deg;
    double q = rint(a / 90.0);
    if (q * 90.0 == a) {                       // fmod(a, 90) == 0
        long long qi = (long long)(a / 90.0);
        if (qi & 1) { nw = h; nh = w; } else { nw = w; nh = h; }
        return;
    }
    double r = a * .01745329251994329, s, c;
    sincos_bounded(r, s, c);
    double cx = c * w, cy = c * h, sx = s * w, sy = s * h;
    double mx = fmax(fmax(fmax(fabs(cx + sy), fabs(cx - sy)), fabs(-cx + sy)), fabs(-cx - sy));
    double my = fmax(fmax(fmax(fabs(sx + cy), fabs(sx - cy)), fabs(-sx + cy)), fabs(-sx - cy));
    nw = (int)mx; nh = (int)my;
}

// ---------------------------------------------------------------- per-lane robot record
struct Robot {
    float px, py;
    double direction, speed, rot_speed, des_speed, des_rot_speed;
    int rot_dir, des_rot_dir;
    int rx, ry, rw, rh;
    // bears only: the way-point of the previous frame and the index into the way-point cycle (ENV:717-718)
    double tgt_x, tgt_y;
    int dyn_index;
};
struct Limits { double min_speed, max_speed, max_rot, max_dv, max_drot; int img_w, img_h; };

__device__ __forceinline__ Limits lane_limits(const ftl_config& c, int lane) {
    const ftl_robot_params& p = (lane == 0) ? c.leader : (lane == 1 ? c.follower : c.bear);
    Limits L;
    L.min_speed = p.min_speed; L.max_speed = p.max_speed; L.max_rot = p.max_rotation_speed;
    L.max_dv = p.max_speed_change; L.max_drot = p.max_rotation_speed_change; L.img_w = p.img_w; L.img_h = p.img_h;
    return L;
}

__device__ __forceinline__ void command_turn(Robot& r, const Limits& L, double des, int dir) {       // classes.py:109-117
    r.des_rot_speed = (des <= L.max_rot) ? des : L.max_rot;
    r.des_rot_dir = dir;
}
__device__ __forceinline__ void command_forward(Robot& r, const Limits& L, double s) {               // classes.py:119-127
    if (s > L.max_speed) s = L.max_speed;
    if (s < L.min_speed) s = L.min_speed;
    r.des_speed = s;
}
// classes.py:165-182 (controller 134-163 inlined); `active` lanes commit, the rest keep their state
__device__ __forceinline__ void robot_move(Robot& r, const Limits& L, bool active, double& dir_sin, double& dir_cos) {
    // _turn_processing
    int rot_dir = r.rot_dir;
    if (rot_dir == 0) rot_dir = r.des_rot_dir;
    double change;
    if (rot_dir == r.des_rot_dir) {
        double needed = fabs(r.rot_speed - r.des_rot_speed);
        change = (needed <= L.max_drot) ? needed : L.max_drot;
        if (r.des_rot_speed < r.rot_speed) change = -1 * change;
    } else {
        double needed = fabs(r.des_rot_speed + r.rot_speed);
        change = -((needed <= L.max_drot) ? needed : L.max_drot);
    }
    double nr = r.rot_speed + change;
    if (nr < 0) rot_dir = -1 * rot_dir;
    double rot_speed = fabs(nr);
    // _speed_processing
    double needed = fabs(r.speed - r.des_speed);
    double dv = (L.max_dv <= needed) ? L.max_dv : needed;
    if (r.speed > r.des_speed) dv = -1 * dv;
    double speed = r.speed + dv;

    double direction = r.direction;
    int rx = r.rx, ry = r.ry, rw = r.rw, rh = r.rh;
    bool turning = active && (rot_speed != 0);
    if (turning) direction = angle_correction(direction + rot_dir * rot_speed);
    double s, c;
    sincos_bounded(direction * kDeg2Rad, s, c);
    dir_sin = s; dir_cos = c;          // sin / cos of the direction the robot leaves the frame with
    if (turning) {
        // New hitbox size = pygame.transform.rotate(image, -direction) (classes.py:173-175).  The bounding box only
        // needs int(|cos|w+|sin|h): take it from the sin/cos of the movement (the rotate call rounds the angle to
        // float32, which moves the value by < (w+h)*4e-7) unless that value sits next to an integer or the float32
        // angle is a multiple of 90 degrees -- then rotate_size() evaluates the reference expression itself.
        double a32 = (double)(float)(-direction);
        double vw = fabs(c) * L.img_w + fabs(s) * L.img_h, vh = fabs(s) * L.img_w + fabs(c) * L.img_h;
        double tol = (double)(L.img_w + L.img_h) * 1e-6;
        double fw = vw - floor(vw), fh = vh - floor(vh);
        // multiple of 90?  k = rint(a32 / 90) from a multiplication (an f64 division costs ~35 instructions): if a32 IS a
        // multiple the product is k(1 + eps) and rounds to k, if it is not no integer k satisfies k * 90 == a32
        bool exact = (rint(a32 * (1.0 / 90.0)) * 90.0 == a32) || fw < tol || fw > 1.0 - tol || fh < tol || fh > 1.0 - tol;
        int nw = (int)vw, nh = (int)vh;
        if (exact) rotate_size(L.img_w, L.img_h, -direction, nw, nh);
        int cx = rx + (rw >> 1), cy = ry + (rh >> 1);
        rw = nw; rh = nh; rx = cx - (nw >> 1); ry = cy - (nh >> 1);
    }
    float mx = (float)(c * speed), my = (float)(s * speed);
    float px = r.px + mx, py = r.py + my;
    double dx = (double)px - (double)(rx + (rw >> 1));
    double dy = (double)py - (double)(ry + (rh >> 1));
    rx += (int)dx; ry += (int)dy;       // Rect.move_ip truncates toward zero
    if (active) {
        r.rot_dir = rot_dir; r.rot_speed = rot_speed; r.speed = speed; r.direction = direction;
        r.px = px; r.py = py; r.rx = rx; r.ry = ry; r.rw = rw; r.rh = rh;
    }
}
// int(angle_to_point(c, t)) -- all that classes.py:187 keeps of the angle.  -DFTL_FAST_ATAN settles the whole degree by a float32
// estimate of the angle (the float32 roundings of the exact differences, the division and the polynomial together stay below 3e-4
// degrees) unless the estimate lies within 2e-3 degrees of a whole degree, where the reference expression itself decides: parity-green,
// 100 float64 operations fewer per frame -- and 4 % SLOWER on the frame kernel (180 against 173 us on config B, interleaved runs of round
// 3; round 2 saw the same with its own version): the wavefront keeps the exact path for the 0.4 % of the lanes that need it, and the extra
// branches and live values cost more than the arithmetic saved.  Off.
__device__ __forceinline__ int angle_to_point_int(double cx, double cy, double tx, double ty) {
#ifndef FTL_FAST_ATAN
    return (int)angle_to_point(cx, cy, tx, ty);
#else
    const double rx = tx - cx, ry = ty - cy;
    if (rx == 0.0) return 0;                                 // misc.py:24: res = 0
    const float fx = (float)rx, fy = (float)ry;
    const float ax = fabsf(fx), ay = fabsf(fy);
    const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
    const float a = lo / hi;                                 // hi > 0: rx != 0
    const float t = a * a;
    // atan(a) on [0, 1]: odd minimax polynomial of degree 11 (|error| < 2e-7 rad)
    float p = __builtin_fmaf(t, -0.0117212f, 0.05265332f);
    p = __builtin_fmaf(t, p, -0.11643287f); p = __builtin_fmaf(t, p, 0.19354346f); p = __builtin_fmaf(t, p, -0.33262347f); p = __builtin_fmaf(t, p, 0.99997726f);
    float d = a * p * 57.29577951308232f;                    // degrees in [0, 45]
    d = ay > ax ? 90.0f - d : d;                             // [0, 90]
    d = fx < 0.0f ? 180.0f - d : d;                          // [0, 180]
    d = fy < 0.0f ? 360.0f - d : d;                          // [0, 360]
    const float fl = floorf(d), fr = d - fl;
    if (fr > 2.0e-3f && fr < 1.0f - 2.0e-3f) return (int)fl; // (360 itself has fr = 0 and takes the exact path)
    return (int)angle_to_point(cx, cy, tx, ty);
#endif
}

// classes.py:184-215: steering toward (tx,ty); has_speed false means "speed=None" (use the distance)
__device__ __forceinline__ void steer_to_point(Robot& r, const Limits& L, double tx, double ty, bool has_speed, double speed) {
    double new_speed = has_speed ? speed : euclid_f64((double)r.px, (double)r.py, tx, ty);
    int desirable = angle_to_point_int((double)r.px, (double)r.py, tx, ty);
    int cur = (int)r.direction;
    int delta, dir;
    if (desirable - cur > 0) {
        if (desirable - cur > 180) { delta = cur + (360 - desirable); dir = -1; }
        else { delta = desirable - cur; dir = 1; }
    } else {
        if (cur - desirable > 180) { dir = 1; delta = (360 - cur) + desirable; }
        else { dir = -1; delta = cur - desirable; }
    }
    command_turn(r, L, (double)delta, dir);
    command_forward(r, L, new_speed);
}

__device__ __forceinline__ bool rects_collide(int ax, int ay, int aw, int ah, int bx, int by, int bw, int bh) {
    if (aw == 0 || ah == 0 || bw == 0 || bh == 0) return false;
    return ax < bx + bw && ay < by + bh && ax + aw > bx && ay + ah > by;
}

// ---- tracker rings (LeaderPositionsTracker_v2 state, sensors.py:156-176): absolute point index -> ring slot --------
__device__ __forceinline__ double* hist_slot(const FtlDevParams& P, int env, int abs_idx) {
    return P.hist + ((size_t)env * P.cfg.corr_cap + (abs_idx & (P.cfg.corr_cap - 1))) * 2;
}
__device__ __forceinline__ double* corr_slot(const FtlDevParams& P, int env, int abs_idx) {
    return P.corr + ((size_t)env * P.cfg.corr_cap + (abs_idx & (P.cfg.corr_cap - 1))) * 4;
}
__device__ __forceinline__ float4* corr32_slot(const FtlDevParams& P, int env, int abs_idx) {
    return reinterpret_cast<float4*>(P.corr32) + (size_t)env * P.cfg.corr_cap + (abs_idx & (P.cfg.corr_cap - 1));
}
// ---- LeaderCorridor_Prev_lasers_v2.scan (sensors.py:883-962): the rays ----------------------------------------------
// Segment table entry classes (sensors.py:644-660): which sensors see an entry is decided per class
enum { SEG_STATIC = 0, SEG_DYNAMIC = 1, SEG_CORRIDOR = 2, SEG_GREEN = 3, SEG_CLASSES = 4 };

// One obstacle segment A->B (float32, as stored by np.array(..., dtype=np.float32), sensors.py:672) against the ray
// origin C (float32) -> end E (float64).  Returns true on intersection and the SQUARED distance of the intersection
// point from the origin (sqrt is monotone: min over distances == sqrt of min over squared distances, so the sqrt of
// sensors.py:920-921 is taken once per output element instead of once per hit).
__device__ __forceinline__ bool hit_segment(float cx, float cy, double ex, double ey, float ex32, float ey32, float4 sg, double& d2) {
    const float ax = sg.x, ay = sg.y, bx = sg.z, by = sg.w;
    // ccw / intersect with the dtype flow of sensors.py:608-614:
    //   t1 = ccw(A,C,D), t2 = ccw(B,C,D), t4 = ccw(A,B,D) multiply an f64 difference by an f32-rounded difference and
    //   compare in f64; t3 = ccw(A,B,C) is evaluated entirely in float32.
    float cax = cx - ax, cay = cy - ay, cbx = cx - bx, cby = cy - by, bax = bx - ax, bay = by - ay;
    bool t3 = cay * bax > bay * cax;
    // Fast filter: the same three orientation values in float32.  With every coordinate below 2^11 and every
    // difference below ~2^10 the float32 estimate is within 0.3 of the real value (error budget in DESIGN.md), so an
    // estimate beyond +-0.5 fixes the sign the float64 expression of the reference yields; otherwise it is re-evaluated
    // exactly as the reference does.
    float fdax = ex32 - ax, fday = ey32 - ay, fdbx = ex32 - bx, fdby = ey32 - by;
    float s1 = fday * cax - cay * fdax, s2 = fdby * cbx - cby * fdbx, s4 = fday * bax - bay * fdax;
    bool t1 = s1 > 0.0f, t2 = s2 > 0.0f, t4 = s4 > 0.0f;
    if (!(fabsf(s1) > 0.5f && fabsf(s2) > 0.5f && fabsf(s4) > 0.5f)) {
        double day = ey - (double)ay, dax = ex - (double)ax, dby = ey - (double)by, dbx = ex - (double)bx;
        t1 = day * (double)cax > (double)cay * dax;
        t2 = dby * (double)cbx > (double)cby * dbx;
        t4 = day * (double)bax > (double)bay * dax;
    }
    if (!((t1 != t2) && (t3 != t4))) return false;
    // seg_intersect (sensors.py:626-640)
    double rbx = ex - (double)cx, rby = ey - (double)cy;
    float dpx = ax - cx, dpy = ay - cy;
    float dapx = -bay, dapy = bax;
    double denom = (double)dapx * rbx + (double)dapy * rby;
    float num = dapx * dpx + dapy * dpy;
    double t = (double)num / denom;
    double x = t * rbx + (double)cx, y = t * rby + (double)cy;
    double qx = x - (double)cx, qy = y - (double)cy;
    d2 = qx * qx + qy * qy;
    return true;
}

}  // namespace ftl

// ---------------------------------------------------------------- kernel 2: the ray casts
// One wavefront per env.  Every phase flattens its work items into one index space so that the 64 lanes stay busy.
// Phase 1 (one SOURCE per lane): every obstacle any sensor of this env could see -- static rects, the leader / bear
//   rects of the last H snapshots, the corridor polyline points and green-zone caps of those snapshots -- is tested
//   against the sensors' reach around the follower; survivors are expanded to segments and compacted into a per-class
//   table in LDS (segment f32x4 + bit mask of the snapshots that contain it).  A segment wholly outside the reach box
//   cannot intersect any ray, so dropping it is exact.
// Phase 2 (one RAY per lane): ray ends of all sensors (sensors.py:888-891) and the per-(ray, snapshot) minima.
// Phase 3 (one SEGMENT of the table per lane): the lane works out once what its segment subtends at the follower; every sensor of
//   the pass then maps that arc -- widened by a slack that dwarfs every rounding error -- to its own ray indices (typically 0-3 of
//   the 12 / 24 rays) and appends (segment lane, ray) pairs to a candidate list in LDS.  The list is tested densely, 64 pairs per
//   pass, with the reference's intersection test; a hit is folded into the nearest squared distance of (ray, snapshot) with a
//   64-bit LDS atomic min (non-negative doubles order like their bit patterns).
// Phase 4 (one RAY per lane): H minima -> the H output rows of the ray.
// HM = compile-time number of history accumulators: 5, 8, 10 (the shipped training configs) or FTL_HMAX = 12
// atan2 for the candidate-ray arc of phase 3 only: |error| <= 2e-5 rad (Abramowitz-Stegun 4.4.47 polynomial on [0,1] +
// octant folding), two orders of magnitude inside the arc slack (>= 0.01 rad) that absorbs it.  Never used for a value
// that reaches an output.
__device__ __forceinline__ float arc_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
    const float a = hi > 0.0f ? __fdividef(lo, hi) : 0.0f;
    const float t = a * a;
    // (explicit fma: the translation unit is compiled with -ffp-contract=off for the code that follows the reference operation by
    //  operation; this value only selects candidate rays, see above)
    float r = a * __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, 0.0208351f, -0.0851330f), 0.1801410f), -0.3302995f), 0.9998660f);
    r = ay > ax ? 1.5707963267948966f - r : r;
    r = x < 0.0f ? 3.141592653589793f - r : r;
    return y < 0.0f ? -r : r;
}

#ifdef FTL_PROFILE_RAYS       // diagnostic build only (profiles/tools/path_counts.py): cycles per phase of the ray kernel
__device__ unsigned long long g_rcyc[16];
__shared__ unsigned long long s_rcyc[16];
#define FTL_RTIC(slot) do { unsigned long long _t = __builtin_readcyclecounter(); if (threadIdx.x == 0) s_rcyc[slot] += _t - _rprev; _rprev = _t; } while (0)
#define FTL_RTIC_INIT unsigned long long _rprev = __builtin_readcyclecounter()
#else
#define FTL_RTIC(slot) do { } while (0)
#define FTL_RTIC_INIT do { } while (0)
#endif

// EXPL = the config uses one of the rarer sensor features -- rays at explicit angles (LeaderCorridor_lasers) or the pad_sectors
// row layout: compiled apart so that the common kernels carry none of that code (it cost 3 % even when never executed)
// The per-sensor loops run to the compile-time bound FTL_MAX_LASERS with the count as a guard: unrolled, the config fields they
// read become loop-invariant scalar loads that the compiler hoists out of the chunk loops (the ray kernel got 5 % faster).
// (LeaderCorridor_lasers_compas entries are cast by ftl_aux_kernel in float64; configs that have one run the EXPL instantiations)
#define FTL_FOR_LASERS(k) _Pragma("unroll") for (int k = 0; k < FTL_MAX_LASERS; k++) if (k < c.n_lasers && !(EXPL && c.lasers[k].compas))

// SPLIT = the launch covers one of the interleaved halves of the slot groups (two-stream mode) instead of all envs
// CAPPED = the LDS copy of the corridor ring is smaller than the ring itself: a window that does not fit is read in place (below)
template <int HM, bool EXPL = false, bool SPLIT = false, bool CAPPED = false>
__global__ void __launch_bounds__(FTL_WAVE, FTL_RAYS_WPE) ftl_rays_kernel(const FtlDevParams* __restrict__ Pp, const FtlCall C) {
    extern __shared__ __align__(16) unsigned char lds[];
    using namespace ftl;
    const FtlDevParams& P = *Pp;
    const ftl_config& c = P.cfg;
    // One launch for all envs: block b serves env b (any order will do).  A launch for one of `parts` interleaved halves must
    // serve exactly the envs its frame kernel served: slot (b % epw) of this launch's (b / epw)-th group of the slot -> env map.
    int env = blockIdx.x;
    if (SPLIT) {
        const int gslot = ((int)(blockIdx.x / C.epw) * C.parts + C.part) * C.epw + (int)(blockIdx.x % C.epw);
        if (gslot >= P.n_envs) return;
        env = P.perm ? P.perm[gslot] : gslot;
    }
    if (env >= P.n_envs) return;
    if (C.mode == 1 && C.mask && !C.mask[env]) return;
    const int lane = threadIdx.x;
#ifdef FTL_PROFILE_RAYS
    if (lane < 16) s_rcyc[lane] = 0;
    __syncthreads();
#endif
    FTL_RTIC_INIT;
    const int hmax = P.hmax;
    const int nrect_dyn = P.R - 1;      // leader + bears per snapshot
    const int cmask = P.corr_lds_cap - 1;   // LDS copy of the corridor ring: a power of two of slots, a window of consecutive points maps 1:1
    // LDS: f32 corridor ring | near rects (int4 + mask; static class first region, dynamic class second) | corridor
    //      segment references (u32) | green caps (f32x4 + mask) | class counters | ray ends | minima.
    // Rect edges and corridor segments are expanded on the fly in phase 3, so the table stays small (occupancy).
    const int cap_rs = c.n_static + hmax, cap_rd = hmax * (P.R - 2 > 0 ? P.R - 2 : 0) + 1;
    const int cap_cr = 2 * P.corr_lds_cap, cap_gr = 2 * hmax;
    float4* s_corr = reinterpret_cast<float4*>(lds);
    int4* s_rect = reinterpret_cast<int4*>(s_corr + P.corr_lds_cap);             // [cap_rs + cap_rd]
    float4* s_green = reinterpret_cast<float4*>(s_rect + cap_rs + cap_rd);       // [cap_gr]
    unsigned* s_rmask = reinterpret_cast<unsigned*>(s_green + cap_gr);           // [cap_rs + cap_rd]
    unsigned* s_cref = s_rmask + cap_rs + cap_rd;                                // [cap_cr]: p & cmask | side << 15 | mask << 16
    unsigned* s_gmask = s_cref + cap_cr;                                         // [cap_gr]
    int* s_cnt = reinterpret_cast<int*>(s_gmask + cap_gr);                       // [SEG_CLASSES] table entries per class, then [2] visible edges of the rect classes
    int* s_ecnt = s_cnt + SEG_CLASSES;
    unsigned short* s_edge = reinterpret_cast<unsigned short*>(s_cnt + 8);       // [4 * (cap_rs + cap_rd)]: rect slot << 2 | edge, the edges facing the follower
    const int n_u32 = (cap_rs + cap_rd) + cap_cr + cap_gr + 8 + 2 * (cap_rs + cap_rd);   // words since the last 16-byte aligned array
    double2* s_ray = reinterpret_cast<double2*>(s_rmask + ((n_u32 + 3) & ~3));     // 16-byte aligned [total_rays]
    // The minima are kept as the float32 the output array holds (sensors.py:896-901: float32(min of the float64 distances) -- rounding is
    // monotone, so the minimum of the rounded values is the rounded minimum; non-negative floats order like their bit patterns).  Half
    // the LDS of float64 minima: with the frame kernel of another stream on the same CU, LDS is what limits how many of these wavefronts fit.
    unsigned* s_best = reinterpret_cast<unsigned*>(s_ray + P.total_rays);                                // [HM][total_rays]
    float* s_miss = reinterpret_cast<float*>(s_best + (size_t)P.total_rays * HM);                        // [total_rays] float32(|ray end - origin|)
    unsigned short* s_pair = reinterpret_cast<unsigned short*>(s_miss + P.total_rays);                  // [FTL_PAIR_CAP] candidate list of phase 3
    const unsigned kInfBits = 0x7f800000u;                                                            // +inf: "no hit"

    // Round trip 1: everything that is addressed by the env index alone is requested at once -- the scalars, every ring slot
    // of the snapshot windows (one word per lane) and of the snapshot rects (one rect per lane); which slots are valid is
    // sorted out after they have arrived.  Round trip 2 (below): corridor points and the scenario's static rects.
    const int* ei = rec_field(P.env_int, P, env);
    const int* swp = rec_field(P.snap_win, P, env);
    const int4* srp = reinterpret_cast<const int4*>(rec_field(P.snap_rects, P, env));
    const int swv = lane < hmax * 4 ? swp[lane] : 0;
    // per-env scalars come through VECTOR loads (one word per lane) and are made wave-uniform with readlane: streaming
    // them through the scalar cache (s_load) costs several microseconds per miss under this kernel's load
    const int eiv = lane < FTL_EI_COUNT ? ei[lane] : 0;
    const int fpv = lane < 2 ? __float_as_int(rec_field(P.rb_pos, P, env)[2 * 1 + lane]) : 0;                 // robot 1 = the follower
    const int fdv = lane < 2 ? reinterpret_cast<const int*>(rec_field(P.rb_dbl, P, env) + 1 * FTL_RD_COUNT + FTL_RD_DIRECTION)[lane] : 0;
    const int fcv = lane < 4 ? reinterpret_cast<const int*>(rec_field(P.fol_cs, P, env))[lane] : 0;
    const int scen = __builtin_amdgcn_readlane(eiv, FTL_EI_SCEN), snap_count = __builtin_amdgcn_readlane(eiv, FTL_EI_SNAP_COUNT);
    const int scan_ok = __builtin_amdgcn_readlane(eiv, FTL_EI_SCAN_OK), snap_head = __builtin_amdgcn_readlane(eiv, FTL_EI_SNAP_HEAD);
    const int newest = (snap_head == 0 ? hmax : snap_head) - 1;   // ring slot of the newest snapshot
    const float cx = __int_as_float(__builtin_amdgcn_readlane(fpv, 0)), cy = __int_as_float(__builtin_amdgcn_readlane(fpv, 1));
    const double fdir = __hiloint2double(__builtin_amdgcn_readlane(fdv, 1), __builtin_amdgcn_readlane(fdv, 0));
    const double fcd = __hiloint2double(__builtin_amdgcn_readlane(fcv, 1), __builtin_amdgcn_readlane(fcv, 0));     // cos / sin of fdir (frame kernel)
    const double fsd = __hiloint2double(__builtin_amdgcn_readlane(fcv, 3), __builtin_amdgcn_readlane(fcv, 2));
    float* out_base = C.out.lasers + (size_t)env * P.lasers_len;
    const int nsnap = snap_count < hmax ? snap_count : hmax;       // valid snapshots, newest = snap_count-1
    const unsigned all_snaps = (1u << nsnap) - 1u;                  // bit a = age a (nsnap <= hmax <= FTL_HMAX = 12)
    const int4* stp = reinterpret_cast<const int4*>(P.scen.static_rects) + (size_t)scen * c.n_static;

#pragma nounroll
    for (int which = 0; which < 2; which++) {
        // The lane index is made opaque once per pass (it shadows the kernel's `lane` from here on): left visible, every per-lane address of
        // the pass -- LDS slots, the lane's rects -- is formed once before this loop and kept alive across it, and at the 80 registers that
        // six wavefronts per SIMD allow the compiler paid for that with a register pair in scratch (512 B of scratch writes per env-step in
        // the PMC write counter, and a reload in the middle of the pass).
        int lane_v = (int)threadIdx.x;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        int n_sens = 0; float lmax = 0.0f;
        FTL_FOR_LASERS(k) if (c.lasers[k].after_tracker == which) { n_sens++; lmax = fmaxf(lmax, (float)c.lasers[k].length); }
        if (n_sens == 0) continue;
        // this lane's snapshot rect and static rect: requested here, with the pass's other loads, and dead after phase 1 (held across
        // the passes they cost the test loop of phase 3 eight registers)
        // (with room for both in one wavefront -- 37 + 10 lanes on the bench workload -- the snapshot rects sit in the lanes behind the
        //  static ones and phase 1 pushes both with ONE pass of its rect code)
        const int ndyn = hmax * nrect_dyn;
        const bool merged = c.n_static + ndyn <= FTL_WAVE;
        const int dl = merged ? lane - c.n_static : lane;                   // index of this lane's snapshot rect
        const int4 dynq = (dl >= 0 && dl < ndyn) ? srp[dl] : make_int4(0, 0, 0, 0);
        const int4 stq = lane < c.n_static ? stp[lane] : make_int4(0, 0, 0, 0);
        if (!((scan_ok >> which) & 1)) {       // sensors.py:893/962: the reference raises UnboundLocalError here
            for (int k = 0; k < c.n_lasers; k++) if (c.lasers[k].after_tracker == which && !(EXPL && c.lasers[k].compas)) {
                const int Wd = c.lasers[k].count * (c.lasers[k].pad_sectors ? 4 : 1);
                for (int i = lane; i < c.lasers[k].history * Wd; i += FTL_WAVE) {
                    out_base[c.lasers[k].out_offset + i] = (float)c.lasers[k].length;
                    if (C.out.policy_obs && P.pol_off[k] >= 0)       // clip(length / length, 0, 1)
                        C.out.policy_obs[(size_t)env * P.pol_h * P.pol_width + (i / Wd) * P.pol_width + P.pol_off[k] + (i % Wd)] = 1.0f;
                }
            }
            continue;
        }
        // corridor windows of the valid snapshots as this group of sensors saw them; age a = 0 newest
        int win_lo[HM], win_hi[HM];
        int umin = 0x7fffffff, umax = 0;
#pragma unroll
        for (int a = 0; a < HM; a++) {
            win_lo[a] = 0; win_hi[a] = 0;
            if (a < nsnap) {
                int slot = newest - a; slot += slot < 0 ? hmax : 0;
                win_lo[a] = __builtin_amdgcn_readlane(swv, 4 * slot + 2 * which); win_hi[a] = __builtin_amdgcn_readlane(swv, 4 * slot + 2 * which + 1);   // wave-uniform -> SGPRs
                umin = min(umin, win_lo[a]); umax = max(umax, win_hi[a]);
            }
        }
        FTL_RTIC(7);
#if defined(FTL_RAYS_STOP) && FTL_RAYS_STOP == 0      // diagnostic builds (profiles/tools/rays_phase_valu.sh): stop after a phase
        continue;
#endif
        __syncthreads();
        if (lane < SEG_CLASSES + 2) s_cnt[lane] = 0;
        // The windows of the snapshots span a few dozen corridor points; the tracker's ring is sized for the worst case (a leader that
        // crawls under a speed regime: 256 points), and LDS of that size would cost this kernel a third of its wavefronts.  The LDS copy
        // holds P.corr_lds_cap points; a longer span (rare) is not staged -- phase 3 then reads its corridor segments from the ring in
        // global memory, uncompacted (same values, same results).
        const bool staged = !CAPPED || umax - umin <= P.corr_lds_cap;
        auto corr_f32 = [&](int p) { return *corr32_slot(P, env, p); };
        if (staged) for (int p = umin + lane; p < umax; p += FTL_WAVE) s_corr[p & cmask] = corr_f32(p);
        __syncthreads();
        FTL_RTIC(0);
#if defined(FTL_RAYS_STOP) && FTL_RAYS_STOP == 1
        continue;
#endif

        // ---- phase 1: culled, compacted segment table; sources flattened: statics | snapshot rects | corridor points | caps
        const float reach = lmax + 2.0f;
        const float bx0 = cx - reach, bx1 = cx + reach, by0 = cy - reach, by1 = cy + reach;
        // Only the edges that FACE the follower go to the work list.  A ray that reaches a back-facing edge of an axis-aligned rect
        // has entered the rect through a facing edge at a smaller distance, and the reference's own test detects that entry unless the
        // ray passes within rounding (~1e-13 px: the float32 orientation test of an axis-aligned edge is an exact sign test, the
        // others are float64) of a corner -- so the minimum over the facing edges IS the reference's minimum over all four.  A
        // follower inside or on the rect keeps all four.
        auto push_rect = [&](int cls, int4 q, unsigned sm) {
            if (sm == 0u) return;
            const float l = (float)q.x, t = (float)q.y, r = (float)(q.x + q.z), b = (float)(q.y + q.w);
            if (r < bx0 || l > bx1 || b < by0 || t > by1) return;
            const int slot = (cls == SEG_STATIC ? 0 : cap_rs) + atomicAdd(&s_cnt[cls], 1);
            s_rect[slot] = q; s_rmask[slot] = sm;
            unsigned em = (cy > b ? 1u : 0u) | (cx > r ? 2u : 0u) | (cy < t ? 4u : 0u) | (cx < l ? 8u : 0u);      // edge order of sensors.py:668-671
            if (em == 0u) em = 15u;
            int at = (cls == SEG_STATIC ? 0 : 4 * cap_rs) + atomicAdd(&s_ecnt[cls], __popc(em));
            while (em) { const int e = __ffs(em) - 1; em &= em - 1; s_edge[at++] = (unsigned short)((slot << 2) | e); }
        };
        auto push_corr = [&](int p, int side, float ax, float ay, float bx, float by, unsigned sm) {
            if (fmaxf(ax, bx) < bx0 || fminf(ax, bx) > bx1 || fmaxf(ay, by) < by0 || fminf(ay, by) > by1) return;
            s_cref[atomicAdd(&s_cnt[SEG_CORRIDOR], 1)] = (unsigned)(p & cmask) | ((unsigned)side << 15) | (sm << 16);
        };
        auto push_green = [&](float ax, float ay, float bx, float by, unsigned sm) {
            if (fmaxf(ax, bx) < bx0 || fminf(ax, bx) > bx1 || fmaxf(ay, by) < by0 || fminf(ay, by) > by1) return;
            int at = atomicAdd(&s_cnt[SEG_GREEN], 1);
            s_green[at] = make_float4(ax, ay, bx, by); s_gmask[at] = sm;
        };
        {
            // snapshot rects: index = ring slot * nrect_dyn + object; the leader (object 0) is a static-class object (it sits in
            // game_object_list), bears are the dynamic class.  (1..5 objects per snapshot: a reciprocal instead of a division)
            const int dls = dl < 0 ? 0 : dl;
            const int slot = (int)(((unsigned)dls * P.inv_nrect_dyn) >> 16), o = dls - slot * nrect_dyn;
            int a = newest - slot; a += a < 0 ? hmax : 0;
            const unsigned sm_dyn = (dl >= 0 && dl < ndyn && a < nsnap) ? 1u << a : 0u;
            if (merged) {
                const bool is_st = lane < c.n_static;                      // static rects are identical in every snapshot
                push_rect((is_st || o == 0) ? SEG_STATIC : SEG_DYNAMIC, is_st ? stq : dynq, is_st ? all_snaps : sm_dyn);
            } else {
                push_rect(SEG_STATIC, stq, lane < c.n_static ? all_snaps : 0u);
                for (int w = lane + FTL_WAVE; w < c.n_static; w += FTL_WAVE) push_rect(SEG_STATIC, stp[w], all_snaps);
                push_rect(o == 0 ? SEG_STATIC : SEG_DYNAMIC, dynq, sm_dyn);
            }
            // corridor polylines: segment p -> p+1 belongs to every snapshot whose window holds both points
            if (staged) for (int p = umin + lane; p < umax - 1; p += FTL_WAVE) {
                unsigned sm = 0;
#pragma unroll
                for (int a = 0; a < HM; a++) if (a < nsnap && win_lo[a] <= p && p + 1 < win_hi[a]) sm |= 1u << a;
                if (sm) {
                    float4 u = s_corr[p & cmask], v = s_corr[(p + 1) & cmask];
                    push_corr(p, 0, u.x, u.y, v.x, v.y, sm);      // right border
                    push_corr(p, 1, u.z, u.w, v.z, v.w, sm);      // left border
                }
            }
            else if (lane == 0) s_cnt[SEG_CORRIDOR] = 2 * max(umax - umin - 1, 0);      // every segment of the span, culled where it is fetched
            if (lane < nsnap) {     // green-zone end caps of snapshot age `lane` (sensors.py:648-650)
                int lo = 0, hi = 0;
#pragma unroll
                for (int j = 0; j < HM; j++) if (j == lane) { lo = win_lo[j]; hi = win_hi[j]; }
                float4 u = staged ? s_corr[lo & cmask] : corr_f32(lo), v = staged ? s_corr[(hi - 1) & cmask] : corr_f32(hi - 1);
                push_green(u.x, u.y, u.z, u.w, 1u << lane);
                push_green(v.x, v.y, v.z, v.w, 1u << lane);
            }
        }
        FTL_RTIC(1);
#if defined(FTL_RAYS_STOP) && FTL_RAYS_STOP == 2
        continue;
#endif
        // ---- phase 2: ray ends + accumulators; the rays of ALL sensors of this group share one index space (a sensor
        // of 12 rays alone would leave 52 lanes idle through the f64 sin/cos)
        {
            int n_rays = 0;
            FTL_FOR_LASERS(k) if (c.lasers[k].after_tracker == which) n_rays += c.lasers[k].count;
            for (int g = lane; g < n_rays; g += FTL_WAVE) {
                int i = g, gi = 0, gb = 0; double len = 0; bool found = false;
#pragma unroll
                for (int k = 0; k < FTL_MAX_LASERS; k++) if (k < c.n_lasers) {       // (every sensor: the rotation table is indexed over all of them)
                    const int N = c.lasers[k].count;
                    if (c.lasers[k].after_tracker == which && !(EXPL && c.lasers[k].compas)) {
                        if (!found && i < N) { found = true; len = c.lasers[k].length; gi = gb + i; }
                        if (!found) i -= N;
                    }
                    gb += N;
                }
                // direction of ray i = heading + offset_i (sensors.py:888-891, 609-632): cos / sin by angle addition from the heading's
                // (frame kernel) and the offset's (host) -- within 2-3 ulp of the reference's cos(radians(heading + offset_i)), like the
                // device's own sincos
                const double2 rot = reinterpret_cast<const double2*>(&P.ray_rot[0][0])[gi];
                const double co = fcd * rot.x - fsd * rot.y, s = fsd * rot.x + fcd * rot.y;
                const double ex = (double)cx + co * len, ey = (double)cy + s * len;
                s_ray[g] = make_double2(ex, ey);
                if (P.miss_const) s_miss[g] = (float)len;                            // np.linalg.norm(end - position), sensors.py:925-930
                else { const double qx0 = ex - (double)cx, qy0 = ey - (double)cy; s_miss[g] = (float)sqrt(__builtin_fma(qy0, qy0, qx0 * qx0)); }
#pragma unroll
                for (int j = 0; j < HM; j++) s_best[j * P.total_rays + g] = kInfBits;       // [age][ray]: the lanes of a row read / write consecutive words
            }
        }
        __syncthreads();
        FTL_RTIC(2);
#if defined(FTL_RAYS_STOP) && FTL_RAYS_STOP == 3
        continue;
#endif

        // ---- phase 3: segments x the sensors that react to them x candidate rays ----------------------------------------------
        {
            // One SEGMENT of the table per lane.  What a segment subtends at the follower (the two end-point angles, the closest
            // approach) is the same for every sensor, so it is worked out once; the sensors of the group then take turns, each mapping
            // the arc to its own ray indices (parameters: one packed, host-built record per sensor, wave-uniform) and testing the rays
            // inside it.
            int cls_any = 0;
#pragma nounroll
            for (int k = 0; k < c.n_lasers; k++) { const unsigned fl = P.ray_sens[k].flags; if ((int)((fl >> 5) & 1u) == which && !(fl & 64u)) cls_any |= (int)(fl & 15u); }
            int cq[SEG_CLASSES];
#pragma unroll
            for (int q = 0; q < SEG_CLASSES; q++) cq[q] = ((cls_any >> q) & 1) ? (q < 2 ? s_ecnt[q] : s_cnt[q]) : 0;
            const int n_items = cq[0] + cq[1] + cq[2] + cq[3];
            // table entry (class mq, index m) -> segment + mask of the snapshots that contain it
            auto fetch = [&](int mq, int m, float4& sg, unsigned& sm) {
                if (mq < 2) {                                // a facing edge of a near rect; edges in the order of sensors.py:668-671
                    const int ent = s_edge[(mq == SEG_STATIC ? 0 : 4 * cap_rs) + m];
                    const int at = ent >> 2;
                    const int4 q = s_rect[at]; sm = s_rmask[at];
                    const float l = (float)q.x, t = (float)q.y, r = (float)(q.x + q.z), b = (float)(q.y + q.w);
                    const int e = ent & 3;
                    sg = e == 0 ? make_float4(l, b, r, b) : e == 1 ? make_float4(r, t, r, b) : e == 2 ? make_float4(r, t, l, t) : make_float4(l, b, l, t);
                } else if (mq == SEG_CORRIDOR) {             // polyline segment p -> p+1 of the right (0) / left (1) border
                    if (staged) {
                        const unsigned ref = s_cref[m];
                        const int p = ref & 0x7fff; sm = ref >> 16;
                        const float4 u = s_corr[p], v = s_corr[(p + 1) & cmask];
                        sg = (ref >> 15) & 1u ? make_float4(u.z, u.w, v.z, v.w) : make_float4(u.x, u.y, v.x, v.y);
                    } else {                                 // span too long for the LDS copy: segment m of the span, straight from the ring
                        const int p = umin + (m >> 1);
                        sm = 0;
#pragma unroll
                        for (int a = 0; a < HM; a++) if (a < nsnap && win_lo[a] <= p && p + 1 < win_hi[a]) sm |= 1u << a;
                        const float4 u = corr_f32(p), v = corr_f32(p + 1);
                        sg = (m & 1) ? make_float4(u.z, u.w, v.z, v.w) : make_float4(u.x, u.y, v.x, v.y);
                    }
                } else { sg = s_green[m]; sm = s_gmask[m]; }
            };
            // the reference's intersection test of ray `ray` (index into s_ray / s_best) with segment sgx
            auto test = [&](int ray, const float4& sgx, unsigned smx) {
                const double2 e = s_ray[ray];
                double d2;
                if (hit_segment(cx, cy, e.x, e.y, (float)e.x, (float)e.y, sgx, d2)) {
                    d2 = sqrt(d2);      // the distance itself (sensors.py:920-921); monotone, so the minimum of the roots is the root of the minimum
                    const unsigned bits = __float_as_uint((float)d2);
#pragma unroll
                    for (int j = 0; j < HM; j++) if ((smx >> j) & 1u) atomicMin(&s_best[j * P.total_rays + ray], bits);
                }
            };
            const float fdir_rad = (float)(fdir * kDeg2Rad);
            // Few table entries face a ray at all and fewer face more than one (about 30 candidates in 45 entries per env-step on the
            // bench workload, spread over the sensors): tested where they are found, each round keeps a handful of lanes busy.  The
            // candidates of a chunk -- all sensors -- go to a list of (lane that holds the segment, ray) pairs instead; the list is
            // tested densely, 64 pairs at a time, the segment coming from its lane by a cross-lane read.
            int np_u = 0;                                  // entries in the list (wave-uniform)
            for (int w0 = 0; w0 < n_items; w0 += FTL_WAVE) {
                const int w = w0 + lane;
                int m = -1, mq = 0;
                {   // the flattened list is four runs whose ends are wave-uniform
                    int j = 0, start = 0, end = 0;
#pragma unroll
                    for (int q = 0; q < SEG_CLASSES; q++) {
                        end += cq[q];
                        const bool past = w >= end;
                        j += past ? 1 : 0; start = past ? end : start;
                    }
                    if (w < n_items) { m = w - start; mq = j; }
                }
                FTL_RTIC(3);
                float4 sg = make_float4(0.f, 0.f, 0.f, 0.f); unsigned sm = 0;
                float angA = 0.0f, angB = 0.0f, dmin2 = 3.0e38f;
                if (m >= 0) {
                    fetch(mq, m, sg, sm);
                    if (sm == 0u) m = -1;                    // (unstaged corridor spans list every segment: one outside all windows faces no ray)
                }
                if (m >= 0) {
                    const float ax = sg.x - cx, ay = sg.y - cy, bx = sg.z - cx, by = sg.w - cy;
                    angA = arc_atan2(ay, ax); angB = arc_atan2(by, bx);
                    // closest approach of the segment to the follower (culling only: 2 px of slack in the records' reach)
                    const float ex_ = bx - ax, ey_ = by - ay;
                    const float l2 = __builtin_fmaf(ex_, ex_, ey_ * ey_);
                    const float tt = l2 > 0.0f ? fminf(fmaxf(__fdividef(-__builtin_fmaf(ax, ex_, ay * ey_), l2), 0.0f), 1.0f) : 0.0f;
                    const float nx = __builtin_fmaf(tt, ex_, ax), ny = __builtin_fmaf(tt, ey_, ay);
                    dmin2 = __builtin_fmaf(nx, nx, ny * ny);
                }
                FTL_RTIC(4);
                auto flush = [&](int np) {
                    __syncthreads();                                     // the pairs are in LDS
                    for (int p0 = 0; p0 < np; p0 += FTL_WAVE) {
                        const int p = p0 + lane;
                        const unsigned rec = p < np ? s_pair[p] : 0u;
                        const int L = (int)(rec >> 10);
                        const float4 sgp = make_float4(__shfl(sg.x, L), __shfl(sg.y, L), __shfl(sg.z, L), __shfl(sg.w, L));
                        const unsigned smp = (unsigned)__shfl((int)sm, L);
                        if (p < np) test((int)(rec & 1023u), sgp, smp);
                    }
                    __syncthreads();                                     // read before the next pairs overwrite them
                };
#pragma nounroll
                for (int k = 0; k < c.n_lasers; k++) {
                    const FtlRaySensor rs = P.ray_sens[k];                 // wave-uniform: two scalar loads
                    if ((int)((rs.flags >> 5) & 1u) != which || (rs.flags & 64u)) continue;
                    const int N = rs.count, rbase = rs.rbase;
                    int i0 = 0, cnt = 0;
                    if (m >= 0 && ((rs.flags >> mq) & 1u) && !(dmin2 > rs.reach2)) {      // the sensor reacts to this class and can reach the segment
                        // candidate rays: the arc [uA, uB] the segment subtends, in units of the ray spacing from ray 0
                        const float fN = (float)N;
                        const float phis = __builtin_fmaf(fdir_rad, rs.inv_step, rs.off_u);
                        float uA = __builtin_fmaf(angA, rs.inv_step, -phis), uB = __builtin_fmaf(angB, rs.inv_step, -phis);
                        uA = __builtin_fmaf(-floorf(uA * rs.inv_count), fN, uA); uB = __builtin_fmaf(-floorf(uB * rs.inv_count), fN, uB);   // into [0, N) (an ulp outside is absorbed by the wrap below)
                        float diff = uB - uA; if (diff < 0.0f) diff += fN;
                        float start = uA, wd = diff;
                        if (diff > 0.5f * fN) { start = uB; wd = fN - diff; }
                        if ((rs.flags & 16u) || dmin2 < 4.0f || wd > 0.5f * fN - 0.05f) { i0 = 0; cnt = N; }   // through / next to the origin, or rays
                                                                                              // at explicit angles (<= 7 of them): every ray
                        else {
                            i0 = (int)ceilf(start - rs.slack);                              // slack >= 0.01 rad, far above float error
                            cnt = (int)floorf(start + wd + rs.slack) - i0 + 1;
                            cnt = cnt > N ? N : cnt;
                        }
                    }
#if defined(FTL_RAYS_STOP) && FTL_RAYS_STOP == 5   // diagnostic: decode + fetch + arcs only, no ray tests
                    cnt = cnt == 0x7fffffff ? 1 : 0;
#endif
#ifdef FTL_PROFILE_RAYS
                    if (threadIdx.x == 0) { s_rcyc[8] += 1; }
                    { int mc = cnt; for (int o = 32; o >= 1; o >>= 1) mc = max(mc, __shfl_xor(mc, o)); int ni = __popcll(__ballot(m >= 0)); int sc = cnt; for (int o = 32; o >= 1; o >>= 1) sc += __shfl_xor(sc, o); if (threadIdx.x == 0) { s_rcyc[9] += mc; s_rcyc[10] += sc; s_rcyc[11] += ni; s_rcyc[12] += (mc <= 2); s_rcyc[13] += (mc > 2 && mc <= 4); s_rcyc[14] += (mc > 4 && mc <= 8); s_rcyc[15] += (mc > 8); } }
#endif
                    // (a) segments with more than FTL_WIDE_ARC candidates: the wavefront writes their pairs, one ray per lane; more than a
                    //     wavefront of them (a segment next to the follower under a 180-ray sensor) are tested on the spot
                    unsigned long long wide = __ballot(cnt > FTL_WIDE_ARC);
                    while (wide) {
                        const int L = __ffsll((long long)wide) - 1; wide &= wide - 1;
                        const int i0L = __builtin_amdgcn_readlane(i0, L), cntL = __builtin_amdgcn_readlane(cnt, L);
                        if (cntL > FTL_WAVE) {
                            const float4 sgL = make_float4(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg.x), L)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg.y), L)),
                                                           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg.z), L)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg.w), L)));
                            const unsigned smL = (unsigned)__builtin_amdgcn_readlane((int)sm, L);
                            for (int t = lane; t < cntL; t += FTL_WAVE) {
                                int i = i0L + t; i = i < 0 ? i + N : (i >= N ? i - N : i);
                                test(rbase + i, sgL, smL);
                            }
                            continue;
                        }
                        if (np_u + cntL > FTL_PAIR_CAP) { flush(np_u); np_u = 0; }
                        if (lane < cntL) {
                            int i = i0L + lane; i = i < 0 ? i + N : (i >= N ? i - N : i);
                            s_pair[np_u + lane] = (unsigned short)((L << 10) | (rbase + i));
                        }
                        np_u += cntL;
                    }
                    // (b) the others: each lane appends its own 0..FTL_WIDE_ARC pairs at the prefix sum of the counts (no atomics: the
                    //     count is two bits, a ballot per bit gives every lane its offset and the wavefront the total)
                    const int own = cnt > FTL_WIDE_ARC ? 0 : cnt;
                    int pre = 0, tot = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) if ((FTL_WIDE_ARC >> b) != 0) {
                        const unsigned long long bm = __ballot((own >> b) & 1);
                        pre += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u)) << b;
                        tot += __popcll(bm) << b;
                    }
                    if (np_u + tot > FTL_PAIR_CAP) { flush(np_u); np_u = 0; }       // (tot <= 64 x FTL_WIDE_ARC = FTL_PAIR_CAP)
                    for (int t = 0; t < own; t++) {
                        int i = i0 + t; i = i < 0 ? i + N : (i >= N ? i - N : i);
                        s_pair[np_u + pre + t] = (unsigned short)((lane << 10) | (rbase + i));
                    }
                    np_u += tot;
                }
                if (np_u > 0) { flush(np_u); np_u = 0; }      // before the next chunk replaces the segments in the lanes
            }
        }
        FTL_RTIC(5);
#if defined(FTL_RAYS_STOP) && FTL_RAYS_STOP == 4
        continue;
#endif
        __syncthreads();
        // ---- phase 4: rows, oldest first, newest last (sensors.py:896-901); rows older than the first scan and rays without a
        // hit read |end - origin| (sensors.py:925-930).  pad_sectors (sensors.py:932-953) spreads a row over four N-wide
        // sector blocks; the optional fused ContinuousObserveModifier_sensorPrev output (wrappers.py:200-221) is
        // clip(x / laser_length, 0, 1) in float32, sensors concatenated along the row.
        {
            float* pol = C.out.policy_obs ? C.out.policy_obs + (size_t)env * P.pol_h * P.pol_width : nullptr;
            bool any_pad = false;
            FTL_FOR_LASERS(k) if (EXPL && c.lasers[k].after_tracker == which && c.lasers[k].pad_sectors) {
                any_pad = true;
                const int tot = c.lasers[k].history * 4 * c.lasers[k].count;
                for (int i = lane; i < tot; i += FTL_WAVE) out_base[c.lasers[k].out_offset + i] = 0.0f;
                if (pol && P.pol_off[k] >= 0) for (int i = lane; i < tot; i += FTL_WAVE) {
                    int row = i / (4 * c.lasers[k].count), col = i - row * 4 * c.lasers[k].count;
                    pol[row * P.pol_width + P.pol_off[k] + col] = 0.0f;
                }
            }
            if (any_pad) __syncthreads();
            if constexpr (!EXPL) {
                // One RAY per lane, the rays of all sensors of the pass in one index space (as in phase 2); the lane writes its ray's H rows
                // (one (ray, age) pair per lane, below, takes three rounds on the bench workload and fifteen under a 180-ray sensor).
                int n_rays = 0;
                FTL_FOR_LASERS(k) if (c.lasers[k].after_tracker == which) n_rays += c.lasers[k].count;
                for (int g = lane; g < n_rays; g += FTL_WAVE) {
                    int i = g, H = 0, N = 0, ooff = 0, poff = -1; float flen = 1.0f; bool found = false;
                    FTL_FOR_LASERS(k) if (c.lasers[k].after_tracker == which) {
                        const int Nk = c.lasers[k].count;
                        if (!found && i < Nk) {
                            found = true; N = Nk; H = c.lasers[k].history; ooff = c.lasers[k].out_offset; poff = P.pol_off[k];
                            flen = (float)c.lasers[k].length;                   // python number / float32 array -> float32 division
                        }
                        if (!found) i -= Nk;
                    }
                    const float miss = s_miss[g];
#pragma unroll
                    for (int a2 = 0; a2 < HM; a2++) if (a2 < H) {
                        const unsigned bb = s_best[a2 * P.total_rays + g];
                        const float vf = (a2 < nsnap && bb != kInfBits) ? __uint_as_float(bb) : miss;
                        out_base[ooff + (H - 1 - a2) * N + i] = vf;
                        if (pol && poff >= 0) pol[(H - 1 - a2) * P.pol_width + poff + i] = fminf(fmaxf(vf / flen, 0.0f), 1.0f);
                    }
                }
            } else {
                // the configs with the rarer sensor features (few rays per pass: a 12-ray sensor beside the compas ones): one (ray, age) pair
                // per lane, sensor by sensor (the sensor's parameters stay wave-uniform): pair q = age * N + ray
                int rbase = 0;
                FTL_FOR_LASERS(k) {
                    const int N = c.lasers[k].count;
                    if (c.lasers[k].after_tracker != which) { continue; }
                    const int H = c.lasers[k].history, ooff = c.lasers[k].out_offset, poff = P.pol_off[k], rb = rbase;
                    const bool pad = c.lasers[k].pad_sectors != 0;
                    const float flen = (float)c.lasers[k].length;                   // python number / float32 array -> float32 division
                    const int Wd = pad ? 4 * N : N;
                    for (int q = lane; q < N * H; q += FTL_WAVE) {
                        int a2 = 0;
#pragma unroll
                        for (int j = 1; j < HM; j++) a2 += (q >= j * N) ? 1 : 0;     // q / N without an integer division (a2 < H <= HM)
                        const int i = q - a2 * N;
                        int col = i;
                        if (pad) {
                            const double lis = (double)N / 4.0, di = (double)i;     // lasers_in_sector (sensors.py:938)
                            col = (di < lis ? 0 : (di < 2 * lis ? 1 : (di < 3 * lis ? 2 : 3))) * N + i;
                        }
                        const unsigned bb = s_best[a2 * P.total_rays + rb + i];
                        const float vf = (a2 < nsnap && bb != kInfBits) ? __uint_as_float(bb) : s_miss[rb + i];
                        out_base[ooff + (H - 1 - a2) * Wd + col] = vf;
                        if (pol && poff >= 0) pol[(H - 1 - a2) * P.pol_width + poff + col] = fminf(fmaxf(vf / flen, 0.0f), 1.0f);
                    }
                    rbase += N;
                }
            }
        }
        FTL_RTIC(6);
    }
#ifdef FTL_PROFILE_RAYS
    __syncthreads();
    if (lane < 16) atomicAdd(&g_rcyc[lane], s_rcyc[lane]);
#endif
}

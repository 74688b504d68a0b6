// ftl_device.hpp -- CDNA4 (gfx950) device code of the batched Game.step().
//
// One 64-lane wavefront advances one environment (one 64-thread workgroup per env).  A step() is two launches on
// the same stream, split where the live state is smallest:
//
//   ftl_frames_kernel  -- frames_per_step x Game.frame_step (follow_the_leader_continuous_env.py:947-1141), the two
//       LeaderPositionsTracker_v2 scans of use_sensors (classes.py:263-267, 285-286; sensors.py:243-327), the history
//       snapshot push of the ray sensors (sensors.py:896-897) and _get_obs (1789-1810):
//       * robots live on lanes 0..R-1 (0 leader, 1 follower, 2.. bears): controller + f64 sin/cos integrator + integer
//         hitbox update of AbstractRobot.move() (classes.py:134-182) run once per frame for ALL robots in lock-step,
//         steering (classes.py:184-215) once for leader + bears;
//       * integer-rect collision tests put one static obstacle on each lane and reduce with a ballot;
//       * green-zone window / closest-point searches (1828-1843, 1906-1960) stride the factual trajectory over the
//         lanes and finish with a wave arg-min.
//   ftl_rays_kernel -- LeaderCorridor_Prev_lasers_v2.scan (sensors.py:883-962) for every ray sensor: one RAY per lane,
//       obstacle segments (static rects, H-deep history of dynamic rects, f32 corridor ring) staged in LDS and walked
//       with wave-uniform control flow (uniform distance culling, LDS broadcast reads); one nearest-hit accumulator
//       per history snapshot in registers.
// No MFMA: there is no dense contraction anywhere on this path.
//
// Numerics follow oracle/ftl_oracle.c operation by operation (same dtype flow, explicit fma only where numpy/BLAS
// fuse); the translation unit is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ftl.h"

#define FTL_WAVE 64
#define FTL_HMAX 8          // compile-time cap on max_prev_obs (register accumulators)
#define FTL_DCHUNK 256      // trajectory segment lengths staged per pass of the green-zone walk
#ifndef FTL_FRAMES_WPE
#define FTL_FRAMES_WPE 3    // min waves per SIMD the register allocator must leave room for (tuned on MI355X)
#endif
#ifndef FTL_RAYS_WPE
#define FTL_RAYS_WPE 4
#endif

struct FtlDevParams {
    ftl_config cfg;
    int32_t n_envs, R, lasers_len, total_rays, hmax, lds_frames, lds_rays;
    int32_t rays_k[FTL_MAX_LASERS];   // first global ray id of sensor k
    // per-env state (views into the caller-owned state buffer), all [n_envs][...]
    float* rb_pos; double* rb_dbl; int32_t* rb_int; int32_t* env_int; double* env_dbl;
    float* traj; double* hist; double* corr; int32_t* snap_rects; int32_t* snap_win;
    ftl_scenarios scen;
};
// per-call arguments (passed by value in the kernarg segment)
struct FtlCall {
    const double* action; const int32_t* scen_idx; const uint8_t* mask;
    ftl_outputs out;
    uint32_t flags; int32_t mode;      // mode 0 = step, 1 = reset
};

namespace ftl {

static constexpr double kDeg2Rad = 3.141592653589793 / 180.0;
static constexpr double kRad2Deg = 180.0 / 3.141592653589793;

// ---------------------------------------------------------------- cross-lane helpers
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ double rl_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_readlane(hi, lane), __builtin_amdgcn_readlane(lo, lane));
}

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

// numpy pairwise summation over all n elements (what np.sum does), operands read from LDS
template <typename T>
__device__ __forceinline__ T pairwise_le128(const T* a, int n) {
    if (n < 8) { T r = (T)0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    T r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
template <typename T, int DEPTH>
__device__ T pairwise_rec(const T* a, int n) {       // numpy's recursive halving above 128 elements
    if (n <= 128) return pairwise_le128(a, n);
    if constexpr (DEPTH == 0) return pairwise_le128(a, n);   // unreachable: n <= 128 << levels is validated on the host
    else {
        int n2 = n / 2; n2 -= n2 % 8;
        return pairwise_rec<T, DEPTH - 1>(a, n2) + pairwise_rec<T, DEPTH - 1>(a + n2, n - n2);
    }
}
template <typename T>
__device__ T pairwise_sum(const T* a, int n) { return pairwise_rec<T, 2>(a, n); }   // n <= 512

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
__device__ __forceinline__ void robot_move(Robot& r, const Limits& L, bool active) {
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
    if (turning) {
        // New hitbox size = pygame.transform.rotate(image, -direction) (classes.py:173-175).  The bounding box only
        // needs int(|cos|w+|sin|h): take it from the sin/cos of the movement (the rotate call rounds the angle to
        // float32, which moves the value by < (w+h)*4e-7) unless that value sits next to an integer or the float32
        // angle is a multiple of 90 degrees -- then rotate_size() evaluates the reference expression itself.
        double a32 = (double)(float)(-direction);
        double vw = fabs(c) * L.img_w + fabs(s) * L.img_h, vh = fabs(s) * L.img_w + fabs(c) * L.img_h;
        double tol = (double)(L.img_w + L.img_h) * 1e-6;
        double fw = vw - floor(vw), fh = vh - floor(vh);
        bool exact = (rint(a32 / 90.0) * 90.0 == a32) || fw < tol || fw > 1.0 - tol || fh < tol || fh > 1.0 - tol;
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
// classes.py:184-215: steering toward (tx,ty); has_speed false means "speed=None" (use the distance)
__device__ __forceinline__ void steer_to_point(Robot& r, const Limits& L, double tx, double ty, bool has_speed, double speed) {
    double new_speed = has_speed ? speed : euclid_f64((double)r.px, (double)r.py, tx, ty);
    int desirable = (int)angle_to_point((double)r.px, (double)r.py, tx, ty);
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

// wave arg-min of (f32 value, index) with first-index tie-break (np.argmin)
__device__ __forceinline__ void wave_argmin(float& v, int& idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(v, off);
        int oi = __shfl_xor(idx, off);
        bool take = (ov < v) || (ov == v && oi < idx);
        if (take) { v = ov; idx = oi; }
    }
}

// ---------------------------------------------------------------- the environment held by one wave (frames kernel)
struct EnvCtx {
    const FtlDevParams& P;
    const FtlCall& C;
    int env, lane;
    // LDS carve-up of this wave
    int4* s_static;      // [n_static]
    float* s_d;          // [FTL_DCHUNK] float / reused as double scratch of the tracker
    // wave-uniform scalars
    int scen, cur_target_id, leader_finished, done, crash, is_in_box, is_on_trace, too_close;
    int step_count, finish_timer, traj_len, trk_counter, corr_lo, corr_hi, seed_end, snap_count;
    int error, episodes, green_count, green_len, scan_ok;
    double acc_penalty, overall_reward;
    double cur_tx, cur_ty;
    int route_len;
    Robot rb;            // this lane's robot (lanes >= R hold a benign dummy that is never committed)
    __device__ EnvCtx(const FtlDevParams& p, const FtlCall& c) : P(p), C(c) {}
};

__device__ __forceinline__ const double* route_ptr(const EnvCtx& E) {
    return E.P.scen.route + (size_t)E.scen * E.P.cfg.route_cap * 2;
}
__device__ __forceinline__ float* traj_ptr(const EnvCtx& E) { return E.P.traj + (size_t)E.env * E.P.cfg.traj_cap * 2; }

__device__ __forceinline__ void stage_static(int4* s_static, const FtlDevParams& P, int scen, int lane) {
    const int4* src = reinterpret_cast<const int4*>(P.scen.static_rects) + (size_t)scen * P.cfg.n_static;
    for (int s = lane; s < P.cfg.n_static; s += FTL_WAVE) s_static[s] = src[s];
}

// ---- load / store of the per-env state ------------------------------------------------------------------------
__device__ __forceinline__ void env_load(EnvCtx& E) {
    const FtlDevParams& P = E.P;
    const int* ei = P.env_int + (size_t)E.env * FTL_EI_COUNT;
    const double* ed = P.env_dbl + (size_t)E.env * FTL_ED_COUNT;
    E.scen = ei[FTL_EI_SCEN]; E.cur_target_id = ei[FTL_EI_TARGET_ID]; E.leader_finished = ei[FTL_EI_LEADER_FINISHED];
    E.done = ei[FTL_EI_DONE]; E.crash = ei[FTL_EI_CRASH]; E.is_in_box = ei[FTL_EI_IN_BOX]; E.is_on_trace = ei[FTL_EI_ON_TRACE];
    E.too_close = ei[FTL_EI_TOO_CLOSE]; E.step_count = ei[FTL_EI_STEP_COUNT]; E.finish_timer = ei[FTL_EI_FINISH_TIMER];
    E.traj_len = ei[FTL_EI_TRAJ_LEN]; E.trk_counter = ei[FTL_EI_TRK_COUNTER]; E.corr_lo = ei[FTL_EI_CORR_LO];
    E.corr_hi = ei[FTL_EI_CORR_HI]; E.seed_end = ei[FTL_EI_SEED_END]; E.snap_count = ei[FTL_EI_SNAP_COUNT];
    E.error = ei[FTL_EI_ERROR]; E.episodes = ei[FTL_EI_EPISODES]; E.green_count = ei[FTL_EI_GREEN_COUNT]; E.green_len = ei[FTL_EI_GREEN_LEN];
    E.acc_penalty = ed[FTL_ED_ACC_PENALTY]; E.overall_reward = ed[FTL_ED_OVERALL_REWARD];
    E.cur_tx = ed[FTL_ED_SPARE0]; E.cur_ty = ed[FTL_ED_SPARE1];
    int r = (E.lane < P.R) ? E.lane : 0;     // idle lanes mirror robot 0 (never committed)
    size_t ro = (size_t)E.env * P.R + r;
    E.rb.px = P.rb_pos[2 * ro]; E.rb.py = P.rb_pos[2 * ro + 1];
    const double* rd = P.rb_dbl + ro * FTL_RD_COUNT;
    E.rb.direction = rd[FTL_RD_DIRECTION]; E.rb.speed = rd[FTL_RD_SPEED]; E.rb.rot_speed = rd[FTL_RD_ROT_SPEED];
    E.rb.des_speed = rd[FTL_RD_DES_SPEED]; E.rb.des_rot_speed = rd[FTL_RD_DES_ROT_SPEED];
    const int* ri = P.rb_int + ro * FTL_RI_COUNT;
    E.rb.rx = ri[FTL_RI_X]; E.rb.ry = ri[FTL_RI_Y]; E.rb.rw = ri[FTL_RI_W]; E.rb.rh = ri[FTL_RI_H];
    E.rb.rot_dir = ri[FTL_RI_ROT_DIR]; E.rb.des_rot_dir = ri[FTL_RI_DES_ROT_DIR];
    int b = (E.lane >= 2 && E.lane < P.R) ? E.lane - 2 : 0;
    E.rb.tgt_x = ed[FTL_ED_BEAR_POINTS + 2 * b]; E.rb.tgt_y = ed[FTL_ED_BEAR_POINTS + 2 * b + 1];
    E.rb.dyn_index = ei[FTL_EI_DYN_INDEX0 + b];
    stage_static(E.s_static, P, E.scen, E.lane);
    E.route_len = P.scen.route_len[E.scen];
}

__device__ __forceinline__ void env_store(EnvCtx& E) {
    const FtlDevParams& P = E.P;
    int* ei = P.env_int + (size_t)E.env * FTL_EI_COUNT;
    double* ed = P.env_dbl + (size_t)E.env * FTL_ED_COUNT;
    if (E.lane == 0) {
        ei[FTL_EI_SCEN] = E.scen; ei[FTL_EI_TARGET_ID] = E.cur_target_id; ei[FTL_EI_LEADER_FINISHED] = E.leader_finished;
        ei[FTL_EI_DONE] = E.done; ei[FTL_EI_CRASH] = E.crash; ei[FTL_EI_IN_BOX] = E.is_in_box; ei[FTL_EI_ON_TRACE] = E.is_on_trace;
        ei[FTL_EI_TOO_CLOSE] = E.too_close; ei[FTL_EI_STEP_COUNT] = E.step_count; ei[FTL_EI_FINISH_TIMER] = E.finish_timer;
        ei[FTL_EI_TRAJ_LEN] = E.traj_len; ei[FTL_EI_TRK_COUNTER] = E.trk_counter; ei[FTL_EI_CORR_LO] = E.corr_lo;
        ei[FTL_EI_CORR_HI] = E.corr_hi; ei[FTL_EI_SEED_END] = E.seed_end; ei[FTL_EI_SNAP_COUNT] = E.snap_count;
        ei[FTL_EI_ERROR] = E.error; ei[FTL_EI_EPISODES] = E.episodes; ei[FTL_EI_GREEN_COUNT] = E.green_count; ei[FTL_EI_GREEN_LEN] = E.green_len;
        ei[FTL_EI_SCAN_OK] = E.scan_ok; ei[FTL_EI_SPARE] = 0;
        ed[FTL_ED_ACC_PENALTY] = E.acc_penalty; ed[FTL_ED_OVERALL_REWARD] = E.overall_reward;
        ed[FTL_ED_SPARE0] = E.cur_tx; ed[FTL_ED_SPARE1] = E.cur_ty;
    }
    if (E.lane < P.R) {
        size_t ro = (size_t)E.env * P.R + E.lane;
        P.rb_pos[2 * ro] = E.rb.px; P.rb_pos[2 * ro + 1] = E.rb.py;
        double* rd = P.rb_dbl + ro * FTL_RD_COUNT;
        rd[FTL_RD_DIRECTION] = E.rb.direction; rd[FTL_RD_SPEED] = E.rb.speed; rd[FTL_RD_ROT_SPEED] = E.rb.rot_speed;
        rd[FTL_RD_DES_SPEED] = E.rb.des_speed; rd[FTL_RD_DES_ROT_SPEED] = E.rb.des_rot_speed;
        int* ri = P.rb_int + ro * FTL_RI_COUNT;
        ri[FTL_RI_X] = E.rb.rx; ri[FTL_RI_Y] = E.rb.ry; ri[FTL_RI_W] = E.rb.rw; ri[FTL_RI_H] = E.rb.rh;
        ri[FTL_RI_ROT_DIR] = E.rb.rot_dir; ri[FTL_RI_DES_ROT_DIR] = E.rb.des_rot_dir; ri[FTL_RI_SPARE0] = 0; ri[FTL_RI_SPARE1] = 0;
        if (E.lane >= 2) {
            int b = E.lane - 2;
            ed[FTL_ED_BEAR_POINTS + 2 * b] = E.rb.tgt_x; ed[FTL_ED_BEAR_POINTS + 2 * b + 1] = E.rb.tgt_y;
            ei[FTL_EI_DYN_INDEX0 + b] = E.rb.dyn_index;
        }
    }
}

// ---- reset(): ENV:494-543 from scenario `scen` ----------------------------------------------------------------
__device__ __forceinline__ void env_reset(EnvCtx& E, int scen) {
    const FtlDevParams& P = E.P;
    const ftl_config& c = P.cfg;
    E.scen = scen;
    int r = (E.lane < P.R) ? E.lane : 0;
    size_t so = (size_t)scen * P.R + r;
    E.rb.px = P.scen.robot_pos[2 * so]; E.rb.py = P.scen.robot_pos[2 * so + 1];
    E.rb.direction = P.scen.robot_dir[so];
    E.rb.speed = 0; E.rb.rot_speed = 0; E.rb.des_speed = 0; E.rb.des_rot_speed = 0; E.rb.rot_dir = 0; E.rb.des_rot_dir = 0;
    const int* rr = P.scen.robot_rect + so * 4;
    E.rb.rx = rr[0]; E.rb.ry = rr[1]; E.rb.rw = rr[2]; E.rb.rh = rr[3];
    __syncthreads();
    stage_static(E.s_static, P, scen, E.lane);
    E.route_len = P.scen.route_len[scen];
    // initial leader_factual_trajectory (ENV:533-539)
    int n0 = P.scen.init_traj_len[scen];
    const float2* src = reinterpret_cast<const float2*>(P.scen.init_traj) + (size_t)scen * c.init_traj_cap;
    float2* dst = reinterpret_cast<float2*>(traj_ptr(E));
    for (int k = E.lane; k < n0; k += FTL_WAVE) dst[k] = src[k];
    E.traj_len = n0;
    E.step_count = 0; E.acc_penalty = 0; E.overall_reward = 0;
    E.done = 0; E.crash = 0; E.is_in_box = 0; E.is_on_trace = 0; E.too_close = 0;
    E.cur_target_id = 1; E.leader_finished = 0; E.finish_timer = -1;
    E.green_count = 0; E.green_len = -1; E.error = 0; E.scan_ok = 0;
    float lpx = rl_f(E.rb.px, 0), lpy = rl_f(E.rb.py, 0);
    const double* rt = route_ptr(E);
    if (E.route_len == 0) { E.done = 1; E.cur_tx = (double)lpx; E.cur_ty = (double)lpy; }
    else { int id = E.route_len > 1 ? 1 : 0; E.cur_tx = rt[2 * id]; E.cur_ty = rt[2 * id + 1]; }
    // ENV:717-718: every bear starts from the LAST bear_start_position, (leader - 150, leader - 150) in float32
    E.rb.tgt_x = (double)(lpx - 150.0f); E.rb.tgt_y = (double)(lpy - 150.0f); E.rb.dyn_index = 0;
    E.trk_counter = 0; E.corr_lo = 0; E.corr_hi = 0; E.seed_end = 0; E.snap_count = 0;
    __syncthreads();
}

// ---- one frame: ENV:947-1141 ----------------------------------------------------------------------------------
// ENV:1828-1843, sequential form: f64 running sum of the segment lengths from the newest point backwards
__device__ __forceinline__ int green_walk_seq(EnvCtx& E) {
    const float2* tr = reinterpret_cast<const float2*>(traj_ptr(E));
    const double maxd = E.P.cfg.max_distance;
    double acc = 0.0; int G = 0; int k = E.traj_len - 2; bool stop = false;
    while (k >= 0 && !stop) {
        int cnt = (k + 1 < FTL_DCHUNK) ? k + 1 : FTL_DCHUNK;
        __syncthreads();
        for (int i = E.lane; i < cnt; i += FTL_WAVE) {
            float2 cur = tr[k - i], prev = tr[k - i + 1];
            E.s_d[i] = (float)euclid_f32(prev.x, prev.y, cur.x, cur.y);
        }
        __syncthreads();
        // acc is monotone, so counting acc<=maxd over the chunk equals the prefix count
        for (int i = 0; i < cnt; i++) { acc += (double)E.s_d[i]; G += (acc <= maxd); }
        stop = !(acc <= maxd);
        k -= cnt;
    }
    return G;
}
// Same count from a wave-parallel prefix sum.  The parallel sums differ from the sequential ones by rounding only
// (< 1e-10 for any trajectory this state can hold), so whenever no partial sum lies within 1e-6 of max_distance the
// comparisons -- hence the count -- are identical; otherwise the sequential form decides.
__device__ __forceinline__ int green_walk(EnvCtx& E) {
    const float2* tr = reinterpret_cast<const float2*>(traj_ptr(E));
    const double maxd = E.P.cfg.max_distance;
    const int lane = E.lane;
    double base = 0.0; int G = 0; int k = E.traj_len - 2;
    while (k >= 0) {
        int cnt = (k + 1 < FTL_DCHUNK) ? k + 1 : FTL_DCHUNK;
        // lane handles elements 4*lane .. 4*lane+3 of the chunk (element i <-> points k-i, k-i+1)
        double p[4]; bool valid[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int i = 4 * lane + j;
            valid[j] = i < cnt;
            double d = 0.0;
            if (valid[j]) { float2 cur = tr[k - i], prev = tr[k - i + 1]; d = euclid_f32(prev.x, prev.y, cur.x, cur.y); }
            p[j] = (j == 0) ? d : p[j - 1] + d;
        }
        double incl = p[3];             // inclusive scan of the lane totals
#pragma unroll
        for (int off = 1; off < FTL_WAVE; off <<= 1) {
            double o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        double excl = incl - p[3] + base;
        bool near = false; int below = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double v = excl + p[j];
            if (valid[j]) { near |= fabs(v - maxd) < 1e-6; below += (v <= maxd); }
        }
        if (__ballot(near) != 0ull) return green_walk_seq(E);
        // total of `below` over the wave
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) below += __shfl_xor(below, off);
        G += below;
        base = rl_d(incl, FTL_WAVE - 1) + base;
        if (!(base <= maxd)) break;
        k -= cnt;
    }
    return G;
}

// first-index arg-min of the f32 squared distance to (px,py) over `n` trajectory points, point i = tr[base + i*stride]
__device__ __forceinline__ int closest_point(const EnvCtx& E, float px, float py, int base, int stride, int n) {
    const float2* tr = reinterpret_cast<const float2*>(E.P.traj + (size_t)E.env * E.P.cfg.traj_cap * 2);
    float best = __int_as_float(0x7f800000); int bi = 0x7fffffff;
    for (int i = E.lane; i < n; i += FTL_WAVE) {
        float2 q = tr[base + i * stride];
        float dx = q.x - px, dy = q.y - py;
        float d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; bi = i; }
    }
    wave_argmin(best, bi);
    return bi;
}

__device__ __forceinline__ void frame_step(EnvCtx& E, const Limits& L, double& reward, int& i0, int& i1, int& i2) {
    const FtlDevParams& P = E.P;
    const ftl_config& c = P.cfg;
    const int lane = E.lane;
    E.is_in_box = 0; E.is_on_trace = 0;
    i0 = FTL_MISSION_IN_PROGRESS; i1 = FTL_AGENT_MOVING; i2 = FTL_LEADER_MOVING;

    // state of the other robots as the follower's collision test and the bears' way-points see it (= before
    // any of them moves this frame: the follower moves first, ENV:957, bears ENV:987-995, leader ENV:1048-1058)
    const float lpx0 = rl_f(E.rb.px, 0), lpy0 = rl_f(E.rb.py, 0);
    const double ldir0 = rl_d(E.rb.direction, 0);
    const int orx = E.rb.rx, ory = E.rb.ry, orw = E.rb.rw, orh = E.rb.rh;   // this lane's robot rect before the move

    // leader way-point switch (ENV:978-983), uses the leader position before its move
    if (euclid_f64_lt((double)lpx0, (double)lpy0, E.cur_tx, E.cur_ty, c.leader_pos_epsilon)) {
        E.cur_target_id += 1;
        if (E.cur_target_id >= E.route_len) E.leader_finished = 1;
        else { const double* rt = route_ptr(E); E.cur_tx = rt[2 * E.cur_target_id]; E.cur_ty = rt[2 * E.cur_target_id + 1]; }
    }
    // bears: way-point choice (ENV:722-758, 819-837), one bear per lane; the point depends on the leader pose only
    double tx = E.cur_tx, ty = E.cur_ty;
    const bool is_bear = lane >= 2 && lane < P.R;
    if (c.n_bears > 0 && is_bear) {
        const int b = lane - 2;
        bool near = euclid_f64_lt((double)E.rb.px, (double)E.rb.py, E.rb.tgt_x, E.rb.tgt_y, c.leader_pos_epsilon);
        double off, lvl;
        if (c.move_bear_v4 && (b & 1)) {
            if (near) E.rb.dyn_index += 1;
            if (E.rb.dyn_index > 3) E.rb.dyn_index = 0;
            // p1=(150,+140) p2=(150,-140) p3=(250,-160) p4=(250,+160); per-index orders of ENV:742-749
            const int order = (b == 1) ? 0x2134 /*p4,p3,p1,p2*/ : 0x4213 /*p3,p1,p2,p4*/;
            int p = (order >> (4 * E.rb.dyn_index)) & 0xf;
            lvl = (p <= 2) ? 150.0 : 250.0;
            off = (p == 1) ? 140.0 : (p == 2) ? -140.0 : (p == 3) ? -160.0 : 160.0;
        } else {
            if (near) { E.rb.dyn_index += 1; if (E.rb.dyn_index > 1) E.rb.dyn_index = 0; }
            lvl = 100.0 * (b + 1);
            off = (E.rb.dyn_index == 0) ? -130.0 : 130.0;
        }
        double s, co;                    // rotateVector([lvl,0], leader.direction + off), misc.py:47-53
        sincos_bounded((ldir0 + off) * kDeg2Rad, s, co);
        tx = (double)lpx0 + co * lvl; ty = (double)lpy0 + s * lvl;
        E.rb.tgt_x = tx; E.rb.tgt_y = ty;
    }
    // steering of leader + bears (classes.py:184-215); the follower keeps the commands of step()
    bool steers = (lane == 0 && !E.leader_finished) || is_bear;
    if (steers) steer_to_point(E.rb, L, tx, ty, lane == 0, L.max_speed + 0);
    if (E.leader_finished) {                                   // ENV:1062-1065
        if (lane == 0) { command_forward(E.rb, L, 0); command_turn(E.rb, L, 0, 0); }
        i2 = FTL_LEADER_FINISHED;
    }
    // move(): every robot of the env in lock-step (the finished leader only receives commands)
    bool moves = (lane < P.R) && !(lane == 0 && E.leader_finished);
    robot_move(E.rb, L, moves);

    const float fpx = rl_f(E.rb.px, 1), fpy = rl_f(E.rb.py, 1);
    const int frx = rl_i(E.rb.rx, 1), fry = rl_i(E.rb.ry, 1), frw = rl_i(E.rb.rw, 1), frh = rl_i(E.rb.rh, 1);
    // follower collision (ENV:960-964, 1176-1194): statics, and the leader/bears where they were before moving
    if (!c.ignore_follower_collisions) {
        bool hit = false;
        for (int s = lane; s < c.n_static; s += FTL_WAVE) { int4 q = E.s_static[s]; hit |= rects_collide(frx, fry, frw, frh, q.x, q.y, q.z, q.w); }
        if (lane != 1 && lane < P.R) hit |= rects_collide(frx, fry, frw, frh, orx, ory, orw, orh);
        bool out = (double)fpx > (double)c.width || (double)fpy > (double)c.height || fpx < 0.0f || fpy < 0.0f;
        if (__ballot(hit) != 0ull || out) { E.crash = 1; E.done = 1; i0 = FTL_MISSION_FAIL; i1 = FTL_AGENT_CRASH; }
    }
    // green zone (ENV:968-969); a function of the trajectory only, so it is recomputed when a point was appended
    if (E.green_len != E.traj_len) { E.green_count = green_walk(E); E.green_len = E.traj_len; }
    const int G = E.green_count, n = E.traj_len;
    // _check_agent_position (ENV:1906-1937)
    if (G > 2) {
        const float2* tr = reinterpret_cast<const float2*>(traj_ptr(E));
        int id = closest_point(E, fpx, fpy, n - 2, -1, G);
        float2 q = tr[n - 2 - id];
        if (euclid_f32_le(fpx, fpy, q.x, q.y, c.leader_pos_epsilon)) { E.is_on_trace = 1; E.is_in_box = 1; }
        else if (euclid_f32_le(fpx, fpy, q.x, q.y, c.max_dev)) { E.is_in_box = 1; E.is_on_trace = 0; }
        else {
            int id2 = closest_point(E, fpx, fpy, 0, 1, n);
            float2 q2 = tr[id2];
            if (euclid_f32_le(fpx, fpy, q2.x, q2.y, c.leader_pos_epsilon)) { E.is_on_trace = 1; E.is_in_box = 0; }
        }
    }
    E.too_close = euclid_f32_le(lpx0, lpy0, fpx, fpy, c.min_distance);

    // leader collision (ENV:1068-1072): follower + statics, not the bears
    const float lpx = rl_f(E.rb.px, 0), lpy = rl_f(E.rb.py, 0);
    {
        const int lrx = rl_i(E.rb.rx, 0), lry = rl_i(E.rb.ry, 0), lrw = rl_i(E.rb.rw, 0), lrh = rl_i(E.rb.rh, 0);
        bool hit = false;
        for (int s = lane; s < c.n_static; s += FTL_WAVE) { int4 q = E.s_static[s]; hit |= rects_collide(lrx, lry, lrw, lrh, q.x, q.y, q.z, q.w); }
        if (lane == 1) hit |= rects_collide(lrx, lry, lrw, lrh, E.rb.rx, E.rb.ry, E.rb.rw, E.rb.rh);
        bool out = (double)lpx > (double)c.width || (double)lpy > (double)c.height || lpx < 0.0f || lpy < 0.0f;
        if (__ballot(hit) != 0ull || out) { E.done = 1; i0 = FTL_MISSION_FAIL; i2 = FTL_LEADER_CRASH; }
    }
    // ENV:1074-1075 with the deterministic tick: frame k (1-based since reset) sees get_ticks() == k
    if ((E.step_count + 1) % c.trajectory_saving_period == 0) {
        if (E.traj_len < c.traj_cap) {
            if (lane == 0) { float2* tr = reinterpret_cast<float2*>(traj_ptr(E)); tr[E.traj_len] = make_float2(lpx, lpy); }
            E.traj_len += 1;
            __syncthreads();          // the new point is read by other lanes from the next frame on
        } else E.error |= FTL_ERR_TRAJ_OVERFLOW;
    }
    if (E.leader_finished && E.is_in_box) {                 // ENV:1077-1087
        if (E.finish_timer < 0) E.finish_timer = 0;
        else {
            E.finish_timer += 1;
            if (E.finish_timer > c.frames_per_step * 20) { i0 = FTL_MISSION_SUCCESS; i2 = FTL_LEADER_FINISHED; i1 = FTL_AGENT_FINISHED; E.done = 1; }
        }
    }
    if (E.step_count > c.warm_start) {                      // ENV:1088-1107
        if (c.has_low_reward && E.acc_penalty < c.low_reward) { i0 = FTL_MISSION_FAIL; i2 = FTL_LEADER_MOVING; i1 = FTL_AGENT_LOW_REWARD; E.crash = 1; E.done = 1; }
        if (c.has_max_distance_coef) {
            float dx = fpx - lpx, dy = fpy - lpy;
            float nrm = sqrtf(dx * dx + dy * dy);
            if (nrm > (float)(c.max_distance * c.max_distance_coef)) { i0 = FTL_MISSION_FAIL; i2 = FTL_LEADER_MOVING; i1 = FTL_AGENT_TOO_FAR; E.crash = 1; E.done = 1; }
        }
    }
    // _reward_computation (ENV:1869-1904)
    double res = 0;
    res += c.leader_movement_reward;
    if (E.too_close) res += c.too_close_penalty;
    else {
        if (E.is_in_box && E.is_on_trace) res += c.reward_in_box;
        else if (E.is_in_box) res += c.reward_in_dev;
        else if (E.is_on_trace) res += c.reward_on_track;
        else if (E.step_count > c.warm_start) res += c.not_on_track_penalty;
    }
    if (E.crash) res += c.crash_penalty;
    if (res < 0) E.acc_penalty += res; else E.acc_penalty = 0;
    E.overall_reward += res;
    E.step_count += 1;
    if (E.step_count > c.max_steps) { i0 = FTL_MISSION_FINISHED_BY_TIME; i2 = FTL_LEADER_MOVING; i1 = FTL_AGENT_MOVING; E.done = 1; }
    reward = c.aggregate_reward ? E.overall_reward : res;
}

// ---- LeaderPositionsTracker_v2.scan (sensors.py:243-327) --------------------------------------------------------
__device__ __forceinline__ double* hist_slot(const FtlDevParams& P, int env, int abs_idx) {
    return P.hist + ((size_t)env * P.cfg.corr_cap + (abs_idx % P.cfg.corr_cap)) * 2;
}
__device__ __forceinline__ double* corr_slot(const FtlDevParams& P, int env, int abs_idx) {
    return P.corr + ((size_t)env * P.cfg.corr_cap + (abs_idx % P.cfg.corr_cap)) * 4;
}
// sensors.py:302-317: border pair from hist[i1]-hist[i0] anchored at hist[ia] (absolute indices), appended at corr index `at`
__device__ __forceinline__ void border_pair(EnvCtx& E, int i1, int i0, int ia, int at) {
    const double* p1 = hist_slot(E.P, E.env, i1); const double* p0 = hist_slot(E.P, E.env, i0); const double* a = hist_slot(E.P, E.env, ia);
    double vx, vy;
    if (i1 < E.seed_end || i0 < E.seed_end) {
        vx = p1[0] - p0[0]; vy = p1[1] - p0[1];
        double nrm = sqrt(__builtin_fma(vy, vy, vx * vx));
        double sc = E.P.cfg.corridor_width / nrm;
        vx *= sc; vy *= sc;
    } else {
        float fx = (float)p1[0] - (float)p0[0], fy = (float)p1[1] - (float)p0[1];
        float nrm = sqrtf(fx * fx + fy * fy);
        float sc = (float)E.P.cfg.corridor_width / nrm;
        fx *= sc; fy *= sc; vx = (double)fx; vy = (double)fy;
    }
    // cos/sin of +-90 deg as glibc rounds them (cos(pi/2 rounded) = 6.123233995736766e-17)
    const double c90 = 6.123233995736766e-17, s90 = 1.0, cm90 = 6.123233995736766e-17, sm90 = -1.0;
    double r0 = (c90 * vx + (-s90) * vy) + a[0], r1 = (s90 * vx + c90 * vy) + a[1];
    double l0 = (cm90 * vx + (-sm90) * vy) + a[0], l1 = (sm90 * vx + cm90 * vy) + a[1];
    if (E.lane == 0) { double* q = corr_slot(E.P, E.env, at); q[0] = r0; q[1] = r1; q[2] = l0; q[3] = l1; }
}
// np.sum(np.linalg.norm(diff(hist))) over the window [lo, hi) (sensors.py:288-290)
__device__ __forceinline__ double hist_path_length(EnvCtx& E, int lo, int hi) {
    int m = hi - lo;
    if (m < 2) return 0.0;
    bool any64 = lo < E.seed_end;
    __syncthreads();
    if (any64) {
        double* sd = reinterpret_cast<double*>(E.s_d);
        for (int i = E.lane; i < m - 1; i += FTL_WAVE) {
            const double* p = hist_slot(E.P, E.env, lo + i); const double* q = hist_slot(E.P, E.env, lo + i + 1);
            double dx = p[0] - q[0], dy = p[1] - q[1];
            sd[i] = sqrt(dx * dx + dy * dy);
        }
        __syncthreads();
        return pairwise_sum<double>(sd, m - 1);
    }
    float* sf = E.s_d;
    for (int i = E.lane; i < m - 1; i += FTL_WAVE) {
        const double* p = hist_slot(E.P, E.env, lo + i); const double* q = hist_slot(E.P, E.env, lo + i + 1);
        float dx = (float)p[0] - (float)q[0], dy = (float)p[1] - (float)q[1];
        sf[i] = sqrtf(dx * dx + dy * dy);
    }
    __syncthreads();
    return (double)pairwise_sum<float>(sf, m - 1);
}

__device__ __forceinline__ void tracker_scan(EnvCtx& E) {
    const ftl_config& c = E.P.cfg;
    const float lpx = rl_f(E.rb.px, 0), lpy = rl_f(E.rb.py, 0);
    if (E.trk_counter % c.tracker_saving_period == 0) {
        int len = E.corr_hi - E.corr_lo;
        if (len > 0) {
            const double* last = hist_slot(E.P, E.env, E.corr_hi - 1);
            if (last[0] == (double)lpx && last[1] == (double)lpy) return;     // sensors.py:247-251 (no counter increment)
        }
        bool first = (len == 0 && E.trk_counter == 0);
        if (first) {
            const float fpx = rl_f(E.rb.px, 1), fpy = rl_f(E.rb.py, 1);
            const double fdir = rl_d(E.rb.direction, 1);
            const double lmax = c.leader.max_speed;
            int n;
            if (c.tracker_start_behind) {                     // sensors.py:257-272 (float64 seed points)
                double s, co;
                sincos_bounded(angle_correction(fdir + 180.0) * kDeg2Rad, s, co);
                double sx = 50 * co + (double)fpx, sy = 50 * s + (double)fpy;
                double dist = euclid_f64(sx, sy, (double)lpx, (double)lpy);
                n = (int)(dist / ((double)(c.tracker_saving_period * 5) * lmax));
                if (n < 2 || n > c.corr_cap) { E.error |= (n < 2) ? FTL_ERR_TRACKER_SEED : FTL_ERR_CORR_OVERFLOW; E.trk_counter += 1; return; }
                double stepx = ((double)lpx - sx) / (n - 1), stepy = ((double)lpy - sy) / (n - 1);
                for (int i = E.lane; i < n; i += FTL_WAVE) {
                    double x = (stepx == 0) ? ((double)i / (n - 1)) * ((double)lpx - sx) + sx : (double)i * stepx + sx;
                    double y = (stepy == 0) ? ((double)i / (n - 1)) * ((double)lpy - sy) + sy : (double)i * stepy + sy;
                    if (i == n - 1) { x = (double)lpx; y = (double)lpy; }
                    double* h = hist_slot(E.P, E.env, i); h[0] = x; h[1] = y;
                }
                E.seed_end = n;
            } else {                                          // sensors.py:275-284 (np.linspace(f32,f32) is float32)
                double dist = euclid_f32(fpx, fpy, lpx, lpy);
                n = (int)(dist / ((double)(c.tracker_saving_period * 5) * lmax));
                if (n < 2 || n > c.corr_cap) { E.error |= (n < 2) ? FTL_ERR_TRACKER_SEED : FTL_ERR_CORR_OVERFLOW; E.trk_counter += 1; return; }
                float stepx = (lpx - fpx) / (float)(n - 1), stepy = (lpy - fpy) / (float)(n - 1);
                for (int i = E.lane; i < n; i += FTL_WAVE) {
                    float x = (stepx == 0) ? ((float)i / (float)(n - 1)) * (lpx - fpx) + fpx : (float)i * stepx + fpx;
                    float y = (stepy == 0) ? ((float)i / (float)(n - 1)) * (lpy - fpy) + fpy : (float)i * stepy + fpy;
                    if (i == n - 1) { x = lpx; y = lpy; }
                    double* h = hist_slot(E.P, E.env, i); h[0] = (double)x; h[1] = (double)y;
                }
                E.seed_end = 0;
            }
            E.corr_lo = 0; E.corr_hi = n;
        } else {                                              // sensors.py:286
            // the ring must still hold every point a stored snapshot refers to
            int oldest = E.corr_lo;
            const int* sw = E.P.snap_win + (size_t)E.env * E.P.hmax * 4;
            int nsnap = E.snap_count < E.P.hmax ? E.snap_count : E.P.hmax;
            for (int j = 0; j < nsnap; j++) { int l0 = sw[4 * j], l1 = sw[4 * j + 2]; oldest = min(oldest, min(l0, l1)); }
            if (E.corr_hi + 1 - oldest > c.corr_cap) { E.error |= FTL_ERR_CORR_OVERFLOW; E.trk_counter += 1; return; }
            if (E.lane == 0) { double* h = hist_slot(E.P, E.env, E.corr_hi); h[0] = (double)lpx; h[1] = (double)lpy; }
            E.corr_hi += 1;
        }
        __syncthreads();
        // sensors.py:288-297: drop the oldest points (and border pairs) while the polyline is longer than corridor_length
        double path = hist_path_length(E, E.corr_lo, E.corr_hi);
        while (path > c.corridor_length) {
            if (first) E.error |= FTL_ERR_TRACKER_SEED;        // reference: popleft on the still-empty corridor deque
            E.corr_lo += 1;
            path = hist_path_length(E, E.corr_lo, E.corr_hi);
        }
        int m = E.corr_hi - E.corr_lo;
        if (m > 1) {                                          // sensors.py:299-317
            if (first) {
                // i = m-1..1: vector hist[i]-hist[i-1], anchor hist[m-i-1]; pairs land at corridor index m-1-i
                for (int i = m - 1; i > 0; i--) border_pair(E, E.corr_lo + i, E.corr_lo + i - 1, E.corr_lo + m - i - 1, E.corr_lo + m - 1 - i);
            }
            border_pair(E, E.corr_hi - 1, E.corr_hi - 2, E.corr_hi - 2, E.corr_hi - 1);
        }
        __syncthreads();
    }
    E.trk_counter += 1;
}

// classes.py:255-288 minus the ray casts: both tracker scans, the bookkeeping of the ray sensors' history push
// (sensors.py:893-897: a sensor scans -- and pushes a snapshot -- only while len(corridor) > 1) and the error flag the
// reference would have raised.  The rays themselves are cast by ftl_rays_kernel from the snapshot ring.
__device__ __forceinline__ void sensors_bookkeeping(EnvCtx& E) {
    const FtlDevParams& P = E.P;
    const ftl_config& c = P.cfg;
    if (!c.has_tracker) { E.scan_ok = 0; return; }
    int groups = 0;       // bit g: some ray sensor is scanned in group g (0 = before the tracker's 2nd scan, 1 = after)
    for (int k = 0; k < c.n_lasers; k++) groups |= 1 << (c.lasers[k].after_tracker ? 1 : 0);
    int ok = 0, w0lo = 0, w0hi = 0;
#pragma nounroll
    for (int g = 0; g < 2; g++) {
        tracker_scan(E);
        if ((groups >> g) & 1) {
            if (E.corr_hi - E.corr_lo > 1) ok |= 1 << g;
            else E.error |= FTL_ERR_EMPTY_CORRIDOR;
        }
        if (g == 0) { w0lo = E.corr_lo; w0hi = E.corr_hi; }
    }
    E.scan_ok = ok;
    if (ok) {             // one snapshot per step: dynamic rects + the corridor window each group saw
        int slot = E.snap_count % P.hmax;
        int4* sr = reinterpret_cast<int4*>(P.snap_rects) + ((size_t)E.env * P.hmax + slot) * (P.R - 1);
        if (E.lane < P.R && E.lane != 1) sr[E.lane == 0 ? 0 : E.lane - 1] = make_int4(E.rb.rx, E.rb.ry, E.rb.rw, E.rb.rh);
        if (E.lane == 0) {
            int* sw = P.snap_win + ((size_t)E.env * P.hmax + slot) * 4;
            bool g0 = ok & 1;
            sw[0] = g0 ? w0lo : E.corr_lo; sw[1] = g0 ? w0hi : E.corr_hi; sw[2] = E.corr_lo; sw[3] = E.corr_hi;
        }
        E.snap_count += 1;
    }
}

// ENV:1789-1810
__device__ __forceinline__ void write_obs(EnvCtx& E) {
    float lp[5], fp[5];
    lp[0] = rl_f(E.rb.px, 0); lp[1] = rl_f(E.rb.py, 0); lp[2] = (float)rl_d(E.rb.speed, 0); lp[3] = (float)rl_d(E.rb.direction, 0); lp[4] = (float)rl_d(E.rb.rot_speed, 0);
    fp[0] = rl_f(E.rb.px, 1); fp[1] = rl_f(E.rb.py, 1); fp[2] = (float)rl_d(E.rb.speed, 1); fp[3] = (float)rl_d(E.rb.direction, 1); fp[4] = (float)rl_d(E.rb.rot_speed, 1);
    if (E.lane == 0) {
        float* o = E.C.out.obs_num + (size_t)E.env * FTL_OBS_NUM;
#pragma unroll
        for (int i = 0; i < 5; i++) { o[i] = lp[i]; o[5 + i] = fp[i]; }
        double tx = E.cur_tx, ty = E.cur_ty;
        if (E.route_len > 1) {
            const double* rt = route_ptr(E);
            if (tx == rt[2 * (E.route_len - 1)] && ty == rt[2 * (E.route_len - 1) + 1]) { tx = rt[2 * (E.route_len - 2)]; ty = rt[2 * (E.route_len - 2) + 1]; }
        }
        E.C.out.target[2 * (size_t)E.env] = tx; E.C.out.target[2 * (size_t)E.env + 1] = ty;
    }
}

// ---- LeaderCorridor_Prev_lasers_v2.scan (sensors.py:883-962): the rays ----------------------------------------------
// Segment table entry classes (sensors.py:644-660): which sensors see an entry is decided per class
enum { SEG_STATIC = 0, SEG_DYNAMIC = 1, SEG_CORRIDOR = 2, SEG_GREEN = 3, SEG_CLASSES = 4 };

// One obstacle segment A->B (float32, as stored by np.array(..., dtype=np.float32), sensors.py:672) against the ray
// origin C (float32) -> end E (float64).  Returns true on intersection and the SQUARED distance of the intersection
// point from the origin (sqrt is monotone: min over distances == sqrt of min over squared distances, so the sqrt of
// sensors.py:920-921 is taken once per output element instead of once per hit).
__device__ __forceinline__ bool hit_segment(float cx, float cy, double ex, double ey, float4 sg, double& d2) {
    const float ax = sg.x, ay = sg.y, bx = sg.z, by = sg.w;
    // ccw / intersect with the dtype flow of sensors.py:608-614
    float cax = cx - ax, cay = cy - ay, cbx = cx - bx, cby = cy - by, bax = bx - ax, bay = by - ay;
    double day = ey - (double)ay, dax = ex - (double)ax, dby = ey - (double)by, dbx = ex - (double)bx;
    bool t1 = day * (double)cax > (double)cay * dax;
    bool t2 = dby * (double)cbx > (double)cby * dbx;
    bool t3 = cay * bax > bay * cax;
    bool t4 = day * (double)bax > (double)bay * dax;
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

// ---------------------------------------------------------------- kernel 1: frames + tracker + observation
extern "C" __global__ void __launch_bounds__(FTL_WAVE, FTL_FRAMES_WPE) ftl_frames_kernel(const FtlDevParams* __restrict__ Pp, const FtlCall C) {
    extern __shared__ __align__(16) unsigned char lds[];
    using namespace ftl;
    const FtlDevParams& P = *Pp;
    const int env = blockIdx.x;
    if (env >= P.n_envs) return;
    EnvCtx E(P, C);
    E.env = env; E.lane = threadIdx.x;
    E.s_static = reinterpret_cast<int4*>(lds);
    E.s_d = reinterpret_cast<float*>(lds + (size_t)((P.cfg.n_static + 3) & ~3) * 16 + 16);
    E.scan_ok = 0;
    if (C.mode == 1) {                                   // reset(): ENV:434-543
        if (C.mask && !C.mask[env]) return;
        E.episodes = P.env_int[(size_t)env * FTL_EI_COUNT + FTL_EI_EPISODES];
        env_reset(E, C.scen_idx[env]);
        if (E.lane == 0) {
            C.out.reward[env] = 0.0; C.out.done[env] = (uint8_t)E.done;
            C.out.status[3 * (size_t)env] = 0; C.out.status[3 * (size_t)env + 1] = 0; C.out.status[3 * (size_t)env + 2] = 0;
        }
    } else {                                             // step(action): ENV:908-945
        env_load(E);
        __syncthreads();
        const Limits L = lane_limits(P.cfg, E.lane);
        {
            double a0 = C.action[2 * (size_t)env], a1 = C.action[2 * (size_t)env + 1];
            if (E.lane == 1) {
                command_forward(E.rb, L, a0);                                   // ENV:927
                if (a1 < 0) command_turn(E.rb, L, fabs(a1), -1);                // ENV:928-933
                else if (a1 > 0) command_turn(E.rb, L, a1, 1);
                else command_turn(E.rb, L, 0, 0);
            }
        }
        double reward = 0; int i0 = 0, i1 = 0, i2 = 0;
#pragma nounroll
        for (int f = 0; f < P.cfg.frames_per_step; f++) frame_step(E, L, reward, i0, i1, i2);   // ENV:935-936
        if (E.lane == 0) {
            C.out.reward[env] = reward; C.out.done[env] = (uint8_t)E.done;
            C.out.status[3 * (size_t)env] = (uint8_t)i0; C.out.status[3 * (size_t)env + 1] = (uint8_t)i1; C.out.status[3 * (size_t)env + 2] = (uint8_t)i2;
        }
        if (E.done && (C.flags & FTL_STEP_AUTO_RESET)) {
            // vector-env convention: terminal reward/done/status are kept, the observation is the first one of the
            // next episode (the terminal sensor scan would be discarded, so it is not run)
            E.episodes += 1;
            int next = (E.scen + P.n_envs) % P.scen.n_scenarios;
            __syncthreads();
            env_reset(E, next);
        }
    }
    sensors_bookkeeping(E);                              // ENV:937 / ENV:541 (tracker part of use_sensors)
    write_obs(E);                                        // ENV:938
    env_store(E);
}

// ---------------------------------------------------------------- kernel 2: the ray casts
// Phase 1 (lane-parallel): every obstacle segment any sensor of this env could see -- static rect edges, the leader /
//   bear rect edges of the last H snapshots, the corridor polylines and green-zone caps of those snapshots -- is tested
//   against the sensors' reach around the follower and the survivors are compacted into a per-class segment table in
//   LDS (segment f32x4 + bit mask of the snapshots that contain it).  A segment wholly outside the reach box cannot
//   intersect any ray, so dropping it is exact.
// Phase 2 (per sensor): the 64 lanes are split rays x chunks (12 rays x 5 chunks, 24 x 2, ...): a lane walks every
//   nch-th table entry of the classes its sensor reacts to, keeping one nearest-hit accumulator per snapshot; chunk
//   results are min-combined through shuffles and lane (ray, chunk 0) writes the H rows of its ray.
extern "C" __global__ void __launch_bounds__(FTL_WAVE, FTL_RAYS_WPE) ftl_rays_kernel(const FtlDevParams* __restrict__ Pp, const FtlCall C) {
    extern __shared__ __align__(16) unsigned char lds[];
    using namespace ftl;
    const FtlDevParams& P = *Pp;
    const ftl_config& c = P.cfg;
    const int env = blockIdx.x;
    if (env >= P.n_envs) return;
    if (C.mode == 1 && C.mask && !C.mask[env]) return;
    const int lane = threadIdx.x;
    const int hmax = P.hmax;
    const int nrect_dyn = P.R - 1;      // leader + bears per snapshot
    // LDS: f32 corridor ring | segment table (float4) | masks (u32) | class counters
    const int cap_static = 4 * c.n_static + 4 * hmax, cap_dyn = 4 * hmax * (P.R - 2 > 0 ? P.R - 2 : 0) + 4;
    const int cap_corr = 2 * c.corr_cap, cap_green = 2 * hmax;
    const int off1 = cap_static, off2 = off1 + cap_dyn, off3 = off2 + cap_corr, off4 = off3 + cap_green;
    auto cls_off = [&](int q) { return q == 0 ? 0 : (q == 1 ? off1 : (q == 2 ? off2 : off3)); };
    float4* s_corr = reinterpret_cast<float4*>(lds);
    float4* s_seg = s_corr + c.corr_cap;
    unsigned* s_mask = reinterpret_cast<unsigned*>(s_seg + off4);
    int* s_cnt = reinterpret_cast<int*>(s_mask + off4);

    const int* ei = P.env_int + (size_t)env * FTL_EI_COUNT;
    const int scen = ei[FTL_EI_SCEN], snap_count = ei[FTL_EI_SNAP_COUNT], scan_ok = ei[FTL_EI_SCAN_OK];
    const size_t fo = (size_t)env * P.R + 1;            // follower
    const float cx = P.rb_pos[2 * fo], cy = P.rb_pos[2 * fo + 1];
    const double fdir = P.rb_dbl[fo * FTL_RD_COUNT + FTL_RD_DIRECTION];
    float* out_base = C.out.lasers + (size_t)env * P.lasers_len;
    const int nsnap = snap_count < hmax ? snap_count : hmax;       // valid snapshots, newest = snap_count-1
    const unsigned all_snaps = (1u << nsnap) - 1u;                  // bit a = age a (nsnap <= 8)

#pragma nounroll
    for (int which = 0; which < 2; which++) {
        int n_sens = 0; float lmax = 0.0f;
        for (int k = 0; k < c.n_lasers; k++) if (c.lasers[k].after_tracker == which) { n_sens++; lmax = fmaxf(lmax, (float)c.lasers[k].length); }
        if (n_sens == 0) continue;
        if (!((scan_ok >> which) & 1)) {       // sensors.py:893/962: the reference raises UnboundLocalError here
            for (int k = 0; k < c.n_lasers; k++) if (c.lasers[k].after_tracker == which)
                for (int i = lane; i < c.lasers[k].history * c.lasers[k].count; i += FTL_WAVE) out_base[c.lasers[k].out_offset + i] = (float)c.lasers[k].length;
            continue;
        }
        // corridor windows of the valid snapshots as this group of sensors saw them; age a = 0 newest
        int win_lo[FTL_HMAX], win_hi[FTL_HMAX];
        int umin = 0x7fffffff, umax = 0;
        {
            const int* sw = P.snap_win + (size_t)env * hmax * 4;
#pragma unroll
            for (int a = 0; a < FTL_HMAX; a++) {
                win_lo[a] = 0; win_hi[a] = 0;
                if (a < nsnap) {
                    int slot = (snap_count - 1 - a) % hmax;
                    win_lo[a] = sw[4 * slot + 2 * which]; win_hi[a] = sw[4 * slot + 2 * which + 1];
                    umin = min(umin, win_lo[a]); umax = max(umax, win_hi[a]);
                }
            }
        }
        __syncthreads();
        if (lane < SEG_CLASSES) s_cnt[lane] = 0;
        for (int p = umin + lane; p < umax; p += FTL_WAVE) {
            const double* q = corr_slot(P, env, p);
            s_corr[p % c.corr_cap] = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
        }
        __syncthreads();

        // ---- phase 1: culled, compacted segment table ------------------------------------------------------------------
        const float reach = lmax + 2.0f;
        const float bx0 = cx - reach, bx1 = cx + reach, by0 = cy - reach, by1 = cy + reach;
        auto push_rect = [&](int cls, int4 q, unsigned sm) {           // 4 edges in the order of sensors.py:668-671
            if ((float)(q.x + q.z) < bx0 || (float)q.x > bx1 || (float)(q.y + q.w) < by0 || (float)q.y > by1) return;
            int at = cls_off(cls) + atomicAdd(&s_cnt[cls], 4);
            float l = (float)q.x, t = (float)q.y, r = (float)(q.x + q.z), b = (float)(q.y + q.w);
            s_seg[at] = make_float4(l, b, r, b); s_seg[at + 1] = make_float4(r, t, r, b);
            s_seg[at + 2] = make_float4(r, t, l, t); s_seg[at + 3] = make_float4(l, b, l, t);
            s_mask[at] = sm; s_mask[at + 1] = sm; s_mask[at + 2] = sm; s_mask[at + 3] = sm;
        };
        auto push_seg = [&](int cls, float ax, float ay, float bx, float by, unsigned sm, bool cull) {
            if (cull && (fmaxf(ax, bx) < bx0 || fminf(ax, bx) > bx1 || fmaxf(ay, by) < by0 || fminf(ay, by) > by1)) return;
            int at = cls_off(cls) + atomicAdd(&s_cnt[cls], 1);
            s_seg[at] = make_float4(ax, ay, bx, by); s_mask[at] = sm;
        };
        {   // static rects straight from the scenario pool: identical in every snapshot
            const int4* src = reinterpret_cast<const int4*>(P.scen.static_rects) + (size_t)scen * c.n_static;
            for (int s = lane; s < c.n_static; s += FTL_WAVE) push_rect(SEG_STATIC, src[s], all_snaps);
        }
        {   // leader (a static-class object: it sits in game_object_list) and bears, per snapshot
            const int4* sr = reinterpret_cast<const int4*>(P.snap_rects) + (size_t)env * hmax * nrect_dyn;
            for (int i = lane; i < nsnap * nrect_dyn; i += FTL_WAVE) {
                int a = i / nrect_dyn, o = i - a * nrect_dyn;
                int slot = (snap_count - 1 - a) % hmax;
                push_rect(o == 0 ? SEG_STATIC : SEG_DYNAMIC, sr[slot * nrect_dyn + o], 1u << a);
            }
        }
        // corridor polylines: segment p -> p+1 belongs to every snapshot whose window holds both points
        for (int p = umin + lane; p + 1 < umax; p += FTL_WAVE) {
            unsigned sm = 0;
#pragma unroll
            for (int a = 0; a < FTL_HMAX; a++) if (a < nsnap && win_lo[a] <= p && p + 1 < win_hi[a]) sm |= 1u << a;
            if (!sm) continue;
            float4 u = s_corr[p % c.corr_cap], v = s_corr[(p + 1) % c.corr_cap];
            push_seg(SEG_CORRIDOR, u.x, u.y, v.x, v.y, sm, true);      // right border
            push_seg(SEG_CORRIDOR, u.z, u.w, v.z, v.w, sm, true);      // left border
        }
        // green-zone end caps of every snapshot (sensors.py:648-650)
#pragma unroll
        for (int a = 0; a < FTL_HMAX; a++) if (a < nsnap && lane == a) {
            float4 u = s_corr[win_lo[a] % c.corr_cap], v = s_corr[(win_hi[a] - 1) % c.corr_cap];
            push_seg(SEG_GREEN, u.x, u.y, u.z, u.w, 1u << a, true);
            push_seg(SEG_GREEN, v.x, v.y, v.z, v.w, 1u << a, true);
        }
        __syncthreads();

        // ---- phase 2: rays x chunks per sensor -------------------------------------------------------------------------
#pragma nounroll
        for (int k = 0; k < c.n_lasers; k++) {
            if (c.lasers[k].after_tracker != which) continue;
            const int N = c.lasers[k].count, H = c.lasers[k].history, ooff = c.lasers[k].out_offset;
            const double len = c.lasers[k].length, aoff = c.lasers[k].angle_offset, period = 360.0 / (double)N;
            const int ro = c.lasers[k].react_obstacles;
            unsigned cls_on = 0;          // sensors.py:644-660
            if (ro == 1 || ro == 2) cls_on |= 1u << SEG_STATIC;
            if (ro == 1 || ro == 3) cls_on |= 1u << SEG_DYNAMIC;
            if (c.lasers[k].react_corridor) cls_on |= 1u << SEG_CORRIDOR;
            if (c.lasers[k].react_green) cls_on |= 1u << SEG_GREEN;
            for (int base = 0; base < N; base += FTL_WAVE) {
                const int rp = min(FTL_WAVE, N - base);           // rays in this pass
                const int nch = FTL_WAVE / rp;                   // chunks per ray
                const int ray = base + lane % rp, chunk = lane / rp;
                const bool active = chunk < nch;
                double s, co;
                sincos_bounded(((fdir + aoff) + ray * period) * kDeg2Rad, s, co);    // sensors.py:888-891
                const double ex = (double)cx + co * len, ey = (double)cy + s * len;
                double best[FTL_HMAX];
#pragma unroll
                for (int j = 0; j < FTL_HMAX; j++) best[j] = 1.0e300;
#pragma nounroll
                for (int q = 0; q < SEG_CLASSES; q++) {
                    if (!((cls_on >> q) & 1u)) continue;
                    const int beg = cls_off(q), end = beg + s_cnt[q];
                    for (int m0 = beg; m0 < end; m0 += nch) {
                        const int m = m0 + chunk;
                        double d2;
                        if (active && m < end && hit_segment(cx, cy, ex, ey, s_seg[m], d2)) {
                            const unsigned sm = s_mask[m];
#pragma unroll
                            for (int j = 0; j < FTL_HMAX; j++) if (((sm >> j) & 1u) && d2 < best[j]) best[j] = d2;
                        }
                    }
                }
                // combine the chunks of each ray (lanes ray, ray+rp, ray+2rp, ...)
                for (int ch = 1; ch < nch; ch++) {
#pragma unroll
                    for (int j = 0; j < FTL_HMAX; j++) {
                        double o = __shfl(best[j], (lane % rp) + ch * rp);
                        if (o < best[j]) best[j] = o;
                    }
                }
                if (chunk == 0) {
                    // reading when nothing is hit: |end - origin| (sensors.py:925-930); rows: oldest first, newest last
                    double qx0 = ex - (double)cx, qy0 = ey - (double)cy;
                    const double miss = sqrt(__builtin_fma(qy0, qy0, qx0 * qx0));
#pragma unroll
                    for (int a = 0; a < FTL_HMAX; a++) {
                        if (a < H) {
                            double v = (a < nsnap && best[a] < 1.0e299) ? sqrt(best[a]) : miss;
                            out_base[ooff + (H - 1 - a) * N + ray] = (float)v;
                        }
                    }
                }
            }
        }
    }
}

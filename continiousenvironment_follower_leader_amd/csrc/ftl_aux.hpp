// ftl_aux.hpp -- the sensors of the registry (utils/sensors.py:1291-1307) that are not f32 segment ray casts, and the deprecated
// v1 tracker.  They run in kernels of their own, launched only when the config has them, so that the two hot kernels carry none
// of this code:
//
//   ftl_tracker1_kernel -- LeaderPositionsTracker (sensors.py:148-229), one THREAD per env, between the frame kernel and the ray
//       kernel: append / corridor pair / eat_close_points, then the snapshot bookkeeping the frame kernel does for the v2 tracker.
//   ftl_aux_kernel -- one wavefront per env, after the ray kernel:
//       * LeaderCorridor_lasers_compas (sensors.py:1138-1288): corridor walls in float64, nearest wall's orientation selects the block;
//       * LaserSensor (sensors.py:18-145): point-in-rect marching;
//       * LeaderTrackDetector_vector / _radar (sensors.py:342-487) on the tracker's position history.
// Numerics follow oracle/ftl_oracle.c (compas_scan, lidar_scan, track_vector_scan, track_radar_scan, tracker1_scan).
#pragma once
#include "ftl_device.hpp"

namespace ftl {

// ---------------------------------------------------------------- v1 tracker
__global__ void __launch_bounds__(256) ftl_tracker1_kernel(const FtlDevParams* __restrict__ Pp, const FtlCall C) {
    const FtlDevParams& P = *Pp;
    const ftl_config& c = P.cfg;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= P.n_envs) return;
    if (C.mode == 1 && C.mask && !C.mask[env]) return;
    int* ei = rec_field(P.env_int, P, env);
    int counter = ei[FTL_EI_TRK_COUNTER], hlen = ei[FTL_EI_HIST1_LEN], clen = ei[FTL_EI_CORR_HI], err = 0;
    const float* rpos = rec_field(P.rb_pos, P, env);
    const float lpx = rpos[0], lpy = rpos[1];
    const float fpx = rpos[2], fpy = rpos[3];
    float2* h = reinterpret_cast<float2*>(P.hist1) + (size_t)env * c.hist1_cap;
    bool unchanged = false;
    if (counter % c.tracker_saving_period == 0) {
        if (hlen > 0 && h[hlen - 1].x == lpx && h[hlen - 1].y == lpy) unchanged = true;     // SEN:178-183: returns before anything else
        else {
            if (hlen + 2 > c.hist1_cap) err |= FTL_ERR_HIST1_OVERFLOW;
            else {
                if (hlen == 0) h[hlen++] = make_float2(fpx, fpy);                           // SEN:185-186
                h[hlen++] = make_float2(lpx, lpy);                                          // SEN:187
                if (hlen > 1) {                                                             // SEN:188-195, half-width env.max_dev
                    if (clen + 1 > c.corr_cap) err |= FTL_ERR_CORR_OVERFLOW;
                    else {
                        const float2 p1 = h[hlen - 1], p0 = h[hlen - 2];
                        float fx = p1.x - p0.x, fy = p1.y - p0.y;
                        const float nrm = sqrtf(fx * fx + fy * fy);
                        const float sc = (float)c.corridor_width / nrm;
                        fx *= sc; fy *= sc;
                        const double vx = (double)fx, vy = (double)fy;
                        const double c90 = 6.123233995736766e-17, s90 = 1.0, cm90 = 6.123233995736766e-17, sm90 = -1.0;
                        double* q = corr_slot(P, env, clen);
                        q[0] = (c90 * vx + (-s90) * vy) + (double)p0.x; q[1] = (s90 * vx + c90 * vy) + (double)p0.y;
                        q[2] = (cm90 * vx + (-sm90) * vy) + (double)p0.x; q[3] = (sm90 * vx + cm90 * vy) + (double)p0.y;
                        *corr32_slot(P, env, clen) = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
                        clen += 1;
                    }
                }
            }
        }
    }
    if (!unchanged) {
        counter += 1;
        if (c.trk1_eat_close_points && hlen > 0) {                                          // SEN:211-216
            const float thr = (float)c.trk1_eat_radius;
            int w = 0;
            for (int i = 0; i < hlen; i++) {
                const float2 p = h[i];
                const float dx = p.x - fpx, dy = p.y - fpy;
                if (sqrtf(dx * dx + dy * dy) <= thr) continue;
                if (w != i) h[w] = p;
                w++;
            }
            hlen = w;
        }
    }
    // what the frame kernel's g_sensors() does after the v2 scans: may the ray sensors scan, and one snapshot per step
    int groups = 0, strict = 0;
    for (int k = 0; k < c.n_lasers; k++) { groups |= 1; if (!c.lasers[k].lenient) strict |= 1; }
    int ok = 0;
    if (groups) { if (clen > 1) ok = 1; else if (strict) err |= FTL_ERR_EMPTY_CORRIDOR; }
    if (ok) {
        const int slot = ei[FTL_EI_SNAP_HEAD];
        int4* sr = reinterpret_cast<int4*>(rec_field(P.snap_rects, P, env)) + slot * (P.R - 1);
        for (int r = 0; r < P.R; r++) if (r != 1) sr[r == 0 ? 0 : r - 1] = reinterpret_cast<const int4*>(rec_field(P.rb_int, P, env) + r * FTL_RI_COUNT)[0];
        int* sw = rec_field(P.snap_win, P, env) + slot * 4;
        sw[0] = 0; sw[1] = clen; sw[2] = 0; sw[3] = clen;
        ei[FTL_EI_SNAP_COUNT] += 1;
        ei[FTL_EI_SNAP_HEAD] = (slot + 1 == P.hmax) ? 0 : slot + 1;
    }
    ei[FTL_EI_TRK_COUNTER] = counter; ei[FTL_EI_HIST1_LEN] = hlen; ei[FTL_EI_CORR_LO] = 0; ei[FTL_EI_CORR_HI] = clen; ei[FTL_EI_SCAN_OK] = ok;
    if (err) { ei[FTL_EI_ERROR] |= err; ei[FTL_EI_ERROR_STICKY] |= err; }
}

// ---------------------------------------------------------------- compas: one wall against one ray, every operand float64
__device__ __forceinline__ bool wall_hit(double ax, double ay, double bx, double by, double cx, double cy, double ex, double ey, double& d2) {
    const bool t1 = (ey - ay) * (cx - ax) > (cy - ay) * (ex - ax);   // ccw(A,C,D), sensors.py:608-614
    const bool t2 = (ey - by) * (cx - bx) > (cy - by) * (ex - bx);   // ccw(B,C,D)
    const bool t3 = (cy - ay) * (bx - ax) > (by - ay) * (cx - ax);   // ccw(A,B,C)
    const bool t4 = (ey - ay) * (bx - ax) > (by - ay) * (ex - ax);   // ccw(A,B,D)
    if (!((t1 != t2) && (t3 != t4))) return false;
    const double dax = bx - ax, day = by - ay, dbx = ex - cx, dby = ey - cy, dpx = ax - cx, dpy = ay - cy;      // seg_intersect, 626-640
    const double dapx = -day, dapy = dax;
    const double t = (dapx * dpx + dapy * dpy) / (dapx * dbx + dapy * dby);
    const double x = t * dbx + cx, y = t * dby + cy;
    const double qx = x - cx, qy = y - cy;
    d2 = qx * qx + qy * qy;
    return true;
}

}  // namespace ftl

// One wavefront per env.  LDS: one scratch area shared by the sensors, used one after the other; its size is the largest any sensor of
// the config needs (ftl_aux_lds_bytes below, passed at launch).
#define FTL_COMPAS_STAGE 128         // corridor points the compas sensors stage in LDS (a longer span is read from global memory)
#define FTL_LIDAR_RECTS 128          // objects in range of one lidar (more raise FTL_ERR_LIDAR_OVERFLOW and are ignored)
#ifndef FTL_AUX_WPE
#define FTL_AUX_WPE 5
#endif
__global__ void __launch_bounds__(FTL_WAVE, FTL_AUX_WPE) ftl_aux_kernel(const FtlDevParams* __restrict__ Pp, const FtlCall C) {
    using namespace ftl;
    extern __shared__ __align__(16) unsigned char lds[];
    const FtlDevParams& P = *Pp;
    const ftl_config& c = P.cfg;
    const int env = blockIdx.x, lane = threadIdx.x;
    if (env >= P.n_envs) return;
    if (C.mode == 1 && C.mask && !C.mask[env]) return;
    const int* ei = rec_field(P.env_int, P, env);
    const float cxf = rec_field(P.rb_pos, P, env)[2], cyf = rec_field(P.rb_pos, P, env)[3];        // robot 1 = the follower
    const double cx = (double)cxf, cy = (double)cyf;
    const double fdir = rec_field(P.rb_dbl, P, env)[1 * FTL_RD_COUNT + FTL_RD_DIRECTION];
    float* out_base = C.out.lasers + (size_t)env * P.lasers_len;
    const unsigned long long kInf = 0x7fefffffffffffffull;

    // ---------------- LeaderCorridor_lasers_compas ------------------------------------------------------------------------------
    // Everything the compas sensors read from global memory arrives in three round trips, whatever their number: the env scalars, the
    // corridor windows of all snapshot slots (both scan passes), then the corridor points those windows span, staged as float64 in
    // LDS.  The sensors themselves work on LDS only (this kernel is bound by its wavefronts' serial latency, not by arithmetic).
    bool any_compas = false;
    for (int k = 0; k < c.n_lasers; k++) any_compas = any_compas || c.lasers[k].compas != 0;
    if (any_compas) {
        const int snap_count = ei[FTL_EI_SNAP_COUNT], snap_head = ei[FTL_EI_SNAP_HEAD], scan_ok = ei[FTL_EI_SCAN_OK];
        const int newest = (snap_head == 0 ? P.hmax : snap_head) - 1;
        int* s_winall = reinterpret_cast<int*>(lds);                                            // [hmax][4]: lo, hi of pass 0, lo, hi of pass 1
        double* s_c64 = reinterpret_cast<double*>(lds + (((size_t)P.hmax * 16 + 31) & ~(size_t)31));   // [FTL_COMPAS_STAGE][4] corridor points
        unsigned char* scratch = reinterpret_cast<unsigned char*>(s_c64 + 4 * FTL_COMPAS_STAGE);
        const int nvalid = snap_count < P.hmax ? snap_count : P.hmax;
        if (lane < P.hmax * 4) s_winall[lane] = rec_field(P.snap_win, P, env)[lane];
        __syncthreads();
        int passes = 0;                                   // scan passes (before / after the tracker's own dict entry) that have a compas sensor
        for (int k = 0; k < c.n_lasers; k++) if (c.lasers[k].compas) passes |= 1 << (c.lasers[k].after_tracker ? 1 : 0);
        int base = 0x7fffffff, top = 0;
        for (int a = 0; a < nvalid; a++) {
            int slot = newest - a; slot += slot < 0 ? P.hmax : 0;
            for (int w = 0; w < 2; w++) if ((passes >> w) & 1) { base = min(base, s_winall[4 * slot + 2 * w]); top = max(top, s_winall[4 * slot + 2 * w + 1]); }
        }
        const bool staged = nvalid > 0 && top - base <= FTL_COMPAS_STAGE;      // (the v1 tracker's corridor is never trimmed: it can outgrow the stage)
        if (staged) for (int p = base + lane; p < top; p += FTL_WAVE) {
            const double4 q = *reinterpret_cast<const double4*>(corr_slot(P, env, p));
            *reinterpret_cast<double4*>(s_c64 + 4 * (p - base)) = q;
        }
        __syncthreads();
        auto corr_pt = [&](int p) -> const double* { return staged ? s_c64 + 4 * (p - base) : corr_slot(P, env, p); };
        for (int k = 0; k < c.n_lasers; k++) {
            const ftl_laser_cfg& Lc = c.lasers[k];
            if (!Lc.compas) continue;
            const int N = Lc.count, H = Lc.history, W = 5 * N, which = Lc.after_tracker;
            float* out = out_base + Lc.out_offset;
            float* pol = (C.out.policy_obs && P.pol_off[k] >= 0) ? C.out.policy_obs + (size_t)env * P.pol_h * P.pol_width + P.pol_off[k] : nullptr;
            const float flen = (float)Lc.length;
            if (!((scan_ok >> which) & 1)) {      // SEN:1192/1244: the reference raises UnboundLocalError (error bit set by the tracker code)
                for (int i = lane; i < H * W; i += FTL_WAVE) {
                    const float v = (i % W) < N ? flen : 0.0f;
                    out[i] = v;
                    if (pol) pol[(i / W) * P.pol_width + (i % W)] = fminf(fmaxf(v / flen, 0.0f), 1.0f);
                }
                continue;
            }
            double2* s_ray = reinterpret_cast<double2*>(scratch);                               // [N]
            unsigned long long* s_best = reinterpret_cast<unsigned long long*>(s_ray + N);      // [N][H]: squared distance, wall class in the 2 low bits
            int* s_win = reinterpret_cast<int*>(s_best + N * H);                                // [H][2]
            __syncthreads();
            const int nsnap = snap_count < H ? snap_count : H;
            if (lane < H) {
                int lo = 0, hi = 0;
                if (lane < nsnap) {
                    int slot = newest - lane; slot += slot < 0 ? P.hmax : 0;
                    lo = s_winall[4 * slot + 2 * which]; hi = s_winall[4 * slot + 2 * which + 1];
                }
                s_win[2 * lane] = lo; s_win[2 * lane + 1] = hi;
            }
            for (int i = lane; i < N; i += FTL_WAVE) {
                double s, co;
                sincos_bounded(((fdir + Lc.angle_offset) + i * (360.0 / (double)N)) * kDeg2Rad, s, co);
                s_ray[i] = make_double2(cx + co * Lc.length, cy + s * Lc.length);
            }
            for (int i = lane; i < N * H; i += FTL_WAVE) s_best[i] = kInf;
            __syncthreads();
            int umin = 0x7fffffff, umax = 0;
            for (int a = 0; a < nsnap; a++) { umin = min(umin, s_win[2 * a]); umax = max(umax, s_win[2 * a + 1]); }
            const int n_wall = nsnap > 0 ? 2 * max(umax - umin - 1, 0) : 0;
            const int n_items = n_wall + 2 * nsnap;
            // one item per lane -- a wall shared by every snapshot whose window holds both of its points, or an end cap of ONE snapshot --
            // against its candidate rays (the wall stays in registers; N <= 36)
            for (int item = lane; item < n_items; item += FTL_WAVE) {
                double ax, ay, bx, by; unsigned sm = 0; int cls;
                if (item < n_wall) {
                    const int p = umin + (item >> 1), side = item & 1;             // side 0: right wall (class 3), 1: left wall (class 2)
                    for (int a = 0; a < nsnap; a++) if (s_win[2 * a] <= p && p + 1 < s_win[2 * a + 1]) sm |= 1u << a;
                    const double* u = corr_pt(p); const double* v = corr_pt(p + 1);
                    ax = u[2 * side]; ay = u[2 * side + 1]; bx = v[2 * side]; by = v[2 * side + 1];
                    cls = side ? 2 : 3;
                } else {
                    const int a = (item - n_wall) >> 1, back = (item - n_wall) & 1;   // front = corridor[-1] (class 0), back = corridor[0] (class 1)
                    const double* u = corr_pt(back ? s_win[2 * a] : s_win[2 * a + 1] - 1);
                    ax = u[0]; ay = u[1]; bx = u[2]; by = u[3];
                    sm = 1u << a; cls = back;
                }
                if (!sm) continue;
                // Candidate rays, as in phase 3 of ftl_rays_kernel (float32 throughout: this only selects which rays get the reference's
                // float64 test).  A wall whose closest approach is beyond the rays' reach (2 px of slack) cannot be hit; otherwise only the
                // rays whose direction lies inside the arc the wall subtends at the follower (polynomial atan2, widened by >= 0.01 rad, two
                // orders of magnitude above its error) can cross it -- typically 1-3 of the N.  A wall next to the follower keeps every ray.
                int i0 = 0, cnt = N;
                {
                    const float axf = (float)(ax - cx), ayf = (float)(ay - cy), bxf = (float)(bx - cx), byf = (float)(by - cy);
                    const float ux = bxf - axf, uy = byf - ayf, l2 = __builtin_fmaf(ux, ux, uy * uy);
                    const float tt = l2 > 0.0f ? fminf(fmaxf(__fdividef(-__builtin_fmaf(axf, ux, ayf * uy), l2), 0.0f), 1.0f) : 0.0f;
                    const float nx = __builtin_fmaf(tt, ux, axf), ny = __builtin_fmaf(tt, uy, ayf);
                    const float dmin2 = __builtin_fmaf(nx, nx, ny * ny), reach = flen + 2.0f;
                    if (dmin2 > reach * reach) continue;
                    const float fN = (float)N, inv_step = fN * 0.15915494309189535f;
                    const float phis = (float)((fdir + Lc.angle_offset) * kDeg2Rad) * inv_step;
                    float uA = __builtin_fmaf(arc_atan2(ayf, axf), inv_step, -phis), uB = __builtin_fmaf(arc_atan2(byf, bxf), inv_step, -phis);
                    const float invN = __fdividef(1.0f, fN);
                    uA = __builtin_fmaf(-floorf(uA * invN), fN, uA); uB = __builtin_fmaf(-floorf(uB * invN), fN, uB);   // into [0, N) (an ulp outside is absorbed by the wrap below)
                    float diff = uB - uA; if (diff < 0.0f) diff += fN;
                    float start = uA, wd = diff;
                    if (diff > 0.5f * fN) { start = uB; wd = fN - diff; }
                    if (!(dmin2 < 4.0f || wd > 0.5f * fN - 0.05f)) {
                        const float slack = 0.02f + 0.01f * inv_step;
                        i0 = (int)ceilf(start - slack);
                        cnt = (int)floorf(start + wd + slack) - i0 + 1;
                        cnt = cnt > N ? N : cnt;
                    }
                }
                for (int t = 0; t < cnt; t++) {
                    int ray = i0 + t; ray = ray < 0 ? ray + N : (ray >= N ? ray - N : ray);
                    const double2 e = s_ray[ray];
                    double d2;
                    if (wall_hit(ax, ay, bx, by, cx, cy, e.x, e.y, d2)) {
                        // np.argmin takes the FIRST of equal minima in the order front, back, left walls, right walls (SEN:1166): the class code
                        // rides in the two low mantissa bits, so equal distances resolve the same way (3 ulp of float64 before the float32 store)
                        const unsigned long long key = ((unsigned long long)__double_as_longlong(d2) & ~3ull) | (unsigned long long)cls;
                        for (int a = 0; a < nsnap; a++) if ((sm >> a) & 1u) atomicMin(&s_best[ray * H + a], key);
                    }
                }
            }
            __syncthreads();
            // rows, oldest first (SEN:1247): block 0 = rays without a wall hit (|end - position|), then front / back / left / right
            for (int i = lane; i < H * W; i += FTL_WAVE) { out[i] = 0.0f; if (pol) pol[(i / W) * P.pol_width + (i % W)] = 0.0f; }
            __syncthreads();
            for (int w = lane; w < N * H; w += FTL_WAVE) {
                const int ray = w / H, a = w - ray * H;
                const unsigned long long key = s_best[w];
                const double2 e = s_ray[ray];
                int col; double v;
                if (a < nsnap && key != kInf) { col = (1 + (int)(key & 3ull)) * N + ray; v = sqrt(__longlong_as_double((long long)(key & ~3ull))); }
                else { const double qx = e.x - cx, qy = e.y - cy; col = ray; v = sqrt(__builtin_fma(qy, qy, qx * qx)); }
                const float vf = (float)v;
                out[(H - 1 - a) * W + col] = vf;
                if (pol) pol[(H - 1 - a) * P.pol_width + col] = fminf(fmaxf(vf / flen, 0.0f), 1.0f);
            }
            __syncthreads();
        }
    }

    // ---------------- LaserSensor / LeaderTrackDetector_* --------------------------------------------------------------------------
    for (int j = 0; j < c.n_aux; j++) {
        const ftl_aux_cfg& A = c.aux[j];
        float* out = out_base + A.out_offset;
        __syncthreads();
        if (A.kind == FTL_AUX_LIDAR) {
            const int n_ang = A.n_angles, npts = A.points_number;
            float4* s_rect = reinterpret_cast<float4*>(lds);                                      // [FTL_LIDAR_RECTS] rects in range: x0, x1, y0, y1 as float32
            float2* s_end = reinterpret_cast<float2*>(lds + 16 * FTL_LIDAR_RECTS);                // [n_angles] ray ends
            int* s_first = reinterpret_cast<int*>(lds + 16 * FTL_LIDAR_RECTS + 8 * n_ang);        // [n_angles] first marching point inside a rect
            unsigned long long* s_cand = reinterpret_cast<unsigned long long*>(s_first + n_ang + (n_ang & 1));   // [n_angles] rects whose box the ray's box overlaps
            float2* s_u = reinterpret_cast<float2*>(s_cand + n_ang);                              // [points_number] float32(u), float32(1 - u) of marching point i
            int* s_list = reinterpret_cast<int*>(s_u + npts);                                     // [n_angles] rays with a non-empty candidate mask
            int* s_cnt = s_list + n_ang;                                                          // [2] rects in range, rays to march
            if (lane < 2) s_cnt[lane] = 0;
            __syncthreads();
            // objects_in_range (SEN:72-79): leader, static rects, bears whose nearest corner / edge mid-point is within range + 3 m
            const int nobj = 1 + c.n_static + c.n_bears;
            const int scen = ei[FTL_EI_SCEN];
            for (int o = lane; o < nobj; o += FTL_WAVE) {
                int4 q;
                if (o == 0) q = reinterpret_cast<const int4*>(rec_field(P.rb_int, P, env))[0];
                else if (o <= c.n_static) q = reinterpret_cast<const int4*>(P.scen.static_rects)[(size_t)scen * c.n_static + (o - 1)];
                else q = reinterpret_cast<const int4*>(rec_field(P.rb_int, P, env) + (2 + (o - 1 - c.n_static)) * FTL_RI_COUNT)[0];
                const int px[8] = { q.x, q.x, q.x + q.z, q.x + q.z, q.x + (q.z >> 1), q.x, q.x + (q.z >> 1), q.x + q.z };
                const int py[8] = { q.y, q.y + q.w, q.y, q.y + q.w, q.y, q.y + (q.w >> 1), q.y + q.w, q.y + (q.w >> 1) };
                // min over the eight points of sqrt(d2) <= range  <=>  sqrt(min d2) <= range (sqrt and its rounding are monotone); away from
                // the threshold the squared values decide, inside a band wider than every rounding involved the square root is taken
                double m2 = 1.0e300;
#pragma unroll
                for (int t = 0; t < 8; t++) { const double dx = cx - (double)px[t], dy = cy - (double)py[t]; m2 = fmin(m2, dx * dx + dy * dy); }
                const double t2 = A.in_range_px * A.in_range_px;
                const bool in = m2 < t2 * (1.0 - 1e-12) ? true : (m2 > t2 * (1.0 + 1e-12) ? false : sqrt(m2) <= A.in_range_px);
                if (in) {
                    const int at = atomicAdd(s_cnt, 1);
                    if (at < FTL_LIDAR_RECTS) s_rect[at] = make_float4((float)q.x, (float)(q.x + q.z), (float)q.y, (float)(q.y + q.w));
                }
            }
            for (int a = lane; a < n_ang; a += FTL_WAVE) {                         // SEN:88-104
                double angle = -fdir;
                if (a > 0) { const double kk = (double)((a + 1) / 2) * A.angle_step; angle = angle_correction((a & 1) ? -fdir + kk : -fdir - kk); }
                double s, co;
                sincos_bounded(angle * kDeg2Rad, s, co);
                s_end[a] = make_float2(cxf + (float)(A.range_px * co), cyf - (float)(A.range_px * s));      // np.float32 + python float -> float32
                s_first[a] = 0x7fffffff;
            }
            for (int i = lane; i < npts; i += FTL_WAVE) {                          // np.linspace weights of SEN:106-110
                const double u = (double)i / (double)npts;
                s_u[i] = make_float2((float)u, (float)(1.0 - u));
            }
            __syncthreads();
            if (lane == 0 && *s_cnt > FTL_LIDAR_RECTS) {
                int* eiw = rec_field(P.env_int, P, env);
                atomicOr(&eiw[FTL_EI_ERROR], (int)FTL_ERR_LIDAR_OVERFLOW); atomicOr(&eiw[FTL_EI_ERROR_STICKY], (int)FTL_ERR_LIDAR_OVERFLOW);
            }
            const int nin = min(*s_cnt, FTL_LIDAR_RECTS);
            // Every marching point of a ray lies between the follower and the ray's end (a float32 convex combination: within 1e-3 px of
            // the segment), so only the rects whose box comes within 1 px of the ray's box can contain one: a bit mask per ray (the first
            // 64 rects in range; a 65th and later ones are tested for every point).
            for (int a = lane; a < n_ang; a += FTL_WAVE) {
                const float2 e = s_end[a];
                const float x0 = fminf(e.x, cxf) - 1.0f, x1 = fmaxf(e.x, cxf) + 1.0f, y0 = fminf(e.y, cyf) - 1.0f, y1 = fmaxf(e.y, cyf) + 1.0f;
                unsigned long long mk = 0ull;
                for (int o = 0; o < nin && o < 64; o++) {
                    const float4 q = s_rect[o];
                    if (q.x <= x1 && q.y >= x0 && q.z <= y1 && q.w >= y0) mk |= 1ull << o;
                }
                s_cand[a] = mk;
                if (mk || nin > 64) s_list[atomicAdd(s_cnt + 1, 1)] = a;                       // only these rays can hit anything: the others are not marched
            }
            __syncthreads();
            const float inv_npts = 1.0f / (float)npts;
            const int n_march = s_cnt[1] * npts;
            for (int w = lane; w < n_march; w += FTL_WAVE) {                       // SEN:106-121
                const int la = (int)(((float)w + 0.5f) * inv_npts);                // w / npts (w < 2^16: the float quotient is off by far less than half a step)
                const int i = w - la * npts, a = s_list[la];
                const float2 e = s_end[a], uu = s_u[i];
                const float px = e.x * uu.x + cxf * uu.y, py = e.y * uu.x + cyf * uu.y;
                bool hit = false;
                unsigned long long mk = s_cand[a];
                while (mk) {
                    const int o = __ffsll((long long)mk) - 1; mk &= mk - 1;
                    const float4 q = s_rect[o];
                    hit = hit || (q.x <= px && px < q.y && q.z <= py && py < q.w);
                }
                for (int o = 64; o < nin; o++) {
                    const float4 q = s_rect[o];
                    hit = hit || (q.x <= px && px < q.y && q.z <= py && py < q.w);
                }
                if (hit) atomicMin(&s_first[a], i);
            }
            __syncthreads();
            if (A.return_all_points) {
                // SEN:112-113, 131-134: every marching point of a ray up to and including its first hit (all of them without a hit), rays in
                // order: [count K][K points or distances][zeros].  Offsets of the rays by a running sum (a few hundred rays at most: lane 0),
                // the points themselves -- the same float32 convex combinations the march tests -- by the whole wavefront.
                const int wd = A.return_only_distances ? 1 : 2;
                int* s_off = s_list;                                               // (the list of marched rays is not needed any more)
                __syncthreads();
                if (lane == 0) {
                    int k = 0;
                    for (int a = 0; a < n_ang; a++) { s_off[a] = k; k += s_first[a] != 0x7fffffff ? s_first[a] + 1 : npts; }
                    s_cnt[0] = k;
                }
                __syncthreads();
                const int K = s_cnt[0];
                if (lane == 0) out[0] = (float)K;
                for (int w = lane; w < n_ang * npts; w += FTL_WAVE) {
                    const int a = (int)(((float)w + 0.5f) * inv_npts), i = w - a * npts;
                    const int cnt_a = s_first[a] != 0x7fffffff ? s_first[a] + 1 : npts;
                    if (i < cnt_a) {
                        const float2 e = s_end[a], uu = s_u[i];
                        const float dx = (e.x * uu.x + cxf * uu.y) - cxf, dy = (e.y * uu.x + cyf * uu.y) - cyf;
                        const int at = s_off[a] + i;
                        if (wd == 1) out[1 + at] = sqrtf(dx * dx + dy * dy); else { out[1 + 2 * at] = dx; out[2 + 2 * at] = dy; }
                    }
                }
                for (int w = 1 + K * wd + lane; w < 1 + n_ang * npts * wd; w += FTL_WAVE) out[w] = 0.0f;
            } else
            for (int a = lane; a < n_ang; a += FTL_WAVE) {
                const float2 e = s_end[a];
                float px = e.x, py = e.y;
                if (s_first[a] != 0x7fffffff) {
                    const double u = (double)s_first[a] / (double)A.points_number;
                    px = e.x * (float)u + cxf * (float)(1.0 - u); py = e.y * (float)u + cyf * (float)(1.0 - u);
                }
                const float dx = px - cxf, dy = py - cyf;
                if (A.return_only_distances) out[a] = sqrtf(dx * dx + dy * dy);
                else { out[2 * a] = dx; out[2 * a + 1] = dy; }
            }
        } else {
            // the tracked leader positions the detector looks at: v1 = the float32 "hist1" list, v2 = a window of the "hist" ring as the
            // detector's dict position saw it; points of the ring below seed_end are float64, the others float32 values
            const bool v1 = c.has_tracker == 1;
            int lo, hi;
            if (v1) { lo = 0; hi = ei[FTL_EI_HIST1_LEN]; }
            else if (A.after_tracker) { lo = ei[FTL_EI_CORR_LO]; hi = ei[FTL_EI_CORR_HI]; }
            else { lo = ei[FTL_EI_HW0_LO]; hi = ei[FTL_EI_HW0_HI]; }
            const int n = hi - lo;
            int s0 = 0, s1 = n;                                                    // slice of [0, n)
            if (A.detectable == 0) s0 = n - A.seq_len > 0 ? n - A.seq_len : 0;     // "new": the last seq_len
            else if (A.detectable == 1) s1 = n < A.seq_len ? n : A.seq_len;        // "old": the first seq_len
            auto point = [&](int i, double& hx, double& hy) {
                if (v1) { const float2 q = reinterpret_cast<const float2*>(P.hist1)[(size_t)env * c.hist1_cap + i]; hx = (double)q.x; hy = (double)q.y; }
                else { const double* q = hist_slot(P, env, lo + i); hx = q[0]; hy = q[1]; }
            };
            if (A.kind == FTL_AUX_TRACK_VECTOR) {                                  // SEN:362-381
                for (int i = lane; i < A.seq_len; i += FTL_WAVE) {
                    float vx = 0.0f, vy = 0.0f;
                    if (s0 + i < s1) { double hx, hy; point(s0 + i, hx, hy); vx = (float)(hx - cx); vy = (float)(hy - cy); }
                    out[2 * i] = vx; out[2 * i + 1] = vy;
                }
            } else {                                                               // SEN:423-476
                unsigned* s_rad = reinterpret_cast<unsigned*>(lds);                // [sectors] float bits of the nearest distance (0x7f800000: none)
                const int S = A.radar_sectors;
                for (int s = lane; s < S; s += FTL_WAVE) s_rad[s] = 0x7f800000u;
                __syncthreads();
                double sd, cd, sr, cr;
                sincos_bounded(fdir * kDeg2Rad, sd, cd);
                double rdir = fdir + 90; if (rdir >= 360) rdir -= 360;
                sincos_bounded(rdir * kDeg2Rad, sr, cr);
                const double dvx = cd * 1 + (-sd) * 0, dvy = sd * 1 + cd * 0, rvx = cr * 1 + (-sr) * 0, rvy = sr * 1 + cr * 0;
                const double nd = sqrt(__builtin_fma(dvy, dvy, dvx * dvx)), nr = sqrt(__builtin_fma(rvy, rvy, rvx * rvx));
                const bool any64 = !v1 && (lo + s0) < ei[FTL_EI_SEED_END] && s0 < s1;   // a float64 point in the slice makes the whole array float64
                const double sa = 3.141592653589793 / (double)A.radar_sectors;
                const float inv_sa = (float)((double)A.radar_sectors / 3.141592653589793);
                for (int i = s0 + lane; i < s1; i += FTL_WAVE) {
                    double hx, hy; point(i, hx, hy);
                    double vx, vy, dist;
                    if (any64) { vx = hx - cx; vy = hy - cy; dist = sqrt(vx * vx + vy * vy); }
                    else { const float fx = (float)hx - cxf, fy = (float)hy - cyf; vx = (double)fx; vy = (double)fy; dist = (double)sqrtf(fx * fx + fy * fy); }
                    const double dot_d = vx * dvx + vy * dvy, dot_r = vx * rvx + vy * rvy;
                    // The sector is floor(angle to the right-hand vector / sa) for points ahead of the follower (SEN:463-476).  A float32
                    // atan2 of (|cross|, dot) gives that angle to ~3e-7 rad (the reference's own arccos(dot / norms) is within 2e-8 of it);
                    // unless it lies within 4e-6 rad of a sector boundary, or the point sits within 1e-9 of the follower's side-to-side
                    // axis (where "ahead" is decided), the sector is settled without the two float64 arccos of the reference expression.
                    const float ang = atan2f(fabsf((float)(vx * rvy - vy * rvx)), (float)dot_r);
                    const float qf = ang * inv_sa;
                    const float tf = floorf(qf);
                    const bool settled = dist > 0.0 && fabs(dot_d) > 1e-9 * dist && (qf - tf) > 4e-6f * inv_sa && (tf + 1.0f - qf) > 4e-6f * inv_sa;
                    if (settled) {
                        const int t = (int)tf;
                        if (dot_d > 0.0 && t < S) atomicMin(&s_rad[t], __float_as_uint((float)dist));     // behind the follower: ar < 0, no sector
                        continue;
                    }
                    const double ad = acos(dot_d / (dist * nd));
                    double ar = acos(dot_r / (dist * nr));
                    if (ad > 3.141592653589793 / 2) ar = -ar;
                    int s = (int)floor(ar / sa);                                   // candidate sector; the reference's own comparisons decide
                    for (int t = s - 1; t <= s + 1; t++)
                        if (t >= 0 && t < S && ar >= sa * t && ar < sa * (t + 1)) atomicMin(&s_rad[t], __float_as_uint((float)dist));
                }
                __syncthreads();
                for (int s = lane; s < S; s += FTL_WAVE) out[s] = s_rad[s] == 0x7f800000u ? 0.0f : __uint_as_float(s_rad[s]);
            }
        }
    }
}

// dynamic LDS of ftl_aux_kernel for a config
static inline size_t ftl_aux_lds_bytes(const ftl_config& c) {
    size_t need = 64;
    int hmax = 1;
    for (int k = 0; k < c.n_lasers; k++) hmax = c.lasers[k].history > hmax ? c.lasers[k].history : hmax;     // = FtlDevParams::hmax
    for (int k = 0; k < c.n_lasers; k++) if (c.lasers[k].compas) {
        const size_t b = (((size_t)hmax * 16 + 31) & ~(size_t)31) + (size_t)32 * FTL_COMPAS_STAGE
                       + (size_t)c.lasers[k].count * 16 + (size_t)c.lasers[k].count * c.lasers[k].history * 8 + (size_t)c.lasers[k].history * 8;
        need = b > need ? b : need;
    }
    for (int j = 0; j < c.n_aux; j++) {
        const ftl_aux_cfg& a = c.aux[j];
        const size_t b = a.kind == FTL_AUX_LIDAR ? (size_t)16 * FTL_LIDAR_RECTS + (size_t)24 * a.n_angles + (size_t)8 * a.points_number + 32 : a.kind == FTL_AUX_TRACK_RADAR ? (size_t)4 * a.radar_sectors : 0;
        need = b > need ? b : need;
    }
    return (need + 15) & ~(size_t)15;
}

// ftl_gazebo.hpp -- follower-relative ("Gazebo") tracker + ray sensors (include/ftl_gazebo.h): device kernel and
// the C-ABI entry points.  Included at the end of ftl_abi.hip (same translation unit: it shares fail() / ftl_last_error()).
// Reference: src/arctic_gym/gazebo_utils/gazebo_tracker.py ("GZ"); numerics follow oracle/ftl_oracle_gazebo.c line by line.
//
// One wavefront per robot.  The tracker scan (GZ:17-172) is a short sequential program over <= 64 history points: lane 0 runs it on
// an LDS copy of the history / corridor.  The new obstacle-line snapshot (GZ:180-200) is built by all lanes into the sensor's ring of
// float32 segment lists -- explicit lists, unlike the 2-D env: the whole corridor shifts by the follower's displacement every call,
// so snapshots share nothing.  Ray casting: (segment, ray) pairs of every snapshot strided over the lanes, nearest squared distance
// per (ray, snapshot) by a 64-bit LDS atomic min.  Dtype flow: every difference with the follower position (the Python list [0, 0],
// an int array) is float64, B - A of a float32 segment stays float32 (oracle header).
#pragma once
#include "../../include/ftl_gazebo.h"
#include "ftl_device.hpp"

struct FtlGzParams {
    ftl_gz_config cfg;
    int32_t n_envs, lasers_len, max_seg;
    int32_t off[FTL_GZ_MAX_LASERS];
    int32_t* gz_int;          // [n][8]: counter, hist_len, corr_len, error, (snap_count, head) per sensor
    double* gz_hist;          // [n][FTL_GZ_HIST_CAP][2]
    double* gz_corr;          // [n][FTL_GZ_HIST_CAP][4]
    float4* seg[FTL_GZ_MAX_LASERS];      // [n][history][max_seg]
    int32_t* seg_n[FTL_GZ_MAX_LASERS];   // [n][history]
};
struct FtlGzCall {
    const double* leader; const double* yaw; const double* delta; const double* pts1; const double* pts2; const int32_t* n_pts;
    const uint8_t* mask; float* lasers; int32_t mode;     // mode 0 = step, 1 = reset
};

namespace ftl {

__device__ __forceinline__ double gz_norm1d(double x, double y) { return sqrt(__builtin_fma(y, y, x * x)); }
__device__ double gz_path_length(const double (*h)[2], int n) {            // GZ:130-132: np.sum of the m = n-1 consecutive distances (numpy pairwise, m <= 63)
    const int m = n - 1;
    if (m < 1) return 0.0;
    auto d = [&](int i) { const double dx = h[i][0] - h[i + 1][0], dy = h[i][1] - h[i + 1][1]; return sqrt(dx * dx + dy * dy); };
    if (m < 8) { double r = 0.0; for (int i = 0; i < m; i++) r += d(i); return r; }
    double r[8]; int i;
    for (int j = 0; j < 8; j++) r[j] = d(j);
    for (i = 8; i < m - (m % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += d(i + j);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < m; i++) res += d(i);
    return res;
}

}  // namespace ftl

__global__ void __launch_bounds__(FTL_WAVE) ftl_gz_kernel(const FtlGzParams P, const FtlGzCall C) {
    using namespace ftl;
    __shared__ double s_hist[FTL_GZ_HIST_CAP][2];
    __shared__ double s_corr[FTL_GZ_HIST_CAP][4];
    __shared__ int s_int[8];
    __shared__ double2 s_ray[36];
    __shared__ unsigned long long s_best[36 * FTL_HMAX];
    __shared__ int s_cnt;
    const ftl_gz_config& c = P.cfg;
    const int env = blockIdx.x, lane = threadIdx.x;
    if (env >= P.n_envs) return;
    if (C.mode == 1) {                       // tracker.reset() + laser.reset(), SEN:223-226, 964-968
        if (C.mask && !C.mask[env]) return;
        if (lane < 8) P.gz_int[(size_t)env * 8 + lane] = 0;
        return;
    }
    int* gi = P.gz_int + (size_t)env * 8;
    if (lane < 8) s_int[lane] = gi[lane];
    for (int i = lane; i < FTL_GZ_HIST_CAP; i += FTL_WAVE) {
        s_hist[i][0] = P.gz_hist[((size_t)env * FTL_GZ_HIST_CAP + i) * 2]; s_hist[i][1] = P.gz_hist[((size_t)env * FTL_GZ_HIST_CAP + i) * 2 + 1];
        for (int q = 0; q < 4; q++) s_corr[i][q] = P.gz_corr[((size_t)env * FTL_GZ_HIST_CAP + i) * 4 + q];
    }
    __syncthreads();
    const double yaw = C.yaw[env] * kRad2Deg;            // np.degrees(follower_orientation)[2]
    // ---- GazeboLeaderPositionsTracker_v2.scan, GZ:17-172 (lane 0)
    if (lane == 0) {
        int counter = s_int[0], n = s_int[1], m = s_int[2], err = s_int[3];
        const double lx = C.leader[2 * (size_t)env], ly = C.leader[2 * (size_t)env + 1];
        const double dx = C.delta[2 * (size_t)env], dy = C.delta[2 * (size_t)env + 1];
        for (int i = 0; i < n; i++) { s_hist[i][0] = rint((s_hist[i][0] - dx) * 1e5) / 1e5; s_hist[i][1] = rint((s_hist[i][1] - dy) * 1e5) / 1e5; }   // GZ:46-55
        for (int i = 0; i < m; i++) { s_corr[i][0] -= dx; s_corr[i][1] -= dy; s_corr[i][2] -= dx; s_corr[i][3] -= dy; }                                // GZ:58-79
        bool count = true;
        if (counter % 3 == 0) {
            if (n > 0 && gz_norm1d(lx - s_hist[n - 1][0], ly - s_hist[n - 1][1]) < 1) count = false;       // GZ:101-106
            else {
                if (n == 0 && counter == 0) {                                                             // GZ:108-123
                    double s, co;
                    sincos_bounded(angle_correction(yaw + 180) * kDeg2Rad, s, co);
                    const double sx = 10 * co + 0, sy = 10 * s + 0;
                    const int k = 10;
                    const double stepx = (lx - sx) / (k - 1), stepy = (ly - sy) / (k - 1);
                    for (int i = 0; i < k; i++) {
                        double x = (stepx == 0) ? ((double)i / (k - 1)) * (lx - sx) + sx : (double)i * stepx + sx;
                        double y = (stepy == 0) ? ((double)i / (k - 1)) * (ly - sy) + sy : (double)i * stepy + sy;
                        if (i == k - 1) { x = lx; y = ly; }
                        s_hist[i][0] = x; s_hist[i][1] = y;
                    }
                    n = k;
                } else if (n == 0) { err |= FTL_ERR_TRACKER_SEED; }                                       // reference: IndexError on hist[-1]
                else {                                                                                    // GZ:124-130
                    const double last = gz_norm1d(s_hist[n - 1][0] - 0, s_hist[n - 1][1] - 0), cur = gz_norm1d(lx - 0, ly - 0);
                    if (cur > last && cur < 25) {
                        if (n >= FTL_GZ_HIST_CAP) err |= FTL_ERR_CORR_OVERFLOW;
                        else { s_hist[n][0] = lx; s_hist[n][1] = ly; n++; }
                    }
                }
                if (n > 0) {
                    double path = gz_path_length(s_hist, n);                                              // GZ:133-144
                    while (path > 25) {
                        if (n > 0) { for (int i = 0; i + 1 < n; i++) { s_hist[i][0] = s_hist[i + 1][0]; s_hist[i][1] = s_hist[i + 1][1]; } n--; }
                        if (m > 0) { for (int i = 0; i + 1 < m; i++) for (int q = 0; q < 4; q++) s_corr[i][q] = s_corr[i + 1][q]; m--; }
                        path = gz_path_length(s_hist, n);
                    }
                    if (n > 1) {                                                                          // GZ:147-163
                        auto pair = [&](int i1, int i0, int ia) {
                            double vx = s_hist[i1][0] - s_hist[i0][0], vy = s_hist[i1][1] - s_hist[i0][1];
                            const double sc = 2 / gz_norm1d(vx, vy);
                            vx *= sc; vy *= sc;
                            const double c90 = 6.123233995736766e-17, s90 = 1.0, cm90 = 6.123233995736766e-17, sm90 = -1.0;
                            if (m >= FTL_GZ_HIST_CAP) { err |= FTL_ERR_CORR_OVERFLOW; return; }
                            s_corr[m][0] = (c90 * vx + (-s90) * vy) + s_hist[ia][0]; s_corr[m][1] = (s90 * vx + c90 * vy) + s_hist[ia][1];
                            s_corr[m][2] = (cm90 * vx + (-sm90) * vy) + s_hist[ia][0]; s_corr[m][3] = (sm90 * vx + cm90 * vy) + s_hist[ia][1];
                            m++;
                        };
                        if (counter == 0) for (int i = n - 1; i > 0; i--) pair(i, i - 1, n - i - 1);
                        pair(n - 1, n - 2, n - 2);
                    }
                }
            }
        }
        if (count) counter += 1;
        s_int[0] = counter; s_int[1] = n; s_int[2] = m; s_int[3] = err;
    }
    __syncthreads();
    for (int i = lane; i < FTL_GZ_HIST_CAP; i += FTL_WAVE) {
        P.gz_hist[((size_t)env * FTL_GZ_HIST_CAP + i) * 2] = s_hist[i][0]; P.gz_hist[((size_t)env * FTL_GZ_HIST_CAP + i) * 2 + 1] = s_hist[i][1];
        for (int q = 0; q < 4; q++) P.gz_corr[((size_t)env * FTL_GZ_HIST_CAP + i) * 4 + q] = s_corr[i][q];
    }
    const int Cn = s_int[2];
    const int n_pts = min(max(C.n_pts[env], 0), c.max_pts);
    const double* p1 = C.pts1 + (size_t)env * c.max_pts * 2; const double* p2 = C.pts2 + (size_t)env * c.max_pts * 2;
    const unsigned long long kInf = 0x7fefffffffffffffull;
    // ---- GazeboCorridor_Prev_lasers_v2.scan, GZ:203-297
    for (int k = 0; k < c.n_lasers; k++) {
        const ftl_gz_laser_cfg& L = c.lasers[k];
        const int N = L.count, H = L.history, W = L.pad_sectors ? 4 * N : N;
        float* out = C.lasers + (size_t)env * P.lasers_len + P.off[k];
        if (Cn <= 1) {                       // GZ:216/297: UnboundLocalError in the reference
            for (int i = lane; i < H * W; i += FTL_WAVE) out[i] = (float)L.length;
            if (lane == 0) s_int[3] |= FTL_ERR_EMPTY_CORRIDOR;
            continue;
        }
        int cnt = s_int[4 + 2 * k], head = s_int[5 + 2 * k];
        float4* ring = P.seg[k] + (size_t)env * H * P.max_seg;
        int* ring_n = P.seg_n[k] + (size_t)env * H;
        {   // collect_obstacle_edges (GZ:180-200) -> ring slot `head`; float32 cast of GZ:198
            float4* dst = ring + (size_t)head * P.max_seg;
            int base = 0;
            if (L.react_corridor) {
                for (int i = lane; i < Cn - 1; i += FTL_WAVE) {
                    dst[2 * i] = make_float4((float)s_corr[i][0], (float)s_corr[i][1], (float)s_corr[i + 1][0], (float)s_corr[i + 1][1]);
                    dst[2 * i + 1] = make_float4((float)s_corr[i][2], (float)s_corr[i][3], (float)s_corr[i + 1][2], (float)s_corr[i + 1][3]);
                }
                base = 2 * (Cn - 1);
            }
            if (L.react_green) {
                if (lane == 0) dst[base] = make_float4((float)s_corr[0][0], (float)s_corr[0][1], (float)s_corr[0][2], (float)s_corr[0][3]);
                if (lane == 1) dst[base + 1] = make_float4((float)s_corr[Cn - 1][0], (float)s_corr[Cn - 1][1], (float)s_corr[Cn - 1][2], (float)s_corr[Cn - 1][3]);
                base += 2;
            }
            if (L.react_obstacles) {
                for (int i = lane; i < n_pts - 1; i += FTL_WAVE) {
                    const double ax = p1[2 * i], ay = p1[2 * i + 1], bx = p1[2 * i + 2], by = p1[2 * i + 3];
                    if (gz_norm1d(ax - bx, ay - by) < 0.5) dst[base + i] = make_float4((float)ax, (float)ay, (float)bx, (float)by);
                    else dst[base + i] = make_float4((float)ax, (float)ay, (float)p2[2 * i], (float)p2[2 * i + 1]);
                }
                base += max(n_pts - 1, 0);
            }
            if (lane == 0) ring_n[head] = base;
        }
        cnt += 1; const int newest = head; head = (head + 1 == H) ? 0 : head + 1;
        if (lane == 0) { s_int[4 + 2 * k] = cnt; s_int[5 + 2 * k] = head; }
        for (int i = lane; i < N; i += FTL_WAVE) {                                   // GZ:209-212
            double s, co;
            sincos_bounded(((yaw - 45) + i * (360.0 / (double)N)) * kDeg2Rad, s, co);
            s_ray[i] = make_double2(co * L.length, s * L.length);
        }
        for (int i = lane; i < N * H; i += FTL_WAVE) s_best[i] = kInf;
        __syncthreads();                                                             // also makes this wave's ring stores visible to its loads below
        const int nsnap = cnt < H ? cnt : H;
        for (int a = 0; a < nsnap; a++) {                                            // age 0 = newest
            int slot = newest - a; slot += slot < 0 ? H : 0;
            const float4* sg = ring + (size_t)slot * P.max_seg;
            const int ns = a == 0 ? ((L.react_corridor ? 2 * (Cn - 1) : 0) + (L.react_green ? 2 : 0) + (L.react_obstacles ? max(n_pts - 1, 0) : 0)) : ring_n[slot];
            // One segment per lane against its candidate rays only, as in phase 3 of ftl_rays_kernel: the rays whose direction lies inside
            // the arc the segment subtends at the origin (the follower), widened by >= 0.01 rad (polynomial float32 atan2, error 2e-5 rad),
            // are the only ones that can cross it; a segment through / next to the origin (within 1 % of the rays' length) or beyond the rays' reach keeps all / none.
            const float fN = (float)N, inv_step = fN * 0.15915494309189535f, invN = 1.0f / fN;
            const float phis = (float)((yaw - 45.0) * kDeg2Rad) * inv_step, slack = 0.02f + 0.01f * inv_step;
            const float reach = (float)L.length * 1.01f, near2 = 1e-4f * reach * reach;     // (metres here, not pixels: relative margins)
            for (int si = lane; si < ns; si += FTL_WAVE) {
                const float4 q = sg[si];
                int i0 = 0, nc = N;                       // candidate rays i0 .. i0 + nc - 1 (mod N)
                {
                    const float ex_ = q.z - q.x, ey_ = q.w - q.y;
                    const float l2 = __builtin_fmaf(ex_, ex_, ey_ * ey_);
                    const float tt = l2 > 0.0f ? fminf(fmaxf(__fdividef(-__builtin_fmaf(q.x, ex_, q.y * ey_), l2), 0.0f), 1.0f) : 0.0f;
                    const float nx = __builtin_fmaf(tt, ex_, q.x), ny = __builtin_fmaf(tt, ey_, q.y);
                    const float dmin2 = __builtin_fmaf(nx, nx, ny * ny);
                    if (dmin2 > reach * reach) continue;
                    float uA = __builtin_fmaf(arc_atan2(q.y, q.x), inv_step, -phis), uB = __builtin_fmaf(arc_atan2(q.w, q.z), inv_step, -phis);
                    uA = __builtin_fmaf(-floorf(uA * invN), fN, uA); uB = __builtin_fmaf(-floorf(uB * invN), fN, uB);
                    float diff = uB - uA; if (diff < 0.0f) diff += fN;
                    float start = uA, wd = diff;
                    if (diff > 0.5f * fN) { start = uB; wd = fN - diff; }
                    if (!(dmin2 < near2 || wd > 0.5f * fN - 0.05f)) {
                        i0 = (int)ceilf(start - slack);
                        nc = (int)floorf(start + wd + slack) - i0 + 1;
                        nc = nc > N ? N : nc;
                    }
                }
                const double ax = q.x, ay = q.y, bx = q.z, by = q.w;
                const float bax = q.z - q.x, bay = q.w - q.y;
                for (int t = 0; t < nc; t++) {
                    int ray = i0 + t; ray = ray < 0 ? ray + N : (ray >= N ? ray - N : ray);
                    const double2 e = s_ray[ray];
                    const bool t1 = (e.y - ay) * (0 - ax) > (0 - ay) * (e.x - ax);       // ccw(A,C,D), SEN:608-614 with C = [[0, 0]] (int)
                    const bool t2 = (e.y - by) * (0 - bx) > (0 - by) * (e.x - bx);       // ccw(B,C,D)
                    const bool t3 = (0 - ay) * (double)bax > (double)bay * (0 - ax);     // ccw(A,B,C)
                    const bool t4 = (e.y - ay) * (double)bax > (double)bay * (e.x - ax); // ccw(A,B,D)
                    if (!((t1 != t2) && (t3 != t4))) continue;
                    const double dapx = (double)(-bay), dapy = (double)bax;              // seg_intersect, SEN:626-640
                    const double tq = (dapx * ax + dapy * ay) / (dapx * e.x + dapy * e.y);
                    const double x = tq * e.x + 0, y = tq * e.y + 0;
                    atomicMin(&s_best[ray * H + a], (unsigned long long)__double_as_longlong(x * x + y * y));
                }
            }
        }
        __syncthreads();
        if (L.pad_sectors) { for (int i = lane; i < H * W; i += FTL_WAVE) out[i] = 0.0f; __syncthreads(); }
        for (int w = lane; w < N * H; w += FTL_WAVE) {                               // rows: oldest first (GZ:222, 288)
            const int ray = w / H, a = w - ray * H;
            const unsigned long long key = s_best[w];
            const double2 e = s_ray[ray];
            const double v = (a < nsnap && key != kInf) ? sqrt(__longlong_as_double((long long)key)) : gz_norm1d(e.x, e.y);
            int col = ray;
            if (L.pad_sectors) { const double lis = (double)N / 4.0, di = (double)ray; col = (di < lis ? 0 : (di < 2 * lis ? 1 : (di < 3 * lis ? 2 : 3))) * N + ray; }
            out[(H - 1 - a) * W + col] = (float)v;
        }
        __syncthreads();
    }
    __syncthreads();
    if (lane < 8) gi[lane] = s_int[lane];
}

// ---------------------------------------------------------------- C-ABI (include/ftl_gazebo.h)
struct ftl_gz_handle {
    FtlGzParams P;
    int device;
    size_t state_bytes, o_int, o_hist, o_corr, o_seg[FTL_GZ_MAX_LASERS], o_segn[FTL_GZ_MAX_LASERS];
    bool bound;
};

extern "C" {

int ftl_gz_create(const ftl_gz_config* cfg, int32_t n_envs, int32_t device, ftl_gz_handle** out) {
    if (!cfg || !out) return fail(FTL_E_INVALID, "null argument");
    if (n_envs <= 0) return fail(FTL_E_INVALID, "n_envs must be positive");
    if (device < 0) return fail(FTL_E_INVALID, "device < 0: this library has no CPU path");
    if (cfg->n_lasers < 0 || cfg->n_lasers > FTL_GZ_MAX_LASERS) return fail(FTL_E_INVALID, "n_lasers out of range");
    if (cfg->max_pts < 0 || cfg->max_pts > 4096) return fail(FTL_E_INVALID, "max_pts out of range");
    for (int k = 0; k < cfg->n_lasers; k++) {
        const ftl_gz_laser_cfg& l = cfg->lasers[k];
        if (!(l.count == 12 || l.count == 20 || l.count == 24 || l.count == 36)) return fail(FTL_E_INVALID, "Invalid number of laser beams, should be 12,24,20 or 36");
        if (l.history <= 0 || l.history > FTL_HMAX) return fail(FTL_E_INVALID, "max_prev_obs must be in 1..12");
        if (!(l.length > 0)) return fail(FTL_E_INVALID, "bad laser_length");
    }
    ftl_gz_handle* h = new (std::nothrow) ftl_gz_handle();
    if (!h) return fail(FTL_E_DEVICE, "out of host memory");
    memset(&h->P, 0, sizeof h->P);
    h->P.cfg = *cfg; h->P.n_envs = n_envs; h->device = device; h->bound = false;
    h->P.max_seg = 2 * FTL_GZ_HIST_CAP + 2 + cfg->max_pts;
    int off = 0;
    for (int k = 0; k < cfg->n_lasers; k++) { h->P.off[k] = off; off += cfg->lasers[k].history * cfg->lasers[k].count * (cfg->lasers[k].pad_sectors ? 4 : 1); }
    h->P.lasers_len = off;
    const size_t n = (size_t)n_envs;
    size_t cur = 0;
    h->o_int = cur; cur = align_up(cur + n * 8 * 4, 256);
    h->o_hist = cur; cur = align_up(cur + n * FTL_GZ_HIST_CAP * 2 * 8, 256);
    h->o_corr = cur; cur = align_up(cur + n * FTL_GZ_HIST_CAP * 4 * 8, 256);
    for (int k = 0; k < cfg->n_lasers; k++) {
        h->o_seg[k] = cur; cur = align_up(cur + n * cfg->lasers[k].history * (size_t)h->P.max_seg * 16, 256);
        h->o_segn[k] = cur; cur = align_up(cur + n * cfg->lasers[k].history * 4, 256);
    }
    h->state_bytes = cur;
    *out = h;
    return FTL_OK;
}
void ftl_gz_destroy(ftl_gz_handle* h) { delete h; }
size_t ftl_gz_state_bytes(const ftl_gz_handle* h) { return h ? h->state_bytes : 0; }
int32_t ftl_gz_lasers_len(const ftl_gz_handle* h) { return h ? h->P.lasers_len : 0; }
int ftl_gz_bind_state(ftl_gz_handle* h, void* dev_state, size_t bytes) {
    if (!h || !dev_state) return fail(FTL_E_INVALID, "null argument");
    if (bytes < h->state_bytes) return fail(FTL_E_INVALID, "state buffer too small");
    if (((uintptr_t)dev_state) & 255) return fail(FTL_E_INVALID, "state buffer must be 256-byte aligned");
    unsigned char* b = (unsigned char*)dev_state;
    h->P.gz_int = (int32_t*)(b + h->o_int); h->P.gz_hist = (double*)(b + h->o_hist); h->P.gz_corr = (double*)(b + h->o_corr);
    for (int k = 0; k < h->P.cfg.n_lasers; k++) { h->P.seg[k] = (float4*)(b + h->o_seg[k]); h->P.seg_n[k] = (int32_t*)(b + h->o_segn[k]); }
    h->bound = true;
    return FTL_OK;
}
int ftl_gz_state_field(const ftl_gz_handle* h, const char* name, size_t* offset, size_t* per_env, int32_t* dtype) {
    if (!h || !name) return fail(FTL_E_INVALID, "null argument");
    size_t o, p; int d;
    if (!strcmp(name, "gz_int")) { o = h->o_int; p = 8; d = 0; }
    else if (!strcmp(name, "gz_hist")) { o = h->o_hist; p = FTL_GZ_HIST_CAP * 2; d = 2; }
    else if (!strcmp(name, "gz_corr")) { o = h->o_corr; p = FTL_GZ_HIST_CAP * 4; d = 2; }
    else return fail(FTL_E_INVALID, std::string("unknown state field ") + name);
    if (offset) *offset = o; if (per_env) *per_env = p; if (dtype) *dtype = d;
    return FTL_OK;
}
static int gz_launch(ftl_gz_handle* h, const FtlGzCall& call, void* stream) {
    if (!h->bound) return fail(FTL_E_STATE, "ftl_gz_bind_state has not been called");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(ftl_gz_kernel, dim3((unsigned)h->P.n_envs), dim3(FTL_WAVE), 0, (hipStream_t)stream, h->P, call);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    return FTL_OK;
}
int ftl_gz_reset(ftl_gz_handle* h, const uint8_t* mask, void* stream) {
    if (!h) return fail(FTL_E_INVALID, "null argument");
    FtlGzCall call; memset(&call, 0, sizeof call); call.mask = mask; call.mode = 1;
    return gz_launch(h, call, stream);
}
int ftl_gz_step(ftl_gz_handle* h, const double* leader_pos, const double* yaw, const double* delta, const double* pts1, const double* pts2,
                const int32_t* n_pts, float* lasers, void* stream) {
    if (!h || !leader_pos || !yaw || !delta || !n_pts || (h->P.lasers_len > 0 && !lasers) || (h->P.cfg.max_pts > 0 && (!pts1 || !pts2)))
        return fail(FTL_E_INVALID, "null argument");
    FtlGzCall call; memset(&call, 0, sizeof call);
    call.leader = leader_pos; call.yaw = yaw; call.delta = delta; call.pts1 = pts1; call.pts2 = pts2; call.n_pts = n_pts; call.lasers = lasers; call.mode = 0;
    return gz_launch(h, call, stream);
}

}  // extern "C"

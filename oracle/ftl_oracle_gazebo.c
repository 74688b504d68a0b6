/*
 * ftl_oracle_gazebo.c -- CPU ORACLE (test infrastructure, NOT the product) of the follower-relative tracker / ray sensor:
 * a plain-C restatement of reference src/arctic_gym/gazebo_utils/gazebo_tracker.py ("GZ"), GazeboLeaderPositionsTracker_v2.scan
 * (GZ:17-172) and GazeboCorridor_Prev_lasers_v2.collect_obstacle_edges / scan (GZ:180-297).  Only tests/ may link or call it.
 *
 * Pinning: tests/golden/gazebo_*.npz, produced by tests/golden/gen/make_golden_gazebo.py from the UNMODIFIED reference classes
 * (loaded by file path: the package __init__ imports ray).  Under this image's numpy 2.2 the laser's scan() raises ValueError at
 * its `corridor_lines_item != []` test (GZ:226; numpy 1.x -- what the reference ran on -- evaluated it to True with a
 * DeprecationWarning); the generator restores that one legacy comparison result and changes nothing else (see its docstring).
 *
 * Dtype flow (verified against those vectors): history / corridor are float64; obstacle lines are cast to float32 (GZ:198);
 * follower_position is the Python list [0, 0], so np.array([follower_position]) is an INT array and every difference with it is
 * float64 -- unlike the 2-D env, where the origin is float32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ftl_gazebo.h"

#define PI_D 3.141592653589793
static const double DEG2RAD = PI_D / 180.0, RAD2DEG = 180.0 / PI_D;

typedef struct { float a[2], b[2]; } gseg_t;
typedef struct { gseg_t* s; int n; } gsnap_t;

typedef struct ftlo_gz {
    ftl_gz_config cfg;
    double hist[FTL_GZ_HIST_CAP][2]; int hist_len;
    double corr[FTL_GZ_HIST_CAP][4]; int corr_len;
    int counter;
    uint32_t error;
    gsnap_t* snaps[FTL_GZ_MAX_LASERS];
    int lasers_len, off[FTL_GZ_MAX_LASERS];
} ftlo_gz;

static int gwidth(const ftl_gz_laser_cfg* L) { return L->pad_sectors ? 4 * L->count : L->count; }
static double angle_correction(double a) { if (a >= 360.0) return a - 360.0; if (a < 0.0) return 360.0 + a; return a; }
static double norm1d(double x, double y) { return sqrt(fma(y, y, x * x)); }      /* np.linalg.norm of a 1-D float64 vector */
static double pairwise(const double* a, int n) {                                  /* np.sum */
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
    return pairwise(a, n2) + pairwise(a + n2, n - n2);
}
static double path_length(const ftlo_gz* e) {                                      /* GZ:130-132 */
    double d[FTL_GZ_HIST_CAP];
    int m = e->hist_len - 1;
    if (m < 1) return 0.0;
    for (int i = 0; i < m; i++) { double dx = e->hist[i][0] - e->hist[i + 1][0], dy = e->hist[i][1] - e->hist[i + 1][1]; d[i] = sqrt(dx * dx + dy * dy); }
    return pairwise(d, m);
}
static void popleft(ftlo_gz* e) {                                                   /* GZ:136-139: no error on an empty corridor */
    if (e->hist_len > 0) { memmove(e->hist, e->hist + 1, sizeof(e->hist[0]) * (size_t)(e->hist_len - 1)); e->hist_len--; }
    if (e->corr_len > 0) { memmove(e->corr, e->corr + 1, sizeof(e->corr[0]) * (size_t)(e->corr_len - 1)); e->corr_len--; }
}
static void border_pair(ftlo_gz* e, int i1, int i0, int ia) {                       /* GZ:148-163, max_dev = 2 */
    double vx = e->hist[i1][0] - e->hist[i0][0], vy = e->hist[i1][1] - e->hist[i0][1];
    const double sc = 2 / norm1d(vx, vy);
    vx *= sc; vy *= sc;
    const double c90 = cos(90.0 * DEG2RAD), s90 = sin(90.0 * DEG2RAD), cm90 = cos(-90.0 * DEG2RAD), sm90 = sin(-90.0 * DEG2RAD);
    if (e->corr_len >= FTL_GZ_HIST_CAP) { e->error |= FTL_ERR_CORR_OVERFLOW; return; }
    double* q = e->corr[e->corr_len++];
    q[0] = (c90 * vx + (-s90) * vy) + e->hist[ia][0]; q[1] = (s90 * vx + c90 * vy) + e->hist[ia][1];
    q[2] = (cm90 * vx + (-sm90) * vy) + e->hist[ia][0]; q[3] = (sm90 * vx + cm90 * vy) + e->hist[ia][1];
}

/* GazeboLeaderPositionsTracker_v2.scan, GZ:17-172 */
static void gz_tracker_scan(ftlo_gz* e, const double leader[2], double yaw_rad, const double delta[2]) {
    const double yaw = yaw_rad * RAD2DEG;                                        /* np.degrees(follower_orientation)[2] */
    for (int i = 0; i < e->hist_len; i++) {                                      /* GZ:46-55: shift + np.round(.., 5) */
        e->hist[i][0] = rint((e->hist[i][0] - delta[0]) * 1e5) / 1e5;
        e->hist[i][1] = rint((e->hist[i][1] - delta[1]) * 1e5) / 1e5;
    }
    for (int i = 0; i < e->corr_len; i++) {                                      /* GZ:58-79: shift, not rounded */
        e->corr[i][0] -= delta[0]; e->corr[i][1] -= delta[1]; e->corr[i][2] -= delta[0]; e->corr[i][3] -= delta[1];
    }
    if (e->counter % 3 == 0) {                                                   /* saving_period = 3, GZ:43 */
        if (e->hist_len > 0 && norm1d(leader[0] - e->hist[e->hist_len - 1][0], leader[1] - e->hist[e->hist_len - 1][1]) < 1)
            return;                                                              /* GZ:101-106: the counter is NOT advanced */
        if (e->hist_len == 0 && e->counter == 0) {                               /* GZ:108-123: 10 points from 10 behind the follower */
            const double th = angle_correction(yaw + 180) * DEG2RAD;
            const double sx = 10 * cos(th) + 0, sy = 10 * sin(th) + 0;
            const int n = 10;
            const double stepx = (leader[0] - sx) / (n - 1), stepy = (leader[1] - sy) / (n - 1);
            for (int i = 0; i < n; i++) {
                double x = (stepx == 0) ? ((double)i / (n - 1)) * (leader[0] - sx) + sx : (double)i * stepx + sx;
                double y = (stepy == 0) ? ((double)i / (n - 1)) * (leader[1] - sy) + sy : (double)i * stepy + sy;
                if (i == n - 1) { x = leader[0]; y = leader[1]; }
                e->hist[i][0] = x; e->hist[i][1] = y;
            }
            e->hist_len = n;
        } else {                                                                 /* GZ:124-130 */
            if (e->hist_len == 0) { e->error |= FTL_ERR_TRACKER_SEED; e->counter += 1; return; }   /* reference: IndexError on hist[-1] */
            const double last = norm1d(e->hist[e->hist_len - 1][0] - 0, e->hist[e->hist_len - 1][1] - 0);
            const double cur = norm1d(leader[0] - 0, leader[1] - 0);
            if (cur > last && cur < 25) {
                if (e->hist_len >= FTL_GZ_HIST_CAP) e->error |= FTL_ERR_CORR_OVERFLOW;
                else { e->hist[e->hist_len][0] = leader[0]; e->hist[e->hist_len][1] = leader[1]; e->hist_len++; }
            }
        }
        double path = path_length(e);                                            /* GZ:133-144, max_distance = 25 */
        while (path > 25) { popleft(e); path = path_length(e); }
        if (e->hist_len > 1) {                                                   /* GZ:147-163 */
            const int m = e->hist_len;
            if (e->counter == 0) for (int i = m - 1; i > 0; i--) border_pair(e, i, i - 1, m - i - 1);
            border_pair(e, m - 1, m - 2, m - 2);
        }
    }
    e->counter += 1;
}

static void gpush(gseg_t* s, int* n, double ax, double ay, double bx, double by) {
    s[*n].a[0] = (float)ax; s[*n].a[1] = (float)ay; s[*n].b[0] = (float)bx; s[*n].b[1] = (float)by; (*n)++;
}
/* GZ:180-200 */
static gseg_t* gz_collect(const ftlo_gz* e, const ftl_gz_laser_cfg* L, const double* p1, const double* p2, int n_pts, int* out_n) {
    const int C = e->corr_len;
    gseg_t* s = (gseg_t*)malloc(sizeof(gseg_t) * (size_t)(2 * C + 2 + (n_pts > 0 ? n_pts : 0) + 1));
    int n = 0;
    if (L->react_corridor) for (int i = 0; i < C - 1; i++) { gpush(s, &n, e->corr[i][0], e->corr[i][1], e->corr[i + 1][0], e->corr[i + 1][1]);
                                                              gpush(s, &n, e->corr[i][2], e->corr[i][3], e->corr[i + 1][2], e->corr[i + 1][3]); }
    if (L->react_green) { gpush(s, &n, e->corr[0][0], e->corr[0][1], e->corr[0][2], e->corr[0][3]);
                          gpush(s, &n, e->corr[C - 1][0], e->corr[C - 1][1], e->corr[C - 1][2], e->corr[C - 1][3]); }
    if (L->react_obstacles)
        for (int i = 0; i < n_pts - 1; i++) {
            if (norm1d(p1[2 * i] - p1[2 * i + 2], p1[2 * i + 1] - p1[2 * i + 3]) < 0.5) gpush(s, &n, p1[2 * i], p1[2 * i + 1], p1[2 * i + 2], p1[2 * i + 3]);
            else gpush(s, &n, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
        }
    *out_n = n;
    return s;
}
/* SEN:608-614 with C = np.array([[0, 0]]) (int): every difference with C is float64, B - A stays float32 */
static int gz_hit(const gseg_t* s, double dx, double dy) {
    const double ax = s->a[0], ay = s->a[1], bx = s->b[0], by = s->b[1];
    const float bax = s->b[0] - s->a[0], bay = s->b[1] - s->a[1];
    const int t1 = (dy - ay) * (0 - ax) > (0 - ay) * (dx - ax);                  /* ccw(A,C,D) */
    const int t2 = (dy - by) * (0 - bx) > (0 - by) * (dx - bx);                  /* ccw(B,C,D) */
    const int t3 = (0 - ay) * (double)bax > (double)bay * (0 - ax);              /* ccw(A,B,C) */
    const int t4 = (dy - ay) * (double)bax > (double)bay * (dx - ax);            /* ccw(A,B,D) */
    return (t1 != t2) && (t3 != t4);
}
/* SEN:626-640: da float32, db = D - C and dp = A - C float64 */
static double gz_hit_point(const gseg_t* s, double ex, double ey, double* ox, double* oy) {
    const float dax = s->b[0] - s->a[0], day = s->b[1] - s->a[1];
    const double dapx = (double)(-day), dapy = (double)dax;
    const double denom = dapx * ex + dapy * ey;
    const double num = dapx * (double)s->a[0] + dapy * (double)s->a[1];
    const double t = num / denom;
    const double x = t * ex + 0, y = t * ey + 0;
    *ox = x; *oy = y;
    return sqrt(x * x + y * y);
}
/* GZ:203-297 */
static void gz_laser_scan(ftlo_gz* e, int k, double yaw_rad, const double* p1, const double* p2, int n_pts, float* out) {
    const ftl_gz_laser_cfg* L = &e->cfg.lasers[k];
    const int N = L->count, H = L->history, W = gwidth(L);
    const double yaw = yaw_rad * RAD2DEG, period = 360.0 / N;
    if (e->corr_len <= 1) {            /* GZ:216/297: all_obs_arr unbound -> UnboundLocalError */
        e->error |= FTL_ERR_EMPTY_CORRIDOR;
        for (int i = 0; i < H * W; i++) out[i] = (float)L->length;
        return;
    }
    gsnap_t* hs = e->snaps[k];
    free(hs[0].s);
    memmove(hs, hs + 1, sizeof(gsnap_t) * (size_t)(H - 1));
    hs[H - 1].s = gz_collect(e, L, p1, p2, n_pts, &hs[H - 1].n);
    if (L->pad_sectors) for (int i = 0; i < H * W; i++) out[i] = 0.0f;
    for (int i = 0; i < N; i++) {
        const double th = ((yaw - 45) + i * period) * DEG2RAD;                   /* GZ:211-212 */
        const double ex = cos(th) * L->length, ey = sin(th) * L->length;
        for (int j = 0; j < H; j++) {
            double bx = ex, by = ey, best = 0; int found = 0;
            for (int m = 0; m < hs[j].n; m++)                                    /* a reset snapshot is one zero segment: never hit */
                if (gz_hit(&hs[j].s[m], ex, ey)) {
                    double x, y, d = gz_hit_point(&hs[j].s[m], ex, ey, &x, &y);
                    if (!found || d < best) { best = d; bx = x; by = y; found = 1; }
                }
            const float val = (float)norm1d(bx, by);                             /* GZ:252 */
            int col = i;
            if (L->pad_sectors) { const double lis = (double)N / 4; col = (((double)i < lis) ? 0 : ((double)i < 2 * lis) ? 1 : ((double)i < 3 * lis) ? 2 : 3) * N + i; }
            out[j * W + col] = val;
        }
    }
}

/* ------------------------------------------------------------------ public (ctypes) API */
ftlo_gz* ftlo_gz_create(const ftl_gz_config* cfg) {
    ftlo_gz* e = (ftlo_gz*)calloc(1, sizeof(ftlo_gz));
    e->cfg = *cfg;
    int off = 0;
    for (int k = 0; k < cfg->n_lasers; k++) { e->off[k] = off; off += cfg->lasers[k].history * gwidth(&cfg->lasers[k]);
                                              e->snaps[k] = (gsnap_t*)calloc((size_t)cfg->lasers[k].history, sizeof(gsnap_t)); }
    e->lasers_len = off;
    return e;
}
void ftlo_gz_destroy(ftlo_gz* e) {
    if (!e) return;
    for (int k = 0; k < e->cfg.n_lasers; k++) { for (int j = 0; j < e->cfg.lasers[k].history; j++) free(e->snaps[k][j].s); free(e->snaps[k]); }
    free(e);
}
void ftlo_gz_reset(ftlo_gz* e) {
    e->hist_len = 0; e->corr_len = 0; e->counter = 0; e->error = 0;
    for (int k = 0; k < e->cfg.n_lasers; k++) for (int j = 0; j < e->cfg.lasers[k].history; j++) { free(e->snaps[k][j].s); e->snaps[k][j].s = NULL; e->snaps[k][j].n = 0; }
}
int ftlo_gz_lasers_len(const ftlo_gz* e) { return e->lasers_len; }
void ftlo_gz_step(ftlo_gz* e, const double* leader, double yaw, const double* delta, const double* p1, const double* p2, int n_pts, float* lasers) {
    gz_tracker_scan(e, leader, yaw, delta);
    for (int k = 0; k < e->cfg.n_lasers; k++) gz_laser_scan(e, k, yaw, p1, p2, n_pts, lasers + e->off[k]);
}
void ftlo_gz_get(const ftlo_gz* e, int32_t* counts, double* hist, double* corr) {
    counts[0] = e->counter; counts[1] = e->hist_len; counts[2] = e->corr_len; counts[3] = (int32_t)e->error;
    memcpy(hist, e->hist, sizeof(double) * 2 * (size_t)e->hist_len);
    memcpy(corr, e->corr, sizeof(double) * 4 * (size_t)e->corr_len);
}

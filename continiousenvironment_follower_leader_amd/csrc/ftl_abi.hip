// ftl_abi.hip -- host side of the C-ABI declared in include/ftl.h: config validation, state layout, kernel launch.
// There is no CPU fallback anywhere in this file: every entry point that computes launches the HIP kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <cstdlib>
#include <cmath>

#include "ftl_device.hpp"
#include "ftl_frames_group.hpp"
#include "ftl_aux.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

struct Field { const char* name; size_t offset, per_env; int dtype; size_t stride; };     // stride: bytes from one env's row to the next
constexpr int FTL_N_FIELDS = 15;

}  // namespace

struct ftl_handle {
    FtlDevParams P;          // host copy of the frozen parameters
    FtlDevParams* dP;        // device copy read by the kernel (library-owned, ~1 KB)
    bool dirty;
    int device;
    size_t state_bytes;
    Field fields[FTL_N_FIELDS];
    void* mt_mem;            // partial sums of ftl_episode_metrics (library-owned)
    bool timing;             // ftl_kernel_timing: events around every launch of a step
    std::vector<hipEvent_t> tev;   // 5 per timed step: before frames | after frames | after rays | after aux | after regroup
    size_t tev_used;
    bool bound, have_scen;
    bool regroup;            // envs are regrouped by expected cost after every launch (off: FTL_NO_REGROUP=1, or too many envs)
    void* rg_mem;            // perm | bh | rank | keys | two key-total buffers (library-owned)
    int* rg_tot;             // [2][FTL_NKEYS]
    unsigned rg_parity, rg_launches, rg_every;
    bool rg_env, rg_every_env, split_env;   // the environment switch was given: it wins over ftl_tune
    int G;                   // lanes per env in the frame kernel (set_lanes)
    int co_envs, cus;        // envs stepped on the device at the same time (ftl_tune; default: this handle's), CUs of the device
    bool g_env;              // FTL_DEBUG_G8 was given
    int rg_slots, rg_epw;    // frame-kernel wavefronts one round holds on this device / envs per wavefront (the cost sort's auto rule)
    // optionally the slot groups are stepped as two interleaved halves on two streams (the caller's stream waits for the side
    // stream): the ray kernel of one half fills the tail of the other half's frame kernel
    hipStream_t side; hipEvent_t ev_fork, ev_join; bool split;
    int win_base, win_count, win_stride; // pool entries the auto-reset draws from (ftl_set_reset_window)
    size_t lds_pad;          // FTL_DEBUG_LDS_PAD (diagnostic: lowers the frame kernel's occupancy without touching the code), read once at create
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int validate(const ftl_config& c, std::string& why) {
    char buf[256];
#define REQ(cond, ...) do { if (!(cond)) { snprintf(buf, sizeof buf, __VA_ARGS__); why = buf; return FTL_E_INVALID; } } while (0)
    REQ(c.abi_version == FTL_ABI_VERSION, "abi_version %d != %d", c.abi_version, FTL_ABI_VERSION);
    REQ(c.width > 0 && c.height > 0, "bad field size");
    REQ(c.frames_per_step > 0, "frames_per_step must be positive");
    REQ(c.rand_fps_hi == 0 || (c.rand_fps_lo > 0 && c.rand_fps_lo < c.rand_fps_hi), "random_frames_per_step must be (low, high) with 0 < low < high");
    REQ(c.trajectory_saving_period > 0, "trajectory_saving_period must be positive");
    REQ(c.n_static >= 0 && c.n_static <= 4096, "n_static out of range");
    REQ(c.n_bears >= 0 && c.n_bears <= FTL_MAX_BEARS, "n_bears out of range (0..%d)", FTL_MAX_BEARS);
    REQ(c.n_lasers >= 0 && c.n_lasers <= FTL_MAX_LASERS, "n_lasers out of range");
    REQ(c.n_lasers == 0 || c.has_tracker, "ray sensors need the tracker (classes.py:280 would raise NameError)");
    REQ(c.traj_cap >= 8 && c.corr_cap >= 8 && c.corr_cap <= 512 && c.route_cap >= 2 && c.init_traj_cap >= 1, "bad capacities");
    REQ((c.corr_cap & (c.corr_cap - 1)) == 0, "corr_cap must be a power of two (the tracker rings are indexed with a mask)");
    REQ(c.init_traj_cap <= c.traj_cap, "init_traj_cap > traj_cap");
    REQ(c.traj_cap % FTL_TRAJ_BLOCK == 0, "traj_cap must be a multiple of FTL_TRAJ_BLOCK");
    if (c.has_tracker == 2) {
        REQ(c.tracker_saving_period > 0, "tracker saving_period must be positive");
        REQ(c.corridor_length > 0 && c.corridor_width > 0, "corridor_length / corridor_width must be positive");
    }
    REQ(c.n_speed_regime <= FTL_MAX_REGIME && c.n_acc_regime <= FTL_MAX_REGIME, "too many regime entries");
    REQ(c.has_tracker >= 0 && c.has_tracker <= 2, "has_tracker must be 0, 1 (v1) or 2 (v2)");
    if (c.has_tracker == 1) REQ(c.tracker_saving_period > 0 && c.hist1_cap >= 8 && c.corridor_width > 0, "bad v1 tracker parameters");
    REQ(c.n_aux >= 0 && c.n_aux <= FTL_MAX_AUX, "n_aux out of range");
    for (int j = 0; j < c.n_aux; j++) {
        const ftl_aux_cfg& a = c.aux[j];
        REQ(a.kind >= FTL_AUX_LIDAR && a.kind <= FTL_AUX_TRACK_RADAR, "aux %d: unknown sensor kind", j);
        if (a.kind == FTL_AUX_LIDAR) {
            REQ(a.n_angles > 0 && a.n_angles <= 512 && a.points_number > 0 && a.range_px > 0, "aux %d: bad lidar parameters", j);
            // the (ray, marching point) index of the lidar is split with a float quotient that is exact below 2^16 items (ftl_aux.hpp)
            REQ(a.points_number <= 1024 && a.n_angles * a.points_number < 65536, "aux %d: lidar with more than 1024 points per ray or 65535 (ray, point) pairs", j);
        }
        else {
            REQ(c.has_tracker != 0, "aux %d: a leader-track detector needs a tracker (classes.py:272 would raise NameError)", j);
            REQ(a.seq_len > 0 && a.detectable >= 0 && a.detectable <= (a.kind == FTL_AUX_TRACK_RADAR ? 2 : 1), "aux %d: bad detector parameters", j);
            if (a.kind == FTL_AUX_TRACK_RADAR) REQ(a.radar_sectors > 0 && a.radar_sectors <= 4096, "aux %d: bad radar_sectors_number", j);
        }
    }
    for (int k = 0; k < c.n_lasers; k++) {
        const ftl_laser_cfg& l = c.lasers[k];
        REQ(l.count > 0 && l.count <= 1024, "laser %d: bad lasers_count", k);
        REQ(l.history > 0 && l.history <= FTL_HMAX, "laser %d: max_prev_obs must be in 1..%d", k, FTL_HMAX);
        REQ(l.react_obstacles >= 0 && l.react_obstacles <= 3, "laser %d: bad react_to_obstacles", k);
        REQ(l.length > 0, "laser %d: bad laser_length", k);
    }
#undef REQ
    return FTL_OK;
}

}  // namespace

extern "C" {

const char* ftl_last_error(void) { return g_err.c_str(); }

size_t ftl_sizeof_config(void) { return sizeof(ftl_config); }
size_t ftl_sizeof_scenarios(void) { return sizeof(ftl_scenarios); }
size_t ftl_sizeof_outputs(void) { return sizeof(ftl_outputs); }
size_t ftl_sizeof_scen_params(void) { return sizeof(ftl_scen_params); }

// Lanes per env of the frame kernel (4 or 8) and everything that follows from the envs per wavefront: the LDS layout of the kernel and the
// slot count of the cost sort's rule.  Configs with more than 2 dynamic obstacles need 8 lanes.  The others take 8 as well -- 8 envs per
// wavefront, five lanes of a group idle -- when the batch is small enough for every such wavefront to have a SIMD of its own: the launch
// then takes as long as its slowest wavefront, and a wavefront with half the envs meets half the rare paths (resets, searches, walks):
// config B at 8,192 envs +3 %, config D +4 %, nothing from 16,384 envs on.  FTL_DEBUG_G8=0/1 overrides.  0 on success.
static int set_lanes(ftl_handle* h) {
    FtlDevParams& P = h->P;
    const ftl_config* cfg = &P.cfg;
    if (!h->g_env) h->G = (P.R > 4 || h->co_envs <= h->cus * 4 * 8) ? 8 : 4;
    if (P.R > 4) h->G = 8;
    const int epw = FTL_WAVE / h->G;
    h->rg_epw = epw;
    const int f_max = cfg->rand_fps_hi > 0 ? cfg->rand_fps_hi - 1 : cfg->frames_per_step;
    size_t o = align_up((size_t)epw * cfg->n_static * 16 + (size_t)epw * 4 + 32, 16);
    P.fr_rec_stride = (int)align_up((size_t)f_max, 16);
    P.fr_rec_off = (int)o; o += (size_t)P.fr_rec_stride * epw;
    P.fr_pend_off = (int)o; o += (size_t)epw * (P.fr_defer ? f_max - 1 : 1) * 16;
    P.fr_env_off = (int)o; o += (size_t)epw * 4 + 16 + (size_t)epw * 16;      // slot -> env, item counter, box of the trajectory block being filled
    P.fr_lds = (int)o;
    h->dirty = true;
    return o > 64 * 1024 ? 1 : 0;
}

int ftl_create(const ftl_config* cfg, int32_t n_envs, int32_t device, ftl_handle** out) {
    if (!cfg || !out) return fail(FTL_E_INVALID, "null argument");
    if (n_envs <= 0) return fail(FTL_E_INVALID, "n_envs must be positive");
    if (device < 0) return fail(FTL_E_INVALID, "device < 0: this library has no CPU path");
    std::string why;
    int rc = validate(*cfg, why);
    if (rc) return fail(rc, why);
    ftl_handle* h = new (std::nothrow) ftl_handle();
    if (!h) return fail(FTL_E_DEVICE, "out of host memory");
    memset(&h->P, 0, sizeof h->P);
    h->P.cfg = *cfg;
    h->device = device;
    h->bound = false; h->have_scen = false; h->dP = nullptr; h->dirty = true;
    h->rg_mem = nullptr; h->rg_tot = nullptr; h->rg_parity = 0; h->rg_launches = 0; h->mt_mem = nullptr; h->timing = false; h->tev_used = 0;
    h->side = nullptr; h->ev_fork = nullptr; h->ev_join = nullptr;
    {   // measured: +9 % with random_frames_per_step (long frame kernels whose tails the other half's ray kernel fills), -1 % with a
        // fixed 10 frames per step -- so it is on for the former only; FTL_SPLIT=0/1 overrides
        const char* sp = getenv("FTL_SPLIT");
        h->split = n_envs >= 8192 && (sp ? sp[0] == '1' : cfg->rand_fps_hi > 0) && cfg->has_tracker != 1;   // (the v1 tracker kernel covers all envs at once)
        h->split_env = sp != nullptr;
    }
    {   // the scatter pass reads one histogram row per block of 1024 envs: fine up to a few hundred blocks
        const char* off = getenv("FTL_NO_REGROUP");
        const char* ev = getenv("FTL_REGROUP_EVERY");      // tuning knob: rebuild the permutation every k-th launch (default 4)
        h->rg_every = (ev && atoi(ev) > 0) ? (unsigned)atoi(ev) : 4u;       // (round 3: every 4th launch, 212 against 209 M env-steps/s at every 2nd -- with the
                                                                            //  later frames' searches deferred the frame kernel is as fast on a staler order)
        // Sorting the envs by expected cost pays when the frame kernel runs in more than one round of wavefronts (the long ones start
        // first, the short ones fill in behind them: +10 % on config B at 65,536 envs).  When every wavefront is resident from the start
        // the launch takes as long as its slowest wavefront, and a wavefront that holds ALL the expensive envs is slower than any
        // wavefront of an unsorted batch: config E at 32,768 envs -9 %, config D at 4,096 envs -16 % with the sort.  So it is on only
        // beyond one round -- or with random frame counts, whose keys make the wavefronts uniform in length.  FTL_NO_REGROUP=0/1 overrides.
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        const char* g8 = getenv("FTL_DEBUG_G8");
        h->g_env = g8 != nullptr; h->cus = cus; h->co_envs = n_envs;
        h->G = (2 + cfg->n_bears > 4 || (g8 ? g8[0] == '1' : n_envs <= cus * 4 * 8)) ? 8 : 4;       // (set_lanes)
        const int epw_f = FTL_WAVE / h->G;
        const bool beyond_one_round = (n_envs + epw_f - 1) / epw_f > cus * 4 * FTL_FRAMESG_WPE;
        h->regroup = (off ? off[0] != '1' : (beyond_one_round || cfg->rand_fps_hi > 0)) && (n_envs + FTL_RG_BLOCK - 1) / FTL_RG_BLOCK <= 512;
        h->rg_env = off != nullptr; h->rg_every_env = ev && atoi(ev) > 0; h->rg_slots = cus * 4 * FTL_FRAMESG_WPE; h->rg_epw = epw_f;
    }
    FtlDevParams& P = h->P;
    P.n_envs = n_envs;
    P.R = 2 + cfg->n_bears;
    int off = 0, hmax = 1, rays = 0;
    auto width_of = [](const ftl_laser_cfg& l) { return l.count * (l.compas ? 5 : (l.pad_sectors ? 4 : 1)); };
    for (int k = 0; k < cfg->n_lasers; k++) {
        P.cfg.lasers[k].out_offset = off;
        off += cfg->lasers[k].history * width_of(cfg->lasers[k]);
        hmax = cfg->lasers[k].history > hmax ? cfg->lasers[k].history : hmax;
        rays += cfg->lasers[k].count;
    }
    for (int j = 0; j < cfg->n_aux; j++) {       // lidar / detector blocks follow the ray sensors' blocks
        ftl_aux_cfg& a = P.cfg.aux[j];
        a.out_len = a.kind == FTL_AUX_LIDAR ? (a.return_all_points ? 1 + a.n_angles * a.points_number * (a.return_only_distances ? 1 : 2) : a.n_angles * (a.return_only_distances ? 1 : 2))
                  : a.kind == FTL_AUX_TRACK_VECTOR ? 2 * a.seq_len : a.radar_sectors;
        a.out_offset = off; off += a.out_len;
    }
    P.lasers_len = off; P.total_rays = rays; P.hmax = hmax;
    {   // ray directions relative to the heading (glibc cos / sin, as the reference's np.cos(np.radians(.))) and the no-hit reading
        const double deg2rad = 3.141592653589793 / 180.0;
        int g = 0; bool miss_ok = true;
        const double span = (double)(cfg->width > cfg->height ? cfg->width : cfg->height);
        const double margin = 1e-11 * (span > 2048.0 ? span / 2048.0 : 1.0);
        for (int k = 0; k < cfg->n_lasers; k++) {
            const ftl_laser_cfg& l = cfg->lasers[k];
            for (int i = 0; i < l.count && g < FTL_MAX_RAYS; i++, g++) {
                const double a = l.explicit_angles ? l.ray_angles[i & 7] : l.angle_offset + i * (360.0 / (double)l.count);
                P.ray_rot[g][0] = cos(a * deg2rad); P.ray_rot[g][1] = sin(a * deg2rad);
            }
            // float64 |end - origin| = laser_length within 4e-13 (coordinates < 2048): float32(.) is float32(laser_length) unless a
            // rounding boundary of float32 lies that close
            miss_ok = miss_ok && (float)(l.length - margin) == (float)l.length && (float)(l.length + margin) == (float)l.length;
        }
        P.miss_const = miss_ok ? 1 : 0;
        int rb[2] = {0, 0};              // phase 3's per-sensor records; ray indices run over the sensors of one pass (before / after the tracker)
        for (int k = 0; k < cfg->n_lasers; k++) {
            const ftl_laser_cfg& l = cfg->lasers[k];
            FtlRaySensor& rs = P.ray_sens[k];
            const int which = l.after_tracker ? 1 : 0, ro = l.react_obstacles;
            rs.count = l.count; rs.rbase = rb[which];
            rs.reach2 = ((float)l.length + 2.0f) * ((float)l.length + 2.0f);
            rs.inv_step = (float)l.count * 0.15915494309189535f; rs.inv_count = 1.0f / (float)l.count;
            rs.off_u = (float)(l.angle_offset * deg2rad) * rs.inv_step;
            rs.slack = 0.02f + 0.01f * (float)l.count * 0.15915494f;
            rs.flags = ((ro == 1 || ro == 2) ? 1u : 0u) | ((ro == 1 || ro == 3) ? 2u : 0u) | (l.react_corridor ? 4u : 0u) | (l.react_green ? 8u : 0u)
                     | (l.explicit_angles ? 16u : 0u) | (which ? 32u : 0u) | (l.compas ? 64u : 0u);
            if (!l.compas) rb[which] += l.count;
        }
        P.inv_nrect_dyn = (65536u + (unsigned)(P.R - 1) - 1u) / (unsigned)(P.R - 1);
    }
    if (rays > 1023) { delete h; return fail(FTL_E_INVALID, "more than 1023 rays per env (the candidate list of the ray kernel packs a ray index into 10 bits)"); }
    if (hmax * (P.R - 1) > FTL_WAVE) { delete h; return fail(FTL_E_INVALID, "max_prev_obs x (1 + bears) exceeds one wavefront of snapshot rects"); }
    {   // row width / common history of the fused sensorPrev output: the sensors wrappers.py:204, 214 select (in_policy_obs), in dict order
        int w = 0, hcommon = 0;
        for (int k = 0; k < cfg->n_lasers; k++) {
            P.pol_off[k] = -1;
            if (!cfg->lasers[k].in_policy_obs) continue;
            P.pol_off[k] = w; w += width_of(cfg->lasers[k]);
            if (hcommon == 0) hcommon = cfg->lasers[k].history;
            else if (cfg->lasers[k].history != hcommon) hcommon = -1;
        }
        P.pol_width = w; P.pol_h = hcommon;
    }
    // State layout.  The small fields of an env form one record (fields in the order below, the ray kernel's inputs first; 16-byte aligned
    // fields, a stride that is a multiple of 128 bytes); the long ones are [n_envs][per_env] arrays in 256-byte aligned regions.
    const size_t n = (size_t)n_envs;
    struct { const char* name; size_t per_env; int dtype; size_t esz; bool rec; } spec[FTL_N_FIELDS] = {
        {"rb_pos", (size_t)P.R * 2, 1, 4, true}, {"rb_dbl", (size_t)P.R * FTL_RD_COUNT, 2, 8, true}, {"rb_int", (size_t)P.R * FTL_RI_COUNT, 0, 4, true},
        {"env_int", FTL_EI_COUNT, 0, 4, true}, {"env_dbl", FTL_ED_COUNT, 2, 8, true}, {"traj", (size_t)cfg->traj_cap * 2, 1, 4, false},
        {"hist", (size_t)cfg->corr_cap * 2, 2, 8, false}, {"corr", (size_t)cfg->corr_cap * 4, 2, 8, false},
        {"snap_rects", (size_t)hmax * (P.R - 1) * 4, 0, 4, true}, {"snap_win", (size_t)hmax * 4, 0, 4, true},
        {"traj_bb", (size_t)(cfg->traj_cap / FTL_TRAJ_BLOCK) * 4, 1, 4, false}, {"ep_stats", FTL_N_METRICS, 2, 8, false},
        {"hist1", (size_t)(cfg->has_tracker == 1 ? cfg->hist1_cap : 0) * 2, 1, 4, false}, {"fol_cs", 2, 2, 8, true},
        {"corr32", (size_t)cfg->corr_cap * 4, 1, 4, false}};
    static const int rec_order[8] = {3 /*env_int*/, 13 /*fol_cs*/, 0 /*rb_pos*/, 1 /*rb_dbl*/, 9 /*snap_win*/, 8 /*snap_rects*/, 4 /*env_dbl*/, 2 /*rb_int*/};
    size_t ro = 0;
    for (int k = 0; k < 8; k++) {
        const int i = rec_order[k];
        ro = align_up(ro, 16);
        h->fields[i] = Field{spec[i].name, ro, spec[i].per_env, spec[i].dtype, 0};
        ro += spec[i].per_env * spec[i].esz;
    }
    const size_t rec_stride = align_up(ro, 128);
    P.rec_stride = (int32_t)rec_stride;
    for (int k = 0; k < 8; k++) h->fields[rec_order[k]].stride = rec_stride;
    size_t cur = rec_stride * n;
    for (int i = 0; i < FTL_N_FIELDS; i++) {
        if (spec[i].rec) continue;
        cur = align_up(cur, 256);
        h->fields[i] = Field{spec[i].name, cur, spec[i].per_env, spec[i].dtype, spec[i].per_env * spec[i].esz};
        cur += spec[i].per_env * spec[i].esz * n;
    }
    h->state_bytes = align_up(cur, 256);
    {   // corridor ring (f32x4) + near rects (int4 + u32) + corridor refs (u32) + green caps (f32x4 + u32) + counters
        // + ray ends (double2) + per-(ray, snapshot) minima (u64 x HM) + miss readings (f64)
        const size_t rects = (size_t)cfg->n_static + hmax + (size_t)hmax * (P.R - 2 > 0 ? P.R - 2 : 0) + 1;
        // LDS copy of the corridor ring: up to 128 points (the windows of the snapshots span a few dozen; the ring itself is sized
        // for a crawling leader); FTL_DEBUG_CORR_LDS_CAP forces a smaller copy (tests of the unstaged path)
        P.corr_lds_cap = cfg->corr_cap < 128 ? cfg->corr_cap : 128;
        if (const char* lc = getenv("FTL_DEBUG_CORR_LDS_CAP")) { int v = atoi(lc); if (v >= 2 && v <= cfg->corr_cap && (v & (v - 1)) == 0) P.corr_lds_cap = v; }
        P.lds_rays = (int)((size_t)P.corr_lds_cap * 16 + rects * 20 + (size_t)2 * P.corr_lds_cap * 4 + (size_t)2 * hmax * 20 + 64
                           + (size_t)rays * 16 + (size_t)rays * (hmax <= 5 ? 5 : FTL_HMAX) * 4      /* float32 minima, >= the HM of whichever instantiation launch() picks */ + (size_t)rays * 4
                           + rects * 8 + 32                   /* facing-edge list (u16 x 4 per rect) + edge counters */
                           + (size_t)FTL_PAIR_CAP * 2 + 16);  /* candidate list of phase 3 */
    }
    {   // frame kernel LDS: near lists + their counters | frame records (one byte per env and frame) | pending position checks | slot -> env | block boxes
        const int f_max = cfg->rand_fps_hi > 0 ? cfg->rand_fps_hi - 1 : cfg->frames_per_step;
        // The searches of frames 1.. wait for the end of the step when the step is short enough for their items to sit in LDS and the frame
        // count is the same for every env; otherwise every frame's searches run right after it (one item per env at most).
        const char* dv = getenv("FTL_DEFER");
        P.fr_defer = (cfg->rand_fps_hi == 0 && f_max >= 2 && f_max <= 16 && cfg->traj_cap <= 65535 && !(dv && dv[0] == '0')) ? 1 : 0;
        if (cfg->traj_cap > 65535 || f_max > 4095) { delete h; return fail(FTL_E_INVALID, "traj_cap above 65535 or more than 4095 frames per step"); }
        if (set_lanes(h)) { delete h; return fail(FTL_E_INVALID, "the frame kernel needs more than 64 KiB of LDS per wavefront (static rects x frames per step)"); }
    }
    auto debug_pad = [](const char* name) { const char* v = getenv(name); const int p = v ? atoi(v) : 0; return p < 0 ? 0 : (p > 48 * 1024 ? 48 * 1024 : p); };
    P.lds_rays += debug_pad("FTL_DEBUG_LDS_PAD_RAYS");      // diagnostic: occupancy of the ray kernel without touching the code
    h->lds_pad = (size_t)debug_pad("FTL_DEBUG_LDS_PAD");
    if (getenv("FTL_DEBUG_PRINT_LDS")) fprintf(stderr, "ftl: ray kernel LDS %d B per env\n", P.lds_rays);
    if (P.lds_rays > 64 * 1024) { delete h; return fail(FTL_E_INVALID, "config needs more than 64 KiB of LDS per env"); }
    if (ftl_aux_lds_bytes(*cfg) > 64 * 1024) { delete h; return fail(FTL_E_INVALID, "the compas / lidar / radar sensors of this config need more than 64 KiB of LDS per env"); }
    *out = h;
    return FTL_OK;
}

void ftl_destroy(ftl_handle* h) {
    if (!h) return;
    if (h->dP || h->rg_mem || h->side || h->mt_mem) (void)hipSetDevice(h->device);
    if (h->dP) (void)hipFree(h->dP);
    if (h->mt_mem) (void)hipFree(h->mt_mem);
    for (hipEvent_t ev : h->tev) (void)hipEventDestroy(ev);
    if (h->rg_mem) (void)hipFree(h->rg_mem);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

int32_t ftl_lasers_len(const ftl_handle* h) { return h ? h->P.lasers_len : 0; }

int ftl_get_config(const ftl_handle* h, ftl_config* out) {
    if (!h || !out) return fail(FTL_E_INVALID, "null argument");
    *out = h->P.cfg;
    return FTL_OK;
}

size_t ftl_state_bytes(const ftl_handle* h) { return h ? h->state_bytes : 0; }

int ftl_state_field(const ftl_handle* h, const char* name, size_t* offset, size_t* per_env, int32_t* dtype, size_t* stride) {
    if (!h || !name) return fail(FTL_E_INVALID, "null argument");
    for (int i = 0; i < FTL_N_FIELDS; i++)
        if (!strcmp(h->fields[i].name, name)) {
            if (offset) *offset = h->fields[i].offset;
            if (per_env) *per_env = h->fields[i].per_env;
            if (dtype) *dtype = h->fields[i].dtype;
            if (stride) *stride = h->fields[i].stride;
            return FTL_OK;
        }
    return fail(FTL_E_INVALID, std::string("unknown state field ") + name);
}

int ftl_bind_state(ftl_handle* h, void* dev_state, size_t bytes) {
    if (!h || !dev_state) return fail(FTL_E_INVALID, "null argument");
    if (bytes < h->state_bytes) return fail(FTL_E_INVALID, "state buffer too small");
    if (((uintptr_t)dev_state) & 255) return fail(FTL_E_INVALID, "state buffer must be 256-byte aligned");
    unsigned char* b = (unsigned char*)dev_state;
    FtlDevParams& P = h->P;
    P.rb_pos = (float*)(b + h->fields[0].offset); P.rb_dbl = (double*)(b + h->fields[1].offset); P.rb_int = (int32_t*)(b + h->fields[2].offset);
    P.env_int = (int32_t*)(b + h->fields[3].offset); P.env_dbl = (double*)(b + h->fields[4].offset); P.traj = (float*)(b + h->fields[5].offset);
    P.hist = (double*)(b + h->fields[6].offset); P.corr = (double*)(b + h->fields[7].offset);
    P.snap_rects = (int32_t*)(b + h->fields[8].offset); P.snap_win = (int32_t*)(b + h->fields[9].offset);
    P.traj_bb = (float*)(b + h->fields[10].offset); P.ep_stats = (double*)(b + h->fields[11].offset);
    P.hist1 = (float*)(b + h->fields[12].offset); P.fol_cs = (double*)(b + h->fields[13].offset); P.corr32 = (float*)(b + h->fields[14].offset);
    h->bound = true; h->dirty = true;
    return FTL_OK;
}

int ftl_load_scenarios(ftl_handle* h, const ftl_scenarios* pool) {
    if (!h || !pool) return fail(FTL_E_INVALID, "null argument");
    if (pool->n_scenarios <= 0) return fail(FTL_E_INVALID, "empty scenario pool");
    if (!pool->robot_pos || !pool->robot_dir || !pool->robot_rect || !pool->route || !pool->route_len ||
        !pool->init_traj || !pool->init_traj_len || (h->P.cfg.n_static > 0 && !pool->static_rects))
        return fail(FTL_E_INVALID, "scenario pool has null arrays");
    h->P.scen = *pool;
    h->have_scen = true; h->dirty = true;
    h->win_base = 0; h->win_count = pool->n_scenarios; h->win_stride = h->P.n_envs % pool->n_scenarios;
    return FTL_OK;
}

int ftl_set_reset_window(ftl_handle* h, int32_t base, int32_t count, int32_t stride) {
    if (!h) return fail(FTL_E_INVALID, "null argument");
    if (!h->have_scen) return fail(FTL_E_STATE, "ftl_load_scenarios has not been called");
    if (base < 0 || count <= 0 || base + count > h->P.scen.n_scenarios) return fail(FTL_E_INVALID, "reset window outside the scenario pool");
    h->win_base = base; h->win_count = count; h->win_stride = (stride > 0 ? stride : h->P.n_envs) % count;
    return FTL_OK;
}

int ftl_tune(ftl_handle* h, int32_t what, int32_t value) {
    if (!h) return fail(FTL_E_INVALID, "null argument");
    switch (what) {
    case FTL_TUNE_COSCHEDULED_ENVS:
        if (value < h->P.n_envs) return fail(FTL_E_INVALID, "co-scheduled envs below this handle's own");
        h->co_envs = value;
        if (set_lanes(h)) return fail(FTL_E_INVALID, "the frame kernel needs more than 64 KiB of LDS per wavefront (static rects x frames per step)");
        if (!h->rg_env)
            h->regroup = ((value + h->rg_epw - 1) / h->rg_epw > h->rg_slots || h->P.cfg.rand_fps_hi > 0) && (h->P.n_envs + FTL_RG_BLOCK - 1) / FTL_RG_BLOCK <= 512;
        return FTL_OK;
    case FTL_TUNE_REGROUP_EVERY:
        if (value < 1) return fail(FTL_E_INVALID, "regroup interval below 1");
        if (!h->rg_every_env) h->rg_every = (unsigned)value;
        return FTL_OK;
    case FTL_TUNE_TWO_STREAMS:
        if (!h->split_env) h->split = value != 0 && h->P.n_envs >= 8192 && h->P.cfg.has_tracker != 1;
        return FTL_OK;
    }
    return fail(FTL_E_INVALID, "unknown tuning key");
}

// device copy of the frozen parameters: (re)uploaded only after bind_state / load_scenarios, never on the steady-state step path
static int sync_params(ftl_handle* h) {
    hipError_t e;
    if (!h->dP) {
        e = hipMalloc((void**)&h->dP, sizeof(FtlDevParams));
        if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipMalloc(params): ") + hipGetErrorString(e));
        h->dirty = true;
    }
    if (h->dirty) {
        e = hipMemcpy(h->dP, &h->P, sizeof(FtlDevParams), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipMemcpy(params): ") + hipGetErrorString(e));
        h->dirty = false;
    }
    return FTL_OK;
}

static int launch(ftl_handle* h, const FtlCall& call, void* stream) {
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    if (h->regroup && !h->rg_mem) {
        const size_t n = (size_t)h->P.n_envs, nb = (n + FTL_RG_BLOCK - 1) / FTL_RG_BLOCK;
        const size_t o_bh = align_up(n * 4, 256), o_rank = o_bh + align_up(nb * FTL_NKEYS * 4, 256), o_keys = o_rank + align_up(n * 2, 256), o_tot = o_keys + align_up(n, 256);
        e = hipMalloc(&h->rg_mem, o_tot + 2 * FTL_NKEYS * sizeof(int));
        if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipMalloc(regroup): ") + hipGetErrorString(e));
        char* b = (char*)h->rg_mem;
        h->P.perm = (int32_t*)b; h->P.bh = (int32_t*)(b + o_bh); h->P.rank = (uint16_t*)(b + o_rank); h->P.keys = (uint8_t*)(b + o_keys);
        h->rg_tot = (int*)(b + o_tot); h->rg_parity = 0;
        e = hipMemsetAsync(h->rg_tot, 0, 2 * FTL_NKEYS * sizeof(int), (hipStream_t)stream);
        if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipMemsetAsync(regroup): ") + hipGetErrorString(e));
        hipLaunchKernelGGL(ftl::ftl_perm_identity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->P.perm, (int)n);
        h->dirty = true;
    }
    { int rc = sync_params(h); if (rc) return rc; }
    if (h->split && !h->side) {
        if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)
            return fail(FTL_E_DEVICE, "cannot create the side stream / events");
    }
    // one slot range: frame loop (G lanes per env: 4 for <= 2 dynamic obstacles, else 8; configs with leader regimes or random
    // frame counts use the instantiations that carry that code), then the ray sensors of the same envs
    // (the halves are interleaved wavefront by wavefront, so both see the same mix of the cost-sorted slots)
    // optional per-kernel timing (ftl_kernel_timing): up to 512 steps of 4 events each, single-stream mode only
    hipEvent_t* tev = nullptr;
    if (h->timing && !h->split && call.mode == 0 && h->tev_used + 5 <= 5 * 512) {
        while (h->tev.size() < h->tev_used + 5) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return fail(FTL_E_DEVICE, "hipEventCreate"); h->tev.push_back(ev); }
        tev = h->tev.data() + h->tev_used; h->tev_used += 5;
    }
    auto launch_range = [&](int part, int parts, hipStream_t s) {
        const int epw0 = FTL_WAVE / h->G;
        const int n_groups = (h->P.n_envs + epw0 - 1) / epw0;
        const int my_groups = (n_groups - part + parts - 1) / parts;
        FtlCall c2 = call; c2.part = part; c2.parts = parts; c2.epw = epw0;
        const int count = my_groups * epw0;              // slots of this launch (the tail of the last group may be idle)
        const bool reg = h->P.cfg.n_speed_regime >= 0 || h->P.cfg.n_acc_regime >= 0 || h->P.cfg.rand_fps_hi > 0;
        const int epw = FTL_WAVE / h->G;
        size_t lds = (size_t)h->P.fr_lds + h->lds_pad;
        const dim3 grid((count + epw - 1) / epw), block(FTL_WAVE);
        if (tev) (void)hipEventRecord(tev[0], s);
        if (h->G == 4) {
            if (reg) hipLaunchKernelGGL((ftl_frames_group_kernel<4, true>), grid, block, lds, s, h->dP, c2);
            else hipLaunchKernelGGL((ftl_frames_group_kernel<4, false>), grid, block, lds, s, h->dP, c2);
        } else {
            if (reg) hipLaunchKernelGGL((ftl_frames_group_kernel<8, true>), grid, block, lds, s, h->dP, c2);
            else hipLaunchKernelGGL((ftl_frames_group_kernel<8, false>), grid, block, lds, s, h->dP, c2);
        }
        if (h->P.cfg.has_tracker == 1 && part == 0)      // the v1 tracker's scan + snapshot bookkeeping for ALL envs (one thread per env)
            hipLaunchKernelGGL(ftl::ftl_tracker1_kernel, dim3((unsigned)((h->P.n_envs + 255) / 256)), dim3(256), 0, s, h->dP, c2);
        if (tev) (void)hipEventRecord(tev[1], s);
        if (h->P.cfg.n_lasers > 0) {
            bool expl = false;
            for (int k = 0; k < h->P.cfg.n_lasers; k++) expl = expl || h->P.cfg.lasers[k].explicit_angles != 0 || h->P.cfg.lasers[k].pad_sectors != 0 || h->P.cfg.lasers[k].compas != 0;
            const dim3 rgrid(count);
            // CAPPED: the LDS copy of the corridor ring is smaller than the ring (corr_cap > 128, the configs with regimes or the v1 tracker):
            // the instantiations that carry the unstaged path
#define FTL_LAUNCH_RAYS(HM_, EXPL_, SPLIT_) do { \
                if (capped) hipLaunchKernelGGL((ftl_rays_kernel<HM_, EXPL_, SPLIT_, true>), rgrid, block, h->P.lds_rays, s, h->dP, c2); \
                else hipLaunchKernelGGL((ftl_rays_kernel<HM_, EXPL_, SPLIT_, false>), rgrid, block, h->P.lds_rays, s, h->dP, c2); } while (0)
            const bool capped = h->P.corr_lds_cap < h->P.cfg.corr_cap;
            if (parts > 1) {   // two-stream mode: the instantiations that map blocks to one half of the slot groups
                if (!expl && h->P.hmax > 5 && h->P.hmax <= 10) FTL_LAUNCH_RAYS(10, false, true);
                else if (h->P.hmax <= 5) FTL_LAUNCH_RAYS(5, true, true);
                else FTL_LAUNCH_RAYS(FTL_HMAX, true, true);
            } else if (expl) {        // LeaderCorridor_lasers or pad_sectors somewhere in the config: the two instantiations that carry that code
                if (h->P.hmax <= 5) FTL_LAUNCH_RAYS(5, true, false);
                else FTL_LAUNCH_RAYS(FTL_HMAX, true, false);
            } else if (h->P.hmax <= 5) FTL_LAUNCH_RAYS(5, false, false);
            else if (h->P.hmax <= 8) FTL_LAUNCH_RAYS(8, false, false);
            else if (h->P.hmax <= 10) FTL_LAUNCH_RAYS(10, false, false);   // the shipped training configs
            else FTL_LAUNCH_RAYS(FTL_HMAX, false, false);
#undef FTL_LAUNCH_RAYS
        }
        if (tev) (void)hipEventRecord(tev[2], s);
    };
    if (h->split) {
        (void)hipEventRecord(h->ev_fork, (hipStream_t)stream);              // everything the caller queued (actions, ...) happens first
        (void)hipStreamWaitEvent(h->side, h->ev_fork, 0);
        launch_range(0, 2, (hipStream_t)stream);
        launch_range(1, 2, h->side);
        (void)hipEventRecord(h->ev_join, h->side);
        (void)hipStreamWaitEvent((hipStream_t)stream, h->ev_join, 0);        // the caller's stream sees the whole step
    } else launch_range(0, 1, (hipStream_t)stream);
    {   // compas / lidar / leader-track detectors: one more launch, only for configs that have such a sensor
        bool aux = h->P.cfg.n_aux > 0;
        for (int k = 0; k < h->P.cfg.n_lasers; k++) aux = aux || h->P.cfg.lasers[k].compas != 0;
        if (aux) hipLaunchKernelGGL(ftl_aux_kernel, dim3((unsigned)h->P.n_envs), dim3(FTL_WAVE), ftl_aux_lds_bytes(h->P.cfg), (hipStream_t)stream, h->dP, call);
    }
    if (tev) (void)hipEventRecord(tev[3], (hipStream_t)stream);
    // the frame kernel left every env's cost class for its next step: rebuild the slot -> env map.  The classes are stable
    // from step to step unless the frame count is random, so every second launch is enough then.
    if (h->regroup && (h->P.cfg.rand_fps_hi > 0 || call.mode == 1 || (h->rg_launches++ % h->rg_every) == 0)) {
        const unsigned nb = (unsigned)((h->P.n_envs + FTL_RG_BLOCK - 1) / FTL_RG_BLOCK);
        int* tot = h->rg_tot + (h->rg_parity & 1u) * FTL_NKEYS, *tot_next = h->rg_tot + ((h->rg_parity + 1u) & 1u) * FTL_NKEYS;
        h->rg_parity++;
        hipLaunchKernelGGL(ftl::ftl_regroup_count_kernel, dim3(nb), dim3(FTL_RG_BLOCK), 0, (hipStream_t)stream, h->dP, tot);
        hipLaunchKernelGGL(ftl::ftl_regroup_scatter_kernel, dim3(nb), dim3(FTL_RG_BLOCK), 0, (hipStream_t)stream, h->dP, (const int*)tot, tot_next);
    }
    if (tev) (void)hipEventRecord(tev[4], (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    return FTL_OK;
}

static int check_out(const ftl_handle* h, const ftl_outputs* out) {
    if (!out || !out->obs_num || !out->target || !out->reward || !out->done || !out->status || (h->P.lasers_len > 0 && !out->lasers))
        return fail(FTL_E_INVALID, "output arrays missing");
    return FTL_OK;
}

int ftl_reset(ftl_handle* h, const int32_t* scen_idx, const uint8_t* mask, const ftl_outputs* out, void* stream) {
    if (!h || !scen_idx) return fail(FTL_E_INVALID, "null argument");
    if (!h->bound) return fail(FTL_E_STATE, "ftl_bind_state has not been called");
    if (!h->have_scen) return fail(FTL_E_STATE, "ftl_load_scenarios has not been called");
    int rc = check_out(h, out);
    if (rc) return rc;
    if (out->policy_obs && h->P.pol_h <= 0) return fail(FTL_E_INVALID, "policy_obs needs the same max_prev_obs on every ray sensor");
    FtlCall call; call.mode = 1; call.scen_idx = scen_idx; call.mask = mask; call.out = *out; call.action = nullptr; call.flags = 0; call.action_kind = FTL_ACTION_BOX2; call.win_base = h->win_base; call.win_count = h->win_count; call.win_stride = h->win_stride;
    return launch(h, call, stream);
}

int ftl_step(ftl_handle* h, const double* action, const ftl_outputs* out, uint32_t flags, void* stream) {
    return ftl_step_encoded(h, action, FTL_ACTION_BOX2, out, flags, stream);
}

int ftl_step_encoded(ftl_handle* h, const void* action, int32_t encoding, const ftl_outputs* out, uint32_t flags, void* stream) {
    if (!h || !action) return fail(FTL_E_INVALID, "null argument");
    if (encoding < FTL_ACTION_BOX2 || encoding > FTL_ACTION_TURN) return fail(FTL_E_INVALID, "unknown action encoding");
    if (!h->bound) return fail(FTL_E_STATE, "ftl_bind_state has not been called");
    if (!h->have_scen) return fail(FTL_E_STATE, "ftl_load_scenarios has not been called");
    int rc = check_out(h, out);
    if (rc) return rc;
    if (out->policy_obs && h->P.pol_h <= 0) return fail(FTL_E_INVALID, "policy_obs needs the same max_prev_obs on every ray sensor");
    FtlCall call; call.mode = 0; call.action = (const double*)action; call.action_kind = encoding; call.out = *out; call.flags = flags; call.scen_idx = nullptr; call.mask = nullptr; call.win_base = h->win_base; call.win_count = h->win_count; call.win_stride = h->win_stride;
    return launch(h, call, stream);
}

int ftl_kernel_timing(ftl_handle* h, int32_t enable) {
    if (!h) return fail(FTL_E_INVALID, "null argument");
    if (enable && h->split) return fail(FTL_E_UNSUPPORTED, "per-kernel timing is not available in the two-stream mode");
    h->timing = enable != 0; h->tev_used = 0;
    return FTL_OK;
}

int ftl_kernel_times(ftl_handle* h, double* ms, int32_t* n_steps) {
    if (!h || !ms || !n_steps) return fail(FTL_E_INVALID, "null argument");
    if (h->split) return fail(FTL_E_UNSUPPORTED, "per-kernel timing is not available in the two-stream mode");
    ms[0] = ms[1] = ms[2] = ms[3] = 0.0; *n_steps = 0;
    if (h->tev_used == 0) return FTL_OK;
    (void)hipSetDevice(h->device);
    hipError_t e = hipEventSynchronize(h->tev[h->tev_used - 1]);
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    for (size_t i = 0; i + 5 <= h->tev_used; i += 5) {
        for (int k = 0; k < 4; k++) {
            float t = 0.0f;
            e = hipEventElapsedTime(&t, h->tev[i + k], h->tev[i + k + 1]);
            if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipEventElapsedTime: ") + hipGetErrorString(e));
            ms[k] += (double)t;
        }
        *n_steps += 1;
    }
    h->tev_used = 0;
    return FTL_OK;
}

int ftl_episode_metrics(ftl_handle* h, double* dev_metrics, int32_t* dev_errors, uint32_t flags, void* stream) {
    if (!h || !dev_metrics) return fail(FTL_E_INVALID, "null argument");
    if (!h->bound) return fail(FTL_E_STATE, "ftl_bind_state has not been called");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    const int nb = (h->P.n_envs + FTL_MT_BLOCK - 1) / FTL_MT_BLOCK;
    const size_t o_err = align_up((size_t)nb * FTL_N_METRICS * sizeof(double), 256);
    if (!h->mt_mem) {
        e = hipMalloc(&h->mt_mem, o_err + (size_t)nb * 2 * sizeof(int));
        if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("hipMalloc(metrics): ") + hipGetErrorString(e));
    }
    { int rc = sync_params(h); if (rc) return rc; }
    double* part = (double*)h->mt_mem; int* epart = (int*)((char*)h->mt_mem + o_err);
    hipLaunchKernelGGL(ftl::ftl_metrics_partial_kernel, dim3((unsigned)nb), dim3(FTL_MT_THREADS), 0, (hipStream_t)stream, h->dP, part, epart, (flags & FTL_METRICS_CLEAR) ? 1 : 0);
    hipLaunchKernelGGL(ftl::ftl_metrics_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)part, (const int*)epart, nb, dev_metrics, dev_errors);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(FTL_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    return FTL_OK;
}

}  // extern "C"

#include "ftl_gazebo.hpp"      // follower-relative tracker / ray sensors (include/ftl_gazebo.h), same translation unit

#ifdef FTL_WAVE_TIMES
extern "C" int ftl_debug_wave_timeline(unsigned long long* times, unsigned int* info) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(times, HIP_SYMBOL(ftl::g_wt), sizeof(unsigned long long) * 2 * 8192) != hipSuccess ||
           hipMemcpyFromSymbol(info, HIP_SYMBOL(ftl::g_wi), sizeof(unsigned int) * 8192) != hipSuccess;
}
#endif
#ifdef FTL_PROFILE_PATHS
extern "C" int ftl_debug_heavy(unsigned long long* out, int clear) {      // [17] section cycles of the long wavefronts + their number
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ftl::g_cyc_heavy), sizeof(unsigned long long) * 17) != hipSuccess) return 1;
    if (clear) { unsigned long long z[17] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ftl::g_cyc_heavy), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
extern "C" int ftl_debug_wave_times(unsigned long long* out) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ftl::g_wave_t), sizeof(unsigned long long) * 2 * 8192) != hipSuccess;
}
extern "C" int ftl_debug_whist(unsigned int* out, int clear) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ftl::g_whist), sizeof(unsigned int) * 128) != hipSuccess) return 1;
    if (clear) { unsigned int z[128] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ftl::g_whist), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
extern "C" int ftl_debug_prof(unsigned long long* out, int clear) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ftl::g_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(ftl::g_cyc), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
#ifdef FTL_PROFILE_RAYS
    if (hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_rcyc), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (clear) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rcyc), z, sizeof(z)) != hipSuccess) return 1; }
#endif
    if (clear) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ftl::g_prof), z, sizeof(z)) != hipSuccess) return 1;
                 if (hipMemcpyToSymbol(HIP_SYMBOL(ftl::g_cyc), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

// ftl_frames_group.hpp -- the frame-loop kernel: G lanes per environment, 64 / G environments per wavefront (G = 4 for up to 2
// dynamic obstacles, G = 8 for up to 4).
//
// Lane (slot, r) holds robot r of env `slot` (0 leader, 1 follower, 2.. bears): the per-robot instruction stream (controller, f64
// sin/cos integrator, integer hitbox, steering) advances 16 (or 8) envs at once, per-env "scalars" are replicated across the G lanes
// of a group, and the list-shaped work (static-rect collisions, green-zone window, closest trajectory point, tracker sums) is strided
// over the G lanes of the group with group-local reductions (DPP quad_perm moves for G = 4).  Cross-lane traffic never leaves a group
// and is only issued where all lanes of the group are active (loops have group-uniform trip counts, r-dependent branches contain no
// shuffles), so env-level divergence between groups is safe.  (A one-wavefront-per-env version of this kernel was the round's first
// correct path -- 8.4 M env-steps/s, profiles/r01_a_* -- and is gone.)
//
// Semantics are those of oracle/ftl_oracle.c operation by operation; reference citations as there.
#pragma once
#include "ftl_device.hpp"

namespace ftl {

// broadcast the value lane K of this lane's group holds
template <int G, int K>
__device__ __forceinline__ int gb_i(int v) {
    if constexpr (G == 4) return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, false);    // quad_perm:[K,K,K,K]
    else if constexpr (G == 8) {
        // a group of 8 = two quads of one DPP row: every quad broadcasts its own lane K & 3, then the quad that does not hold lane K takes
        // the other quad's copy (row_shr:4 into banks 1 and 3, or row_shl:4 into banks 0 and 2) -- two DPP moves, no trip through the LDS unit
        const int t = __builtin_amdgcn_mov_dpp(v, (K & 3) * 0x55, 0xf, 0xf, false);
        if constexpr (K < 4) return __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xA, false);
        else return __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);
    } else return __shfl(v, (threadIdx.x & ~(G - 1)) + K);
}
template <int G, int K> __device__ __forceinline__ float gb_f(float v) { return __int_as_float(gb_i<G, K>(__float_as_int(v))); }
template <int G, int K> __device__ __forceinline__ double gb_d(double v) {
    return __hiloint2double(gb_i<G, K>(__double2hiint(v)), gb_i<G, K>(__double2loint(v)));
}
// value of lane (r ^ off) of this lane's group.  Offsets 1 and 2 never leave a quad: one DPP move (quad_perm) instead of a
// ds_bpermute round trip through the LDS unit -- this kernel is bound by dependent-instruction latency.
__device__ __forceinline__ int gx_i(int v, int off, int G) {
    if (off == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);      // quad_perm:[1,0,3,2]
    if (off == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);      // quad_perm:[2,3,0,1]
    if (off == 4) {                                                               // the two quads of a group of 8 swap: row_shr:4 / row_shl:4
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xA, false);     // banks 1, 3 <- banks 0, 2
        return __builtin_amdgcn_update_dpp(t, v, 0x104, 0xf, 0x5, false);            // banks 0, 2 <- banks 1, 3 (of the original)
    }
    return __shfl_xor(v, off, G);
}
__device__ __forceinline__ float gx_f(float v, int off, int G) { return __int_as_float(gx_i(__float_as_int(v), off, G)); }
__device__ __forceinline__ double gx_d(double v, int off, int G) { return __hiloint2double(gx_i(__double2hiint(v), off, G), gx_i(__double2loint(v), off, G)); }
__device__ __forceinline__ float gx(float v, int off, int G) { return gx_f(v, off, G); }
__device__ __forceinline__ int gx(int v, int off, int G) { return gx_i(v, off, G); }
__device__ __forceinline__ double gx(double v, int off, int G) { return gx_d(v, off, G); }
// value of lane (r - off) of the group (whatever for r < off), off = 1 or 2
template <int G> __device__ __forceinline__ double g_up_d(double v, int off) {
    if constexpr (G == 4) {
        const int hi = __double2hiint(v), lo = __double2loint(v);
        if (off == 1) return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x90, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x90, 0xf, 0xf, false));   // quad_perm:[0,0,1,2]
        return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x44, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x44, 0xf, 0xf, false));                  // quad_perm:[0,1,0,1]
    } else if constexpr (G == 8) {        // row_shr inside the DPP row (lanes r < off of a group get a neighbour group's value: "whatever", as documented)
        const int hi = __double2hiint(v), lo = __double2loint(v);
        if (off == 1) return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x111, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x111, 0xf, 0xf, false));
        if (off == 2) return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x112, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x112, 0xf, 0xf, false));
        return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x114, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x114, 0xf, 0xf, false));
    } else return __shfl_up(v, off, G);
}
// value of lane k (group-uniform, run-time) of the group
template <int G> __device__ __forceinline__ float g_pick_f(float v, int k) {
    if constexpr (G == 4) {
        const float a = gb_f<4, 0>(v), b = gb_f<4, 1>(v), c2 = gb_f<4, 2>(v), d = gb_f<4, 3>(v);
        return k == 0 ? a : k == 1 ? b : k == 2 ? c2 : d;
    } else return __shfl(v, k, G);
}
template <int G> __device__ __forceinline__ bool group_any(bool p) {
    unsigned long long m = __ballot(p);
    return ((m >> (threadIdx.x & ~(G - 1))) & ((1ull << G) - 1ull)) != 0ull;
}
template <int G> __device__ __forceinline__ double group_excl_scan(double v, int r, double& total) {   // exclusive prefix sum over the group
    double incl = v;
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        double o = g_up_d<G>(incl, off);
        if (r >= off) incl += o;
    }
    total = gb_d<G, G - 1>(incl);
    return incl - v;
}
template <int G> __device__ __forceinline__ int group_sum(int v) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += gx(v, off, G);
    return v;
}
template <int G> __device__ __forceinline__ void group_argmin(float& v, int& idx) {     // first-index ties (np.argmin)
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        float ov = gx(v, off, G);
        int oi = gx(idx, off, G);
        bool take = (ov < v) || (ov == v && oi < idx);
        if (take) { v = ov; idx = oi; }
    }
}

#ifdef FTL_WAVE_TIMES
// diagnostic build only (profiles/tools/wave_timeline.py): start / end of every frame-kernel wavefront of the last launch, nothing else
__device__ unsigned long long g_wt[2 * 8192];      // [wave][start, end] in 100 MHz ticks (s_memrealtime)
__device__ unsigned int g_wi[8192];                // [wave] bit 0: an env was reset, bits 8..: trajectory searches of its envs
#endif
#ifdef FTL_PROFILE_PATHS
// diagnostic build only (profiles/tools/path_counts.py): how often each branch of the frame is taken
__device__ unsigned long long g_prof[16];
#ifdef FTL_PROFILE_NOCOUNT      // cycle sections only: the contended counting atomics would distort them
#define FTL_PROF(slot, cond, n) do { } while (0)
#else
#define FTL_PROF(slot, cond, n) do { if (cond) atomicAdd(&g_prof[slot], (unsigned long long)(n)); } while (0)
#endif
__device__ unsigned long long g_cyc[16];
__device__ unsigned long long g_cyc_heavy[17];     // the same sections over the wavefronts that ran longer than 115 us; [16] = their number
__device__ unsigned long long g_wave_t[2 * 8192];    // [wave][start, end] in 100 MHz ticks (s_memrealtime), last launch
__device__ unsigned int g_whist[2][64];             // wave lifetime histogram (bins of 4096 cycles), [did a reset]
__shared__ unsigned long long s_cyc[16];           // per-wave accumulators, flushed once at the end of the kernel
#define FTL_TIC(slot) do { unsigned long long _t = __builtin_readcyclecounter(); if (threadIdx.x == 0) s_cyc[slot] += _t - _tprev; _tprev = _t; } while (0)
#define FTL_TIC_INIT unsigned long long _tprev = __builtin_readcyclecounter()
#else
#define FTL_PROF(slot, cond, n) do { } while (0)
#define FTL_TIC(slot) do { } while (0)
#define FTL_TIC_INIT do { } while (0)
#endif

// device twin of ftl_mix64 / ftl_uniform01 (include/ftl.h): the stream that replaces random.uniform at ENV:1156
__device__ __forceinline__ unsigned long long d_mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
__device__ __forceinline__ double d_uniform01(unsigned long long seed, unsigned long long env_id, unsigned long long resets, unsigned long long frame) {
    unsigned long long key = d_mix64(seed + 0x9E3779B97F4A7C15ULL * (env_id + 1)) ^ d_mix64(0xD1B54A32D192ED03ULL * (resets + 1));
    return (double)(d_mix64(key + 0x9E3779B97F4A7C15ULL * (frame + 1)) >> 11) * (1.0 / 9007199254740992.0);
}

// device twin of ftl_rand_frames (include/ftl.h): np.random.randint(lo, hi) of ENV:405 / 940
__device__ __forceinline__ int d_rand_frames(const ftl_config& c, int env, int resets, int step_count) {
    const double u = d_uniform01(c.rng_seed, (unsigned long long)(c.env_id_base + env), (unsigned long long)resets,
                                 (unsigned long long)step_count | (1ULL << 40));
    const int v = c.rand_fps_lo + (int)(u * (double)(c.rand_fps_hi - c.rand_fps_lo));
    return v < c.rand_fps_hi ? v : c.rand_fps_hi - 1;
}

// per-lane context: robot r of env `env`, env scalars replicated over the group
struct GCtx {
    int env, r, slot;
    bool valid;          // env < n_envs
    int scen, cur_target_id, leader_finished, done, crash, is_in_box, is_on_trace, too_close;
    int step_count, finish_timer, traj_len, trk_counter, corr_lo, corr_hi, seed_end, snap_count;
    int error, episodes, green_count, green_len, scan_ok, route_len, near_cnt, snap_head, hint, green_tiny, resets, acc_consumed;
#ifdef FTL_WAVE_TIMES
    int dbg_walk, dbg_full;   // diagnostic: exact green walks / whole-trajectory searches of this step
#endif
    int err_acc;         // error bits of episodes that ended inside this launch (g_reset clears `error`): OR-ed into FTL_EI_ERROR_STICKY
    int n_search;        // frames of this step in which this env needed a trajectory search (regrouping key only)
    int fps;             // frames of this step: cfg.frames_per_step, or the env's last draw under random_frames_per_step
    double acc_penalty, overall_reward, cur_tx, cur_ty;
    double cur_mult, cur_acc, cum_speed;   // leader regimes (ENV:412, 449, 591-592, 1143-1174)
    float hx, hy;        // coordinates of trajectory point `hint`
    float clr_g, clr_a;  // lower bounds: distance follower -> any green point / any trajectory point (see g_frame)
    double green_w;      // running length of the green-zone window (approximate; decisions near the threshold are re-derived exactly)
    Robot rb;
};

// The env scalars that no frame reads -- what the tail of frame_step, the tracker and the bookkeeping after the frame loop work on.
// step() loads them AFTER the loop (g_load_late): held across it they cost the loop ~20 registers, which the compiler paid for with
// scratch reloads inside the frames.  REG kernels keep the leader-regime state with the frames, which use it.
template <bool REG>
__device__ __forceinline__ void g_load_late(const FtlDevParams& P, GCtx& E) {
    const int* ei = rec_field(P.env_int, P, E.env);
    const double* ed = rec_field(P.env_dbl, P, E.env);
    E.done = ei[FTL_EI_DONE]; E.crash = ei[FTL_EI_CRASH]; E.is_in_box = ei[FTL_EI_IN_BOX]; E.is_on_trace = ei[FTL_EI_ON_TRACE];
    E.too_close = ei[FTL_EI_TOO_CLOSE]; E.finish_timer = ei[FTL_EI_FINISH_TIMER];
    E.trk_counter = ei[FTL_EI_TRK_COUNTER]; E.corr_lo = ei[FTL_EI_CORR_LO];
    E.corr_hi = ei[FTL_EI_CORR_HI]; E.seed_end = ei[FTL_EI_SEED_END]; E.snap_count = ei[FTL_EI_SNAP_COUNT];
    E.episodes = ei[FTL_EI_EPISODES]; E.snap_head = ei[FTL_EI_SNAP_HEAD];
    E.acc_penalty = ed[FTL_ED_ACC_PENALTY]; E.overall_reward = ed[FTL_ED_OVERALL_REWARD];
    if (!REG) {
        E.acc_consumed = ei[FTL_EI_ACC_CONSUMED];
        E.cur_mult = ed[FTL_ED_CUR_MULT]; E.cur_acc = ed[FTL_ED_CUR_ACC]; E.cum_speed = ed[FTL_ED_CUM_SPEED];
    }
}

// LATE = leave the fields of g_load_late<REG> to a later call
template <int G, bool LATE = false, bool REG = false>
__device__ __forceinline__ void g_load(const FtlDevParams& P, GCtx& E) {
    const int* ei = rec_field(P.env_int, P, E.env);
    const double* ed = rec_field(P.env_dbl, P, E.env);
    E.scen = ei[FTL_EI_SCEN]; E.cur_target_id = ei[FTL_EI_TARGET_ID]; E.leader_finished = ei[FTL_EI_LEADER_FINISHED];
    E.step_count = ei[FTL_EI_STEP_COUNT];
    E.traj_len = ei[FTL_EI_TRAJ_LEN];
    E.error = ei[FTL_EI_ERROR]; E.green_count = ei[FTL_EI_GREEN_COUNT]; E.green_len = ei[FTL_EI_GREEN_LEN];
    E.hint = ei[FTL_EI_HINT]; E.green_tiny = ei[FTL_EI_GREEN_TINY];
    E.resets = ei[FTL_EI_RESETS];
    E.fps = P.cfg.rand_fps_hi > 0 ? ei[FTL_EI_FPS] : P.cfg.frames_per_step;
    E.hx = __int_as_float(ei[FTL_EI_HINT_X]); E.hy = __int_as_float(ei[FTL_EI_HINT_Y]);
    E.clr_g = __int_as_float(ei[FTL_EI_CLR_GREEN]); E.clr_a = __int_as_float(ei[FTL_EI_CLR_ALL]);
    E.cur_tx = ed[FTL_ED_SPARE0]; E.cur_ty = ed[FTL_ED_SPARE1]; E.green_w = ed[FTL_ED_GREEN_W];
    if (!LATE || REG) {
        E.acc_consumed = ei[FTL_EI_ACC_CONSUMED];
        E.cur_mult = ed[FTL_ED_CUR_MULT]; E.cur_acc = ed[FTL_ED_CUR_ACC]; E.cum_speed = ed[FTL_ED_CUM_SPEED];
    }
    if (!LATE) g_load_late<true>(P, E);
    int rr = (E.r < P.R) ? E.r : 0;          // idle lanes mirror robot 0 (never committed)
    const float* rp = rec_field(P.rb_pos, P, E.env) + 2 * rr;
    E.rb.px = rp[0]; E.rb.py = rp[1];
    const double* rd = rec_field(P.rb_dbl, P, E.env) + rr * FTL_RD_COUNT;
    E.rb.direction = rd[FTL_RD_DIRECTION]; E.rb.speed = rd[FTL_RD_SPEED]; E.rb.rot_speed = rd[FTL_RD_ROT_SPEED];
    E.rb.des_speed = rd[FTL_RD_DES_SPEED]; E.rb.des_rot_speed = rd[FTL_RD_DES_ROT_SPEED];
    const int4* ri = reinterpret_cast<const int4*>(rec_field(P.rb_int, P, E.env) + rr * FTL_RI_COUNT);
    int4 r0 = ri[0], r1 = ri[1];
    E.rb.rx = r0.x; E.rb.ry = r0.y; E.rb.rw = r0.z; E.rb.rh = r0.w; E.rb.rot_dir = r1.x; E.rb.des_rot_dir = r1.y;
    int b = (E.r >= 2 && E.r < P.R) ? E.r - 2 : 0;
    E.rb.tgt_x = ed[FTL_ED_BEAR_POINTS + 2 * b]; E.rb.tgt_y = ed[FTL_ED_BEAR_POINTS + 2 * b + 1];
    E.rb.dyn_index = ei[FTL_EI_DYN_INDEX0 + b];
    E.route_len = P.scen.route_len[E.scen];
}

template <int G>
__device__ __forceinline__ void g_store(const FtlDevParams& P, GCtx& E) {
    if (!E.valid) return;
    int* ei = rec_field(P.env_int, P, E.env);
    double* ed = rec_field(P.env_dbl, P, E.env);
    if (E.r == 0) {
        ei[FTL_EI_SCEN] = E.scen; ei[FTL_EI_TARGET_ID] = E.cur_target_id; ei[FTL_EI_LEADER_FINISHED] = E.leader_finished;
        ei[FTL_EI_DONE] = E.done; ei[FTL_EI_CRASH] = E.crash; ei[FTL_EI_IN_BOX] = E.is_in_box; ei[FTL_EI_ON_TRACE] = E.is_on_trace;
        ei[FTL_EI_TOO_CLOSE] = E.too_close; ei[FTL_EI_STEP_COUNT] = E.step_count; ei[FTL_EI_FINISH_TIMER] = E.finish_timer;
        ei[FTL_EI_TRAJ_LEN] = E.traj_len; ei[FTL_EI_TRK_COUNTER] = E.trk_counter; ei[FTL_EI_CORR_LO] = E.corr_lo;
        ei[FTL_EI_CORR_HI] = E.corr_hi; ei[FTL_EI_SEED_END] = E.seed_end; ei[FTL_EI_SNAP_COUNT] = E.snap_count;
        ei[FTL_EI_ERROR] = E.error; ei[FTL_EI_EPISODES] = E.episodes; ei[FTL_EI_GREEN_COUNT] = E.green_count; ei[FTL_EI_GREEN_LEN] = E.green_len;
        ei[FTL_EI_SCAN_OK] = E.scan_ok; ei[FTL_EI_SNAP_HEAD] = E.snap_head; ei[FTL_EI_HINT] = E.hint; ei[FTL_EI_GREEN_TINY] = E.green_tiny;
        ei[FTL_EI_RESETS] = E.resets; ei[FTL_EI_ACC_CONSUMED] = E.acc_consumed; ei[FTL_EI_FPS] = E.fps;
        ei[FTL_EI_HINT_X] = __float_as_int(E.hx); ei[FTL_EI_HINT_Y] = __float_as_int(E.hy);
        ei[FTL_EI_CLR_GREEN] = __float_as_int(E.clr_g); ei[FTL_EI_CLR_ALL] = __float_as_int(E.clr_a);
        ed[FTL_ED_CUR_MULT] = E.cur_mult; ed[FTL_ED_CUR_ACC] = E.cur_acc; ed[FTL_ED_CUM_SPEED] = E.cum_speed;
        ed[FTL_ED_ACC_PENALTY] = E.acc_penalty; ed[FTL_ED_OVERALL_REWARD] = E.overall_reward;
        ed[FTL_ED_SPARE0] = E.cur_tx; ed[FTL_ED_SPARE1] = E.cur_ty; ed[FTL_ED_GREEN_W] = E.green_w;
        // the sticky error word is only ever touched when something went wrong (no load / store on the clean path)
        if ((E.err_acc | E.error) != 0) atomicOr(&ei[FTL_EI_ERROR_STICKY], E.err_acc | E.error);
    }
    if (E.r < P.R) {
        float* rp = rec_field(P.rb_pos, P, E.env) + 2 * E.r;
        rp[0] = E.rb.px; rp[1] = E.rb.py;
        double* rd = rec_field(P.rb_dbl, P, E.env) + E.r * FTL_RD_COUNT;
        rd[FTL_RD_DIRECTION] = E.rb.direction; rd[FTL_RD_SPEED] = E.rb.speed; rd[FTL_RD_ROT_SPEED] = E.rb.rot_speed;
        rd[FTL_RD_DES_SPEED] = E.rb.des_speed; rd[FTL_RD_DES_ROT_SPEED] = E.rb.des_rot_speed;
        int4* ri = reinterpret_cast<int4*>(rec_field(P.rb_int, P, E.env) + E.r * FTL_RI_COUNT);
        ri[0] = make_int4(E.rb.rx, E.rb.ry, E.rb.rw, E.rb.rh); ri[1] = make_int4(E.rb.rot_dir, E.rb.des_rot_dir, 0, 0);
        if (E.r >= 2) {
            int b = E.r - 2;
            ed[FTL_ED_BEAR_POINTS + 2 * b] = E.rb.tgt_x; ed[FTL_ED_BEAR_POINTS + 2 * b + 1] = E.rb.tgt_y;
            ei[FTL_EI_DYN_INDEX0 + b] = E.rb.dyn_index;
        }
    }
}

// reset(): ENV:494-543 from scenario `scen`; executed by the groups whose `go` is set (group-uniform)
template <int G>
__device__ __forceinline__ void g_reset(const FtlDevParams& P, GCtx& E, int scen, bool go) {
    const ftl_config& c = P.cfg;
    int init_n0 = 0; const float2* init_src = nullptr; float2* init_dst = nullptr;
    double2 rt01[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};      // the route's first two way-points: requested with the rest of the
    if (go) {                                                                // scenario's header (one memory round trip), chosen below
        const double2* rt = reinterpret_cast<const double2*>(P.scen.route + (size_t)scen * c.route_cap * 2);
        rt01[0] = rt[0]; rt01[1] = rt[1];
        E.scen = scen;
        int rr = (E.r < P.R) ? E.r : 0;
        size_t so = (size_t)scen * P.R + rr;
        E.rb.px = P.scen.robot_pos[2 * so]; E.rb.py = P.scen.robot_pos[2 * so + 1];
        E.rb.direction = P.scen.robot_dir[so];
        E.rb.speed = 0; E.rb.rot_speed = 0; E.rb.des_speed = 0; E.rb.des_rot_speed = 0; E.rb.rot_dir = 0; E.rb.des_rot_dir = 0;
        const int4 rr4 = reinterpret_cast<const int4*>(P.scen.robot_rect)[so];
        E.rb.rx = rr4.x; E.rb.ry = rr4.y; E.rb.rw = rr4.z; E.rb.rh = rr4.w;
        E.route_len = P.scen.route_len[scen];
        int n0 = P.scen.init_traj_len[scen];              // initial leader_factual_trajectory (ENV:533-539)
        const float2* src = reinterpret_cast<const float2*>(P.scen.init_traj) + (size_t)scen * c.init_traj_cap;
        float2* dst = reinterpret_cast<float2*>(P.traj + (size_t)E.env * c.traj_cap * 2);
        E.traj_len = n0;
        init_n0 = n0; init_src = src; init_dst = dst;       // copied below, block by block, by the whole group
        E.step_count = 0; E.acc_penalty = 0; E.overall_reward = 0;
        E.done = 0; E.crash = 0; E.is_in_box = 0; E.is_on_trace = 0; E.too_close = 0;
        E.cur_target_id = 1; E.leader_finished = 0; E.finish_timer = -1;
        E.green_count = 0; E.green_len = -1; E.green_w = 0.0; E.green_tiny = 0; E.error = 0; E.scan_ok = 0;
        E.trk_counter = 0; E.corr_lo = 0; E.corr_hi = 0; E.seed_end = 0; E.snap_count = 0; E.snap_head = 0;
        E.hint = 0; E.hx = 3.0e38f; E.hy = 3.0e38f; E.clr_g = 0.0f; E.clr_a = 0.0f;     // no cached point, no bound
        if (c.has_tracker == 1 && E.r == 0) rec_field(P.env_int, P, E.env)[FTL_EI_HIST1_LEN] = 0;   // v1 tracker reset(), SEN:223-226
        if (c.rand_fps_hi > 0 && E.fps == 0) E.fps = d_rand_frames(c, E.env, 0, 0);      // the constructor's draw (ENV:405)
        E.cur_mult = 1.0; E.cur_acc = 0.0; E.cum_speed = 0.0; E.resets += 1;      // ENV:449, 591-592; acc_consumed persists (ENV:1170)
    }
    // The initial trajectory (83-261 points).  When only a few envs of the wavefront reset -- the in-kernel auto-reset: one in most such
    // steps -- the WHOLE wavefront copies them, one env at a time: up to 192 points per memory round trip instead of 32 per trip by the
    // env's four lanes (five dependent trips: a third of what a reset cost its wavefront, and a wavefront with a reset is what a small
    // batch's frame kernel waits for).  With many (an explicit reset of every env) the groups copy side by side as before.
    const unsigned long long rmask = __ballot(go && E.r == 0);
    const bool wide_copy = __popcll(rmask) <= 4;
    if (wide_copy) {
        unsigned long long need = rmask;
        while (need) {
            const int Lr = __ffsll((long long)need) - 1; need &= need - 1;
            const int envL = __builtin_amdgcn_readlane(E.env, Lr), scenL = __builtin_amdgcn_readlane(scen, Lr), n0L = __builtin_amdgcn_readlane(init_n0, Lr);
            const float2* src = reinterpret_cast<const float2*>(P.scen.init_traj) + (size_t)scenL * c.init_traj_cap;
            float2* dst = reinterpret_cast<float2*>(P.traj + (size_t)envL * c.traj_cap * 2);
            float4* bbw = reinterpret_cast<float4*>(P.traj_bb) + (size_t)envL * (c.traj_cap / FTL_TRAJ_BLOCK);
            const int lane = threadIdx.x & (FTL_WAVE - 1);
            for (int base = 0; base < n0L; base += 3 * FTL_WAVE) {
                float2 q[3];
#pragma unroll
                for (int t = 0; t < 3; t++) { const int k = base + t * FTL_WAVE + lane; q[t] = src[k < n0L ? k : n0L - 1]; }
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int k = base + t * FTL_WAVE + lane;
                    const bool v = k < n0L;
                    if (v) dst[k] = q[t];
                    float4 box = v ? make_float4(q[t].x, q[t].y, q[t].x, q[t].y) : make_float4(3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f);
#pragma unroll
                    for (int off = FTL_TRAJ_BLOCK / 2; off >= 1; off >>= 1) {      // a block of 32 points = half a wavefront
                        box.x = fminf(box.x, __shfl_xor(box.x, off)); box.y = fminf(box.y, __shfl_xor(box.y, off));
                        box.z = fmaxf(box.z, __shfl_xor(box.z, off)); box.w = fmaxf(box.w, __shfl_xor(box.w, off));
                    }
                    const int k0 = base + t * FTL_WAVE + (lane & ~(FTL_TRAJ_BLOCK - 1));     // first point of this lane's block
                    if ((lane & (FTL_TRAJ_BLOCK - 1)) == 0 && k0 < n0L) bbw[k0 / FTL_TRAJ_BLOCK] = box;
                }
            }
        }
    }
    // group-uniform from here on (go is the same in every lane of a group), so the broadcasts are safe
    if (go && !wide_copy) {
        // The initial trajectory, one block of FTL_TRAJ_BLOCK points at a time: every lane loads its 32 / G points (all loads in flight
        // before the first store -- a load / store loop of possibly aliasing pointers pays one memory round trip per point, 20 us per
        // reset), stores them, and the block's bounding box (search acceleration, see g_range_argmin) comes from the same registers by a
        // min / max over the group.
        float4* bb = reinterpret_cast<float4*>(P.traj_bb) + (size_t)E.env * (c.traj_cap / FTL_TRAJ_BLOCK);
        for (int b = 0; b * FTL_TRAJ_BLOCK < init_n0; b++) {
            float2 q[FTL_TRAJ_BLOCK / G];
#pragma unroll
            for (int t = 0; t < FTL_TRAJ_BLOCK / G; t++) {
                const int k = b * FTL_TRAJ_BLOCK + t * G + E.r;
                q[t] = init_src[k < init_n0 ? k : init_n0 - 1];
            }
            float4 box = make_float4(3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f);
#pragma unroll
            for (int t = 0; t < FTL_TRAJ_BLOCK / G; t++) {
                const int k = b * FTL_TRAJ_BLOCK + t * G + E.r;
                if (k < init_n0) {
                    init_dst[k] = q[t];
                    box.x = fminf(box.x, q[t].x); box.y = fminf(box.y, q[t].y); box.z = fmaxf(box.z, q[t].x); box.w = fmaxf(box.w, q[t].y);
                }
            }
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) {
                box.x = fminf(box.x, gx(box.x, off, G)); box.y = fminf(box.y, gx(box.y, off, G));
                box.z = fmaxf(box.z, gx(box.z, off, G)); box.w = fmaxf(box.w, gx(box.w, off, G));
            }
            if (E.r == 0) bb[b] = box;
        }
    }
    float lpx = gb_f<G, 0>(E.rb.px), lpy = gb_f<G, 0>(E.rb.py);
    if (go) {
        if (E.route_len == 0) { E.done = 1; E.cur_tx = (double)lpx; E.cur_ty = (double)lpy; }
        else { const double2 w = E.route_len > 1 ? rt01[1] : rt01[0]; E.cur_tx = w.x; E.cur_ty = w.y; }
        // ENV:717-718: every bear starts from the LAST bear_start_position, (leader - 150, leader - 150) in float32
        E.rb.tgt_x = (double)(lpx - 150.0f); E.rb.tgt_y = (double)(lpy - 150.0f); E.rb.dyn_index = 0;
    }
}

// static rects that can touch the follower or the leader during this step -> compact per-env list in LDS.
// A robot's hitbox stays inside pos +- ((w+h)/2 + 1) and pos moves at most frames*max_speed per step, so a rect outside
// that swept box (plus slack) can never collide during the step: dropping it from the per-frame tests is exact.
template <int G>
__device__ __forceinline__ void g_build_near(const FtlDevParams& P, GCtx& E, int4* s_near, int* s_cnt, float4* s_box) {
    const ftl_config& c = P.cfg;
    const float lpx = gb_f<G, 0>(E.rb.px), lpy = gb_f<G, 0>(E.rb.py), fpx = gb_f<G, 1>(E.rb.px), fpy = gb_f<G, 1>(E.rb.py);
    const float F = (float)(c.rand_fps_hi > 0 ? c.rand_fps_hi : c.frames_per_step);     // most frames a step can have
    const float ml = 0.5f * (float)(c.leader.img_w + c.leader.img_h) + 4.0f + F * (float)fmax(fabs(c.leader.max_speed), fabs(c.leader.min_speed));
    const float mf = 0.5f * (float)(c.follower.img_w + c.follower.img_h) + 4.0f + F * (float)fmax(fabs(c.follower.max_speed), fabs(c.follower.min_speed));
    if (E.r == 0) s_cnt[E.slot] = 0;
    __syncthreads();
    if (E.valid) {
        const int4* src = reinterpret_cast<const int4*>(P.scen.static_rects) + (size_t)E.scen * c.n_static;
        int4* dst = s_near + (size_t)E.slot * c.n_static;
        // the bounding box of the trajectory block that the next appended point falls into travels with this round trip: the frames
        // that append a point update it in LDS (s_box) instead of reading it back from memory, a dependent round trip each
        const float4 box0 = (reinterpret_cast<const float4*>(P.traj_bb) + (size_t)E.env * (c.traj_cap / FTL_TRAJ_BLOCK))[min(E.traj_len, c.traj_cap - 1) / FTL_TRAJ_BLOCK];
        constexpr int SU = 8;                            // rects per lane in flight at once
        for (int s0 = 0; s0 < c.n_static; s0 += SU * G) {
            int4 qv[SU];
#pragma unroll
            for (int k = 0; k < SU; k++) { const int s = s0 + k * G + E.r; qv[k] = s < c.n_static ? src[s] : make_int4(0, 0, 0, 0); }
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const int s = s0 + k * G + E.r;
                const int4 q = qv[k];
                float x0 = (float)q.x, x1 = (float)(q.x + q.z), y0 = (float)q.y, y1 = (float)(q.y + q.w);
                bool nl = !(x1 < lpx - ml || x0 > lpx + ml || y1 < lpy - ml || y0 > lpy + ml);
                bool nf = !(x1 < fpx - mf || x0 > fpx + mf || y1 < fpy - mf || y0 > fpy + mf);
                if (s < c.n_static && (nl || nf)) dst[atomicAdd(&s_cnt[E.slot], 1)] = q;
            }
        }
        if (E.r == 0) s_box[E.slot] = box0;
    }
    __syncthreads();
    E.near_cnt = s_cnt[E.slot];
}

// ENV:1828-1843 sequential form for one env, run by every lane of the group redundantly (rare path)
__device__ __forceinline__ int g_green_seq(const float2* tr, int n, double maxd) {
    double acc = 0.0; int Gc = 0;
    for (int k = n - 2; k >= 0; k--) {
        float2 cur = tr[k], prev = tr[k + 1];
        acc += euclid_f32(prev.x, prev.y, cur.x, cur.y);
        if (acc <= maxd) Gc++; else break;
    }
    return Gc;
}
// group-parallel prefix form with the same exactness argument as green_walk() of the first-generation kernel
// Segment lengths are float32 values.  While every non-zero one is >= kTinySeg, all of them are multiples of 2^-43 and
// every partial sum stays below 2^9, so float64 addition of them is EXACT: any summation order reproduces the
// reference's sequential sums bit for bit.  Only a window that holds a shorter segment needs the tolerance band.
static constexpr double kTinySeg = 9.5367431640625e-07;   // 2^-20
template <int G>
__device__ __forceinline__ int g_green_walk(const FtlDevParams& P, const GCtx& E, double& wlen, int& tiny) {
    const float2* tr = reinterpret_cast<const float2*>(P.traj + (size_t)E.env * P.cfg.traj_cap * 2);
    const double maxd = P.cfg.max_distance;
    const int n = E.traj_len, cnt = n - 1;          // elements i = 0..cnt-1 <-> points (n-2-i, n-1-i)
    double base = 0.0, wmax = 0.0; int Gc = 0; bool near = false, small = false;
    for (int it0 = 0; it0 * G < cnt; it0 += 4) {    // group-uniform trip count; 4 sub-steps per trip so that their
        float2 cur[4], prev[4];                     // loads are in flight together (one memory round trip per trip)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int i = (it0 + u) * G + E.r;
            int ic = i < cnt ? i : cnt - 1;
            cur[u] = tr[n - 2 - ic]; prev[u] = tr[n - 1 - ic];
        }
        bool done = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int i = (it0 + u) * G + E.r;
            bool v = i < cnt && !done;
            double d = v ? euclid_f32(prev[u].x, prev[u].y, cur[u].x, cur[u].y) : 0.0;
            double tot;
            double pre = group_excl_scan<G>(d, E.r, tot);
            double val = base + pre + d;
            near |= v && (fabs(val - maxd) < 1e-6);
            small |= v && d != 0.0 && d < kTinySeg;
            Gc += (v && val <= maxd) ? 1 : 0;
            if (v && val <= maxd) wmax = fmax(wmax, val);
            base += tot;
            if (!(base <= maxd)) done = true;       // base is group-uniform
        }
        if (done) break;
    }
    Gc = group_sum<G>(Gc);
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) wmax = fmax(wmax, gx(wmax, off, G));
    tiny = group_any<G>(small) || maxd > 500.0;
    if (tiny && group_any<G>(near)) {
        Gc = g_green_seq(tr, n, maxd);
        wmax = 0.0;
        for (int i = 0; i < Gc; i++) { float2 cur = tr[n - 2 - i], prev = tr[n - 1 - i]; wmax += euclid_f32(prev.x, prev.y, cur.x, cur.y); }
    }
    wlen = wmax;
    return Gc;
}

// The same walk by the WHOLE wavefront for one env (tr, n wave-uniform): 64 segments per pass, an inclusive scan over the wavefront.
// One env's walk costs ~3 passes instead of ~10 group passes -- and the other fifteen envs of the wavefront used to wait through those.
// Exactness as above: while no tiny segment is involved every partial sum is exact, whatever the order of the additions.
struct WalkRes { int Gc; double wlen; int tiny; };
__device__ __forceinline__ WalkRes w_green_walk(const float2* tr, int n, double maxd) {
    const int lane = threadIdx.x & (FTL_WAVE - 1), cnt = n - 1;          // elements i = 0..cnt-1 <-> points (n-2-i, n-1-i)
    double base = 0.0, wmax = 0.0; int Gc = 0; bool near = false, small = false;
    for (int i0 = 0; i0 < cnt; i0 += FTL_WAVE) {
        const int i = i0 + lane, ic = i < cnt ? i : cnt - 1;
        const float2 cur = tr[n - 2 - ic], prev = tr[n - 1 - ic];
        const bool v = i < cnt;
        const double d = v ? euclid_f32(prev.x, prev.y, cur.x, cur.y) : 0.0;
        double incl = d;
#pragma unroll
        for (int off = 1; off < FTL_WAVE; off <<= 1) {
            const double o = __hiloint2double(__shfl_up(__double2hiint(incl), off), __shfl_up(__double2loint(incl), off));
            if (lane >= off) incl += o;
        }
        const double val = base + incl;
        near |= v && (fabs(val - maxd) < 1e-6);
        small |= v && d != 0.0 && d < kTinySeg;
        const unsigned long long in = __ballot(v && val <= maxd);          // a prefix of the lanes: the sums do not decrease
        const int k = __popcll(in);
        if (k > 0) wmax = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(val), k - 1), __builtin_amdgcn_readlane(__double2loint(val), k - 1));
        Gc += k;
        base += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(incl), FTL_WAVE - 1), __builtin_amdgcn_readlane(__double2loint(incl), FTL_WAVE - 1));
        if (!(base <= maxd)) break;
    }
    WalkRes r;
    r.tiny = (__ballot(small) != 0ull) || maxd > 500.0;
    if (r.tiny && __ballot(near) != 0ull) {
        Gc = g_green_seq(tr, n, maxd);
        wmax = 0.0;
        for (int i = 0; i < Gc; i++) { float2 cur = tr[n - 2 - i], prev = tr[n - 1 - i]; wmax += euclid_f32(prev.x, prev.y, cur.x, cur.y); }
    }
    r.Gc = Gc; r.wlen = wmax;
    return r;
}

// first-index arg-min of the f32 squared distance to (px,py) over `n` trajectory points, point i = tr[base + i*stride]
template <int G>
__device__ __forceinline__ int g_closest(const float2* tr, int r, float px, float py, int base, int stride, int n) {
    float best = __int_as_float(0x7f800000); int bi = 0x7fffffff;
#pragma unroll 8
    for (int i = r; i < n; i += G) {
        float2 q = tr[base + i * stride];
        float dx = q.x - px, dy = q.y - py;
        float d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; bi = i; }
    }
    group_argmin<G>(best, bi);
    return bi;
}

// arg-min (first index on ties, as np.argmin) of the float32 squared distance to (px,py) over trajectory points
// [lo, hi), restricted to the blocks whose bounding box comes within sqrt(thr2) of the query.  Every point with
// d2 <= thr2 lies in such a block, so the result IS the reference's arg-min whenever that arg-min has d2 <= thr2, and
// "no candidate" (idx == 0x7fffffff) or a result with d2 > thr2 means the reference's arg-min is beyond the threshold
// too -- which is all the callers need (they compare the arg-min's distance with a threshold <= sqrt(thr2)).
// `reversed`: the reference enumerates the range from hi-1 down to lo (the green list), ties go to the HIGHER index.
#ifndef FTL_BOX_UNROLL
#define FTL_BOX_UNROLL 2
#endif
template <int G>
__device__ __forceinline__ void g_range_argmin(const float2* tr, const float4* bb, int r, float px, float py, int lo, int hi,
                                               float thr2, bool reversed, int init, float init_d2, float2 init_p, float& best, int& bi, float2& bp, float& skipmin) {
    best = __int_as_float(0x7f800000); bi = 0x7fffffff; bp = init_p;     // the arg-min's coordinates travel with it: no re-load
    int key = 0x7fffffff;                                     // enumeration order of the reference (smaller = earlier)
    skipmin = __int_as_float(0x7f800000);                     // smallest box distance^2 among the blocks NOT scanned
    float bound = thr2;                                       // blocks farther than this cannot hold the wanted arg-min
    if (init >= lo && init < hi) {                            // a member of the range whose distance is already known
        if (r == 0) { best = init_d2; bi = init; key = reversed ? (hi - 1 - init) : init; }   // (from the hint window, with init_p):
        bound = fminf(bound, init_d2);                         // it bounds the minimum
    }
    const int b_lo = lo / FTL_TRAJ_BLOCK, b_hi = (hi - 1) / FTL_TRAJ_BLOCK;
    // The boxes are walked FTL_BOX_UNROLL x G at a time.  Their loads carry no guard (the index is clamped, the result masked), so they
    // are issued back to back and arrive together: a whole-trajectory search (20-35 boxes) walks them in two memory round trips.
    for (int b00 = b_lo; b00 <= b_hi; b00 += FTL_BOX_UNROLL * G) {      // group-uniform trip count
        float4 bx[FTL_BOX_UNROLL];
#pragma unroll
        for (int u = 0; u < FTL_BOX_UNROLL; u++) bx[u] = bb[min(b00 + u * G + r, b_hi)];
        float bd2u[FTL_BOX_UNROLL];
#pragma unroll
        for (int u = 0; u < FTL_BOX_UNROLL; u++) {
            const float dx = fmaxf(fmaxf(bx[u].x - px, px - bx[u].z), 0.0f), dy = fmaxf(fmaxf(bx[u].y - py, py - bx[u].w), 0.0f);
            bd2u[u] = (b00 + u * G + r <= b_hi) ? dx * dx + dy * dy - 1e-2f : __int_as_float(0x7f800000);   // slack far above the float32 rounding of the box test
        }
#pragma unroll
        for (int u = 0; u < FTL_BOX_UNROLL; u++) {
        const int b0 = b00 + u * G;
        const float bd2 = bd2u[u];
        if (b0 > b_hi) break;                                     // group-uniform
        unsigned m = (unsigned)((__ballot(bd2 <= bound) >> (threadIdx.x & ~(G - 1))) & ((1ull << G) - 1ull));
        if (!(bd2 <= bound)) skipmin = fminf(skipmin, bd2);   // per lane; combined over the group at the end
        while (m) {                                           // group-uniform: every lane of the group sees the same mask
            int k = __ffs(m) - 1; m &= m - 1;
            float kd2 = g_pick_f<G>(bd2, k);
            if (!(kd2 <= bound)) { skipmin = fminf(skipmin, kd2); continue; }   // the bound may have tightened since the mask was formed
            int s0 = (b0 + k) * FTL_TRAJ_BLOCK;
            int i0 = max(s0, lo), i1 = min(s0 + FTL_TRAJ_BLOCK, hi);
            float2 qq[FTL_TRAJ_BLOCK / G];
#pragma unroll
            for (int t = 0; t < FTL_TRAJ_BLOCK / G; t++) qq[t] = tr[min(i0 + r + t * G, i1 - 1)];      // unguarded, in flight together
#pragma unroll
            for (int t = 0; t < FTL_TRAJ_BLOCK / G; t++) {
                const int i = i0 + r + t * G;
                if (i < i1) {
                    float ddx = qq[t].x - px, ddy = qq[t].y - py;
                    float d2 = ddx * ddx + ddy * ddy;
                    int ky = reversed ? (hi - 1 - i) : i;
                    if (d2 < best || (d2 == best && ky < key)) { best = d2; bi = i; key = ky; bp = qq[t]; }
                }
            }
            float gbest = best;                               // tighten the bound with what the group has seen so far
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) gbest = fminf(gbest, gx(gbest, off, G));
            bound = fminf(bound, gbest);
        }
        }
    }
    // combine the lanes: smallest d2, then earliest in the reference's enumeration order
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        float ov = gx(best, off, G); int ok = gx(key, off, G); int oi = gx(bi, off, G);
        float ox = gx(bp.x, off, G), oy = gx(bp.y, off, G);
        bool take = (ov < best) || (ov == best && ok < key);
        if (take) { best = ov; key = ok; bi = oi; bp.x = ox; bp.y = oy; }
        skipmin = fminf(skipmin, gx(skipmin, off, G));
    }
}

// What _check_agent_position's cached point and distance bounds already decide (see g_frame): 0 = search needed,
// 1 = a green point within epsilon, 2 = none within epsilon but one within max_dev, 3 = nothing in reach, 4 = no green
// point in reach but a trajectory point within epsilon.  `quiet` = the caches also outlast a whole step of `fps` frames
// (the follower moves at most `reach`, the green window drops at most a few points).
__device__ __forceinline__ int g_cache_class(const ftl_config& c, const GCtx& E, float fpx, float fpy, int fps, bool want_quiet, bool& quiet) {
    const int n = E.traj_len, g_lo = n - 1 - E.green_count;
    const double eps = c.leader_pos_epsilon, mdev = c.max_dev, far = fmax(mdev, eps);
    // "some point is clearly within epsilon" => the arg-min point (smallest float32 squared distance) is too
    const float eps2_lo = (float)(eps * eps * (1.0 - 1e-5)), dev2_lo = (float)(mdev * mdev * (1.0 - 1e-5));
    const float eps_hi = (float)(eps * (1.0 + 1e-5)) + 1e-3f, far_hi = (float)(far * (1.0 + 1e-5)) + 1e-3f;
    const bool h_green = E.hint >= g_lo && E.hint <= n - 2;
    const float dx = E.hx - fpx, dy = E.hy - fpy, hd2 = dx * dx + dy * dy;
    int fast = 0;
    if (h_green && hd2 < eps2_lo) fast = 1;
    else if (h_green && hd2 < dev2_lo && E.clr_g > eps_hi) fast = 2;
    else if (E.clr_g > far_hi) fast = (E.clr_a > eps_hi) ? 3 : (hd2 < eps2_lo ? 4 : 0);
    quiet = false;
    if (!want_quiet) return fast;                  // (the per-frame caller needs it in the first frame of a step only)
    const float reach = (float)fps * (float)fmax(fabs(c.follower.max_speed), fabs(c.follower.min_speed)) * 1.001f + 1e-3f;
    const float hd = sqrtf(hd2);
    const bool h_stays_green = E.hint >= g_lo + 2 + fps / c.trajectory_saving_period;
    const float eps_lo = (float)(eps * (1.0 - 1e-5)), dev_lo = (float)(mdev * (1.0 - 1e-5));
    if (fast == 1) quiet = h_stays_green && hd + reach < eps_lo;
    else if (fast == 2) quiet = h_stays_green && hd + reach < dev_lo && E.clr_g - reach > eps_hi;
    else if (fast != 0) quiet = E.clr_g - reach > far_hi && (E.clr_a - reach > eps_hi || hd + reach < eps_lo);
    else quiet = false;
    return fast;
}

// One frame leaves a one-byte record per env for the tail of frame_step (g_tail) ...
enum { FR_TOO_CLOSE = 1, FR_IN_BOX = 2, FR_ON_TRACE = 4, FR_COLLISION = 8, FR_LEADER_HIT = 16, FR_LEADER_FINISHED = 32, FR_TOO_FAR = 64,
       FR_PENDING = 128 };   // FR_PENDING: in box / on trace are still to be searched for (g_resolve)
// ... and, when its position check needs a search, a pending item: the follower's position, the trajectory length and the green count
// as that frame saw them (the trajectory only grows, so [0, n) is still there later), the cached point index, frame << 4 | slot
struct FrPend { float fpx, fpy; unsigned short n, gc, hint, key; };

// REG = the config has leader regimes or random_frames_per_step: compiled apart (their mere presence cost the common kernel 2 %)
template <int G, bool REG>
__device__ __forceinline__ void g_frame(const FtlDevParams& P, GCtx& E, const Limits& L, double2& lead_cs, const int4* s_near, int& tick, const bool first, const int f_idx,
                                        unsigned char* s_rec, const int rec_stride, FrPend* s_pend, int* s_pcnt, bool& pend, int& pend_idx, float& new_ad) {
    const ftl_config& c = P.cfg;
    const int r = E.r;
    const bool act = E.valid && r < P.R;
    FTL_TIC_INIT;

    // the other robots as the follower's collision test and the bears' way-points see them: before anybody moves
    const float lpx0 = gb_f<G, 0>(E.rb.px), lpy0 = gb_f<G, 0>(E.rb.py);
    const float fpx0 = gb_f<G, 1>(E.rb.px), fpy0 = gb_f<G, 1>(E.rb.py);
    const double ldir0 = gb_d<G, 0>(E.rb.direction);
    const int orx = E.rb.rx, ory = E.rb.ry, orw = E.rb.rw, orh = E.rb.rh;

    // leader way-point switch (ENV:978-983)
    if (euclid_f64_lt((double)lpx0, (double)lpy0, E.cur_tx, E.cur_ty, c.leader_pos_epsilon)) {
        E.cur_target_id += 1;
        if (E.cur_target_id >= E.route_len) E.leader_finished = 1;
        else {
            const double* rt = P.scen.route + ((size_t)E.scen * c.route_cap + E.cur_target_id) * 2;
            E.cur_tx = rt[0]; E.cur_ty = rt[1];
        }
    }
    FTL_TIC(13);
    // bears: way-point choice (ENV:722-758, 819-837)
    double tx = E.cur_tx, ty = E.cur_ty;
    const bool is_bear = act && r >= 2;
#ifndef FTL_BEAR_SINCOS
    const double lc = gb_d<G, 0>(lead_cs.x), ls = gb_d<G, 0>(lead_cs.y);      // (cross-lane reads stay outside the bears' branch: its lane 0 is inactive there)
#endif
    if (is_bear) {
        const int b = r - 2;
        bool near = euclid_f64_lt((double)E.rb.px, (double)E.rb.py, E.rb.tgt_x, E.rb.tgt_y, c.leader_pos_epsilon);
        double off, lvl;
        bool drawn = false;
        if (c.move_bear_v4 && (b & 1)) {
            if (near) E.rb.dyn_index += 1;
            if (E.rb.dyn_index > 3) E.rb.dyn_index = 0;
            drawn = b >= 4;       // ENV:750-754: bears 5, 7, .. take one of four points drawn anew every frame (ftl_rand_range, include/ftl.h)
            const int order = (b == 1) ? 0x2134 /*p4,p3,p1,p2*/ : 0x4213 /*p3,p1,p2,p4*/;
            int p = (order >> (4 * E.rb.dyn_index)) & 0xf;
            lvl = (p <= 2) ? 150.0 : 250.0;
            off = (p == 1) ? 140.0 : (p == 2) ? -140.0 : (p == 3) ? -160.0 : 160.0;
        } else {
            if (near) { E.rb.dyn_index += 1; if (E.rb.dyn_index > 1) E.rb.dyn_index = 0; }
            lvl = 100.0 * (b + 1);
            off = (E.rb.dyn_index == 0) ? -130.0 : 130.0;
        }
        double s, co;
#ifdef FTL_BEAR_SINCOS
        sincos_bounded((ldir0 + off) * kDeg2Rad, s, co);
#else
        {   // rotateVector([lvl, 0], leader.direction + off) (ENV:730-737, 829-832): cos / sin of the sum by angle addition from the
            // leader's own cos / sin -- its last move left them behind (lead_cs) -- and the offset's (130, 140 or 160 degrees, glibc
            // values): 2-3 ulp from cos(radians(direction + off)), the class of the device's own sincos, for six
            // multiplications instead of a seventy-operation dependent chain in front of the bears' steering
            const double a = fabs(off);
            const double ca = a == 130.0 ? -0.6427876096865394 : a == 140.0 ? -0.7660444431189779 : -0.9396926207859083;
            const double sa0 = a == 130.0 ? 0.766044443118978 : a == 140.0 ? 0.6427876096865395 : 0.3420201433256689;
            const double sa = off < 0.0 ? -sa0 : sa0;
            co = lc * ca - ls * sa; s = ls * ca + lc * sa;
        }
#endif
        tx = (double)lpx0 + co * lvl; ty = (double)lpy0 + s * lvl;
        if (FTL_MAX_BEARS > 4 && drawn) {
            const int lo = (int)c.max_distance, k = 2 * E.rb.dyn_index;
            auto draw = [&](int kk, int stop) {
                const int nn = (stop - lo + 9) / 10;
                const double u = d_uniform01(c.rng_seed, (unsigned long long)(c.env_id_base + E.env), (unsigned long long)E.resets,
                                             (unsigned long long)E.step_count | (1ULL << 41) | ((unsigned long long)b << 44) | ((unsigned long long)kk << 48));
                const int v = (int)(u * (double)nn);
                return (double)(lo + 10 * (v < nn ? v : nn - 1));
            };
            tx = draw(k, c.width - lo); ty = draw(k + 1, c.height - lo);
        }
        E.rb.tgt_x = tx; E.rb.tgt_y = ty;
    }
    // leader speed / acceleration regimes (ENV:1048-1058, 1143-1174); evaluated only while the leader is under way
    double lspeed = c.leader.max_speed + 0;
    if (REG && !E.leader_finished && (c.n_speed_regime >= 0 || c.n_acc_regime >= 0)) {
        double speed = c.leader.max_speed, acceleration = 0;
        if (c.n_speed_regime >= 0) {
            int sel = -1;
            for (int i = 0; i < c.n_speed_regime; i++) if (c.speed_key[i] <= E.step_count) sel = i;   // dict order, last match wins
            if (sel >= 0) {
                if (c.speed_is_range[sel]) {
                    double u = d_uniform01(c.rng_seed, (unsigned long long)(c.env_id_base + E.env), (unsigned long long)E.resets, (unsigned long long)E.step_count);
                    E.cur_mult = c.speed_lo[sel] + (c.speed_hi[sel] - c.speed_lo[sel]) * u;
                } else E.cur_mult = c.speed_lo[sel];
            }
            speed = c.leader.max_speed * E.cur_mult;
        }
        if (c.n_acc_regime >= 0) {
            for (int i = 0; i < c.n_acc_regime; i++)
                if (!((E.acc_consumed >> i) & 1) && c.acc_key[i] <= E.step_count) { E.cur_acc = c.acc_val[i]; E.cum_speed = E.cur_acc; E.acc_consumed |= 1 << i; }
            E.cum_speed += E.cur_acc;
            acceleration = (E.cum_speed * c.leader.max_speed) / E.fps;
        }
        lspeed = speed + acceleration;
    }
    bool steers = (act && r == 0 && !E.leader_finished) || is_bear;
    if (steers) steer_to_point(E.rb, L, tx, ty, r == 0, lspeed);
    if (E.leader_finished) {                                   // ENV:1062-1065
        if (r == 0) { command_forward(E.rb, L, 0); command_turn(E.rb, L, 0, 0); }
    }
    bool moves = act && !(r == 0 && E.leader_finished);
    FTL_TIC(14);
    robot_move(E.rb, L, moves, lead_cs.y, lead_cs.x);
    FTL_TIC(15);

    const float fpx = gb_f<G, 1>(E.rb.px), fpy = gb_f<G, 1>(E.rb.py);
    const int frx = gb_i<G, 1>(E.rb.rx), fry = gb_i<G, 1>(E.rb.ry), frw = gb_i<G, 1>(E.rb.rw), frh = gb_i<G, 1>(E.rb.rh);
    const float lpx = gb_f<G, 0>(E.rb.px), lpy = gb_f<G, 0>(E.rb.py);
    const int lrx = gb_i<G, 0>(E.rb.rx), lry = gb_i<G, 0>(E.rb.ry), lrw = gb_i<G, 0>(E.rb.rw), lrh = gb_i<G, 0>(E.rb.rh);
    // follower + leader collisions against the near statics in one pass (ENV:960-964, 1068-1072, 1176-1194)
    bool fhit = false, lhit = false;
    for (int s = r; s < E.near_cnt; s += G) {
        int4 q = s_near[s];
        fhit |= rects_collide(frx, fry, frw, frh, q.x, q.y, q.z, q.w);
        lhit |= rects_collide(lrx, lry, lrw, lrh, q.x, q.y, q.z, q.w);
    }
    if (act && r != 1) fhit |= rects_collide(frx, fry, frw, frh, orx, ory, orw, orh);     // leader / bears where they were
    if (act && r == 1) lhit |= rects_collide(lrx, lry, lrw, lrh, E.rb.rx, E.rb.ry, E.rb.rw, E.rb.rh);   // follower where it is now
    fhit = group_any<G>(fhit); lhit = group_any<G>(lhit);
    bool f_coll = false;
    if (!c.ignore_follower_collisions) {
        bool out = (double)fpx > (double)c.width || (double)fpy > (double)c.height || fpx < 0.0f || fpy < 0.0f;
        f_coll = fhit || out;
    }
    FTL_TIC(0);
    // green zone (ENV:968-969): recomputed when a point was appended
    const float2* tr = reinterpret_cast<const float2*>(P.traj + (size_t)E.env * c.traj_cap * 2);
#ifndef FTL_ABLATE_GREEN
    bool g_upd = false, g_walk = false; int Gn = 0; double W = 0.0; int tiny = 0;
    if (E.green_len != E.traj_len) {
        g_upd = true;
        // One point was appended since the window was last derived.  The window (ENV:1828-1843: newest segments whose
        // sequential f64 length sum stays <= max_distance) is slid instead of re-walked: add the new segment, drop the
        // oldest ones while the running length exceeds max_distance.  The running length differs from the reference's
        // sums by rounding only (< 1e-10 over a whole episode; it is re-derived from scratch every 1024 points anyway), so every comparison that is not within
        // 1e-6 of the threshold is the reference's comparison; otherwise the full walk decides.
        const int nn = E.traj_len;
        const double maxd = c.max_distance;
        bool exact = (E.green_len != nn - 1) || (nn % 1024 == 0) || E.green_count < 1;
        Gn = E.green_count + 1; W = E.green_w; tiny = E.green_tiny;
        const double band = tiny ? 1e-6 : 0.0;            // exact arithmetic (see kTinySeg) needs no tolerance band
        if (!exact) {
            // everything the slide can need in ONE memory round trip: the two newest points and the four points around
            // the old end of the window (segments q1-q0 [just beyond], q2-q1 [oldest], q3-q2 [second oldest])
            const int o = nn - 2 - Gn;
            float2 p1 = tr[nn - 1], p0 = tr[nn - 2];
            float2 q0 = tr[max(o, 0)], q1 = tr[max(o + 1, 0)], q2 = tr[max(o + 2, 0)], q3 = tr[max(o + 3, 0)];
            asm volatile("" :: "v"(q2.x), "v"(q2.y), "v"(q3.x), "v"(q3.y));    // all six loads before the first wait: left alone, the compiler moves the
                                                                               // last two into the branches that use them, two more round trips
            const double d_beyond = euclid_f32(q1.x, q1.y, q0.x, q0.y), d_old1 = euclid_f32(q2.x, q2.y, q1.x, q1.y),
                         d_old2 = euclid_f32(q3.x, q3.y, q2.x, q2.y);
            const double d_new = euclid_f32(p1.x, p1.y, p0.x, p0.y);
            if (d_new != 0.0 && d_new < kTinySeg) { tiny = 1; exact = true; }     // sums stop being exact: re-derive with the band
            W += d_new;
            double nxt = d_beyond;                      // length of the segment just beyond the window
            bool have_nxt = o >= 0;
            if (W > maxd) {                              // drop the oldest segment
                if (fabs(W - maxd) < band || Gn < 2) exact = true;
                W -= d_old1; Gn -= 1; nxt = d_old1; have_nxt = true;
                if (W > maxd) {                          // and the second oldest
                    if (fabs(W - maxd) < band || Gn < 2) exact = true;
                    W -= d_old2; Gn -= 1; nxt = d_old2;
                    if (W > maxd) exact = true;          // more than two: let the full walk do it
                }
            }
            if (fabs(W - maxd) < band) exact = true;
            if (have_nxt && !(W + nxt > maxd + band)) exact = true;     // the segment beyond the window must stay excluded
        }
#ifdef FTL_DEBUG_EXACT
        if (exact) { int why = (E.green_len != nn - 1) ? 1 : (E.green_count < 1 ? 2 : 3); E.error = (E.error & 0xff) | (why << 8) | ((((E.error >> 16) + 1) & 0xffff) << 16); }
#endif
#ifdef FTL_WAVE_TIMES
        if (exact) E.dbg_walk += 1;
#endif
        g_walk = exact;
    }
#ifndef FTL_ABLATE_EXACT
    if (REG && c.rand_fps_hi > 0) {     // random frame counts: some groups of the wavefront sit this frame out, the walk stays inside the group
        if (g_walk) Gn = g_green_walk<G>(P, E, W, tiny);
    } else {   // the full walks of this frame, one env at a time by the whole wavefront (after a reset; once per 1024 points; near a threshold)
        unsigned long long need = __ballot(g_walk && E.valid && r == 0);
        while (need) {
            const int L = __ffsll((long long)need) - 1; need &= need - 1;
            const int envL = __builtin_amdgcn_readlane(E.env, L), nL = __builtin_amdgcn_readlane(E.traj_len, L);
            const WalkRes wr = w_green_walk(reinterpret_cast<const float2*>(P.traj + (size_t)envL * c.traj_cap * 2), nL, c.max_distance);
            if (((int)threadIdx.x & ~(G - 1)) == (L & ~(G - 1))) { Gn = wr.Gc; W = wr.wlen; tiny = wr.tiny; }
        }
    }
#endif
    if (g_upd) { E.green_count = Gn; E.green_w = W; E.green_len = E.traj_len; E.green_tiny = tiny; }
#endif
    FTL_TIC(1);
    const int Gc = E.green_count, n = E.traj_len;
    // _check_agent_position (ENV:1906-1937).  Its two arg-min searches only feed threshold tests: closest green point
    // within epsilon -> on trace + in box; within max_dev -> in box; otherwise closest point of the WHOLE trajectory
    // within epsilon -> on trace.  Most frames settle them from the env's search caches without touching memory (below); the
    // frames that cannot leave a PENDING item behind and go on: g_position_search() runs for the items of the whole wavefront, sixteen
    // at a time, after the frame that needs their caches refreshed (the first of a step) and after the last one.
    int rec = f_coll ? FR_COLLISION : 0;                      // this frame's record for the tail (FR_* bits)
    FTL_PROF(0, E.valid && r == 0, 1);
    // Search caches (exact: they only decide which points have to be looked at).  clr_g / clr_a are lower bounds on the
    // follower's distance to every green point / every trajectory point: a search leaves behind the smallest distance it
    // saw or proved (block boxes it skipped), the follower's own displacement is subtracted every frame and an appended
    // point enters with its distance.  (hx,hy) are the coordinates of point `hint`; one such point clearly inside a
    // threshold settles "the arg-min is inside it" without touching memory.
    {
        float mx = fpx - fpx0, my = fpy - fpy0;
        float disp = sqrtf(mx * mx + my * my) * 1.000001f + 1e-5f;
        E.clr_g -= disp; E.clr_a -= disp;
    }
    pend = false;
    if (Gc > 2) {
        FTL_PROF(1, E.valid && r == 0, 1);
        // Refresh ahead of need.  A search is paid by the whole wavefront, whichever env asks for it, so the envs whose
        // caches would run out somewhere inside this step all search after the FIRST frame, instead of in different ones.
        bool quiet;
        int fast = g_cache_class(c, E, fpx, fpy, E.fps, first, quiet);
        if (first && !quiet) fast = 0;
        FTL_PROF(2, E.valid && r == 0 && fast != 0, 1);
        FTL_PROF(13, E.valid && r == 0 && first && fast == 0, 1);
        FTL_PROF(14, E.valid && r == 0 && !first && fast == 0, 1);
        if (fast == 1) rec |= FR_ON_TRACE | FR_IN_BOX;
        else if (fast == 2) rec |= FR_IN_BOX;
        else if (fast == 4) rec |= FR_ON_TRACE;
        else if (fast == 0) {
            E.n_search += 1;
            pend = E.valid;
            rec |= FR_PENDING;
#ifdef FTL_WAVE_TIMES
            if (!first) E.dbg_full += 1;          // (diagnostic: items left for the end of the step)
#endif
            if (E.valid && r == 0) {                             // the item: everything the search needs of this frame
                const int k = atomicAdd(s_pcnt, 1);
                pend_idx = k;
                FrPend it;
                it.fpx = fpx; it.fpy = fpy; it.n = (unsigned short)n; it.gc = (unsigned short)Gc; it.hint = (unsigned short)E.hint;
                it.key = (unsigned short)((f_idx << 4) | E.slot);
                s_pend[k] = it;
            }
        }
    }
    FTL_TIC(2);
    if (euclid_f32_le(lpx0, lpy0, fpx, fpy, c.min_distance)) rec |= FR_TOO_CLOSE;
    {   // leader collision (ENV:1068-1072)
        bool out = (double)lpx > (double)c.width || (double)lpy > (double)c.height || lpx < 0.0f || lpy < 0.0f;
        if (lhit || out) rec |= FR_LEADER_HIT;
    }
    // ENV:1074-1075 with the deterministic tick: frame k (1-based since reset) sees get_ticks() == k
    const bool append = tick == 0;           // tick == (step_count + 1) % trajectory_saving_period, kept incrementally
    tick = (tick + 1 == c.trajectory_saving_period) ? 0 : tick + 1;
    new_ad = 3.0e38f;
    if (append) {
        if (E.traj_len < c.traj_cap) {
            if (E.valid && r == 0) {
                float2* tw = reinterpret_cast<float2*>(P.traj + (size_t)E.env * c.traj_cap * 2); tw[E.traj_len] = make_float2(lpx, lpy);
                float4* bb = reinterpret_cast<float4*>(P.traj_bb) + (size_t)E.env * (c.traj_cap / FTL_TRAJ_BLOCK) + E.traj_len / FTL_TRAJ_BLOCK;
                float4* s_box = reinterpret_cast<float4*>(s_pcnt + 4) + E.slot;       // the block's box so far (g_build_near)
                float4 box = (E.traj_len % FTL_TRAJ_BLOCK == 0) ? make_float4(lpx, lpy, lpx, lpy) : *s_box;
                box.x = fminf(box.x, lpx); box.y = fminf(box.y, lpy); box.z = fmaxf(box.z, lpx); box.w = fmaxf(box.w, lpy);
                *s_box = box; *bb = box;
            }
            {   // the new point enters the distance bounds
                float ax = lpx - fpx, ay = lpy - fpy;
                float ad = sqrtf(ax * ax + ay * ay) * 0.999999f - 1e-3f;
                E.clr_g = fminf(E.clr_g, ad); E.clr_a = fminf(E.clr_a, ad);
                new_ad = ad;                 // (a search of this frame that refreshes the bounds afterwards has not seen the point)
            }
            E.traj_len += 1;
        } else E.error |= FTL_ERR_TRAJ_OVERFLOW;
    }
    if (c.has_max_distance_coef) {                          // ENV:1099-1107, the distance part (the tail applies the warm-start gate)
        float dx = fpx - lpx, dy = fpy - lpy;
        float nrm = sqrtf(dx * dx + dy * dy);
        if (nrm > (float)(c.max_distance * c.max_distance_coef)) rec |= FR_TOO_FAR;
    }
    if (E.leader_finished) rec |= FR_LEADER_FINISHED;
    E.step_count += 1;
    if (E.valid && r == 0) s_rec[E.slot * rec_stride + f_idx] = (unsigned char)rec;
    FTL_TIC(3);
}

// The searches of _check_agent_position for one pending (env, frame) item, by one group of G lanes (any group: everything comes from
// the item).  Returns FR_IN_BOX | FR_ON_TRACE; hint / hx / hy / clr_g / clr_a are the env's search caches as this search leaves them.
template <int G>
__device__ __forceinline__ int g_position_search(const ftl_config& c, const float2* tr, const float4* bb, const int r, const float fpx, const float fpy,
                                                 const int n, const int Gc, const int fps, int& hint, float& hx, float& hy, float& clr_g, float& clr_a) {
    const double eps = c.leader_pos_epsilon, mdev = c.max_dev;
    const double far = fmax(mdev, eps);
    const int g_lo = n - 1 - Gc;                           // green points are indices g_lo .. n-2
    const float eps2_lo = (float)(eps * eps * (1.0 - 1e-5));
    int bits = 0;
    // hint window: 4*G points around the point that was closest last time, one memory round trip
    float wbest = __int_as_float(0x7f800000); int widx = 0x7fffffff;      // over the whole window
    float gbest = __int_as_float(0x7f800000); int gidx = 0x7fffffff;      // over its green members (ties -> higher index)
    float2 wp = make_float2(0.0f, 0.0f), gp = wp;                         // their coordinates
    // the green member FARTHEST AHEAD (largest index: the follower is chasing the leader) that is comfortably inside
    // epsilon: as the next cached point it stays valid about twice as long as the closest one
    int aidx = -1; float2 ap = wp;
    const float reach_a = (float)fps * (float)fmax(fabs(c.follower.max_speed), fabs(c.follower.min_speed)) * 1.001f + 1.0f;
    const float ahead2 = fmaxf((float)eps * 0.99999f - reach_a, 0.0f) * fmaxf((float)eps * 0.99999f - reach_a, 0.0f);
    {
        int w0 = hint - G; w0 = w0 < 0 ? 0 : w0;                         // a quarter of the window behind the old point, the rest ahead
        float2 wq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) wq[k] = tr[min(w0 + k * G + r, n - 1)];        // unguarded loads: issued back to back
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = w0 + k * G + r;
            if (i < n) {
                float2 p = wq[k];
                float dx = p.x - fpx, dy = p.y - fpy;
                float d2 = dx * dx + dy * dy;
                if (d2 < wbest) { wbest = d2; widx = i; wp = p; }
                if (i >= g_lo && i <= n - 2 && d2 <= gbest) { gbest = d2; gidx = i; gp = p; }
                if (i >= g_lo + 4 && i <= n - 2 && d2 < ahead2 && i > aidx) { aidx = i; ap = p; }
            }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            float ov = gx(wbest, off, G); int oi = gx(widx, off, G);
            float ox = gx(wp.x, off, G), oy = gx(wp.y, off, G);
            bool take = (ov < wbest) || (ov == wbest && oi < widx);
            if (take) { wbest = ov; widx = oi; wp.x = ox; wp.y = oy; }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            float ov = gx(gbest, off, G); int oi = gx(gidx, off, G);
            float ox = gx(gp.x, off, G), oy = gx(gp.y, off, G);
            bool take = (ov < gbest) || (ov == gbest && oi != 0x7fffffff && (gidx == 0x7fffffff || oi > gidx));
            if (take) { gbest = ov; gidx = oi; gp.x = ox; gp.y = oy; }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            int oi = gx(aidx, off, G); float ox = gx(ap.x, off, G), oy = gx(ap.y, off, G);
            if (oi > aidx) { aidx = oi; ap.x = ox; ap.y = oy; }
        }
    }
    if (gbest < eps2_lo) {                                 // a green point is within epsilon
        bits = FR_ON_TRACE | FR_IN_BOX;
        if (aidx >= 0) { hint = aidx; hx = ap.x; hy = ap.y; } else { hint = gidx; hx = gp.x; hy = gp.y; }
    } else {
        float gb2; int gi; float2 q; float skip;
        g_range_argmin<G>(tr, bb, r, fpx, fpy, g_lo, n - 1, (float)(far * far * (1.0 + 1e-5)) + 1e-2f, true, gidx, gbest, gp, gb2, gi, q, skip);
        clr_g = sqrtf(fmaxf(fminf(gb2, skip), 0.0f)) * 0.999999f - 1e-3f;       // every green point is at least this far
        bool in_eps = false, in_dev = false;
        if (gi != 0x7fffffff) {
            in_eps = euclid_f32_le(fpx, fpy, q.x, q.y, eps);
            in_dev = !in_eps && euclid_f32_le(fpx, fpy, q.x, q.y, mdev);
        }
        if (in_eps) { bits = FR_ON_TRACE | FR_IN_BOX; hint = gi; hx = q.x; hy = q.y; }
        else if (in_dev) { bits = FR_IN_BOX; hint = gi; hx = q.x; hy = q.y; }
        else if (wbest < eps2_lo) { bits = FR_ON_TRACE; hint = widx; hx = wp.x; hy = wp.y; }   // some point is within epsilon
#ifdef FTL_ABLATE_WHOLE      // timing experiment only: what the whole-trajectory searches cost (results differ)
        else if (fps != -12345) {
#else
        else {                                             // closest point of the whole trajectory (ENV:1924-1930)
#endif
            float ab2; int ai; float2 q2; float skip2;
            g_range_argmin<G>(tr, bb, r, fpx, fpy, 0, n, (float)(eps * eps * (1.0 + 1e-5)) + 1e-2f, false, widx, wbest, wp, ab2, ai, q2, skip2);
            clr_a = sqrtf(fmaxf(fminf(ab2, skip2), 0.0f)) * 0.999999f - 1e-3f;  // every trajectory point is at least this far
            if (ai != 0x7fffffff) {
                if (euclid_f32_le(fpx, fpy, q2.x, q2.y, eps)) bits = FR_ON_TRACE;
                hint = ai; hx = q2.x; hy = q2.y;
            }
        }
    }
    return bits;
}

// The pending items of the wavefront.  Two modes.  own >= 0 / -1 (per group): the group searches for its OWN env's item of the frame
// that just ran (index `own`, -1 = none) and takes the refreshed search caches home (hint .. clr_a by reference; a bound the search did
// not touch stays negative).  Otherwise (`spread`): items [0, cnt) go to the groups sixteen at a time, whichever env they belong to --
// the searches of the later frames of a step, whose caches nobody waits for.  The result bits replace FR_PENDING in the frame records.
template <int G>
__device__ __forceinline__ void g_resolve(const FtlDevParams& P, const GCtx& E, const FrPend* s_pend, unsigned char* s_rec, const int rec_stride, const int* s_env,
                                          const int cnt, const bool spread, const int own, int& hint, float& hx, float& hy, float& clr_g, float& clr_a) {
    constexpr int EPW = FTL_WAVE / G;
    const ftl_config& c = P.cfg;
#ifdef FTL_ABLATE_SPREAD     // timing experiment only: what the deferred searches cost (results differ)
    if (spread) return;
#endif
    for (int base = 0; base < (spread ? cnt : 1); base += EPW) {      // wave-uniform trip count
        const int k = spread ? base + E.slot : own;
        if (k >= 0 && k < cnt) {                                     // group-uniform
            const FrPend it = s_pend[k];
            const int oslot = it.key & 15, frame = it.key >> 4, env = s_env[oslot];
            const float2* tr = reinterpret_cast<const float2*>(P.traj + (size_t)env * c.traj_cap * 2);
            const float4* bb = reinterpret_cast<const float4*>(P.traj_bb) + (size_t)env * (c.traj_cap / FTL_TRAJ_BLOCK);
            hint = it.hint; hx = 3.0e38f; hy = 3.0e38f; clr_g = -1.0f; clr_a = -1.0f;
            const int bits = g_position_search<G>(c, tr, bb, E.r, it.fpx, it.fpy, (int)it.n, (int)it.gc,
#ifdef FTL_ABLATE_WHOLE
                                                     spread ? -12345 : E.fps,
#else
                                                     spread ? c.frames_per_step : E.fps,
#endif
                                                     hint, hx, hy, clr_g, clr_a);
            if (E.r == 0) {
                unsigned char* rp = s_rec + oslot * rec_stride + frame;
                *rp = (unsigned char)((*rp & ~(FR_PENDING | FR_IN_BOX | FR_ON_TRACE)) | bits);
            }
        }
    }
}

// The constants of the tail, loaded once per wavefront and made opaque to the compiler: left as loads from the parameter block they are
// re-loaded where they are used (there are no registers to keep them in across the frame loop), and the reward code becomes a chain of
// ten `s_load; s_waitcnt` per frame.  The readfirstlane sits INSIDE the asm (with an "s" input this compiler hands the asm a VGPR
// whenever it has the value in one) together with the wait states the hazard recogniser, which does not look inside an asm, would have
// inserted: one between a VALU write of a VGPR and a v_readfirstlane of it (without it the values came out stale on gfx950), five
// between a VALU write of an SGPR and a VMEM instruction that reads it.
__device__ __forceinline__ int pin_dword(int v) { int o; asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(o) : "v"(v)); return o; }
__device__ __forceinline__ int pin_s(int v) { return pin_dword(v); }
__device__ __forceinline__ double pin_s(double v) { const int hi = pin_dword(__double2hiint(v)), lo = pin_dword(__double2loint(v)); return __hiloint2double(hi, lo); }
struct TailK {
    double r_move, r_too_close, r_in_box, r_in_dev, r_on_track, r_not_on_track, r_crash, low_reward;
    int warm_start, max_steps, flags;
};
enum { TK_LOW_REWARD = 1, TK_AGGREGATE = 2 };
__device__ __forceinline__ TailK tail_consts(const ftl_config& c) {
    TailK K;
    K.r_move = pin_s(c.leader_movement_reward); K.r_too_close = pin_s(c.too_close_penalty); K.r_in_box = pin_s(c.reward_in_box);
    K.r_in_dev = pin_s(c.reward_in_dev); K.r_on_track = pin_s(c.reward_on_track); K.r_not_on_track = pin_s(c.not_on_track_penalty);
    K.r_crash = pin_s(c.crash_penalty); K.low_reward = pin_s(c.low_reward);
    K.warm_start = pin_s(c.warm_start); K.max_steps = pin_s(c.max_steps);
    K.flags = pin_s((c.has_low_reward ? TK_LOW_REWARD : 0) | (c.aggregate_reward ? TK_AGGREGATE : 0));
    return K;
}

// The tail of frame_step for one recorded frame (ENV:1077-1141): finish timer, early stopping, reward, step count, termination codes.
// Nothing in a frame reads what it computes (the reference keeps simulating after `done`), so it runs over the records after the loop.
__device__ __forceinline__ void g_tail(const TailK& K, GCtx& E, const int rec, int& sc, double& reward, int& i0, int& i1, int& i2) {
    E.is_in_box = (rec & FR_IN_BOX) ? 1 : 0; E.is_on_trace = (rec & FR_ON_TRACE) ? 1 : 0; E.too_close = (rec & FR_TOO_CLOSE) ? 1 : 0;
    i0 = FTL_MISSION_IN_PROGRESS; i1 = FTL_AGENT_MOVING; i2 = FTL_LEADER_MOVING;
    if (rec & FR_LEADER_FINISHED) i2 = FTL_LEADER_FINISHED;          // ENV:1062-1065
    if (rec & FR_COLLISION) { E.crash = 1; E.done = 1; i0 = FTL_MISSION_FAIL; i1 = FTL_AGENT_CRASH; }      // ENV:960-964
    if (rec & FR_LEADER_HIT) { E.done = 1; i0 = FTL_MISSION_FAIL; i2 = FTL_LEADER_CRASH; }                 // ENV:1068-1072
    if ((rec & FR_LEADER_FINISHED) && E.is_in_box) {       // ENV:1077-1087
        if (E.finish_timer < 0) E.finish_timer = 0;
        else {
            E.finish_timer += 1;
            if (E.finish_timer > E.fps * 20) { i0 = FTL_MISSION_SUCCESS; i2 = FTL_LEADER_FINISHED; i1 = FTL_AGENT_FINISHED; E.done = 1; }
        }
    }
    const bool warm = sc > K.warm_start;
    if (warm) {                                             // ENV:1088-1107
        if ((K.flags & TK_LOW_REWARD) && E.acc_penalty < K.low_reward) { i0 = FTL_MISSION_FAIL; i2 = FTL_LEADER_MOVING; i1 = FTL_AGENT_LOW_REWARD; E.crash = 1; E.done = 1; }
        if (rec & FR_TOO_FAR) { i0 = FTL_MISSION_FAIL; i2 = FTL_LEADER_MOVING; i1 = FTL_AGENT_TOO_FAR; E.crash = 1; E.done = 1; }
    }
    double res = 0;                                         // _reward_computation (ENV:1869-1904)
    res += K.r_move;
    if (E.too_close) res += K.r_too_close;
    else {
        if (E.is_in_box && E.is_on_trace) res += K.r_in_box;
        else if (E.is_in_box) res += K.r_in_dev;
        else if (E.is_on_trace) res += K.r_on_track;
        else if (warm) res += K.r_not_on_track;
    }
    if (E.crash) res += K.r_crash;
    if (res < 0) E.acc_penalty += res; else E.acc_penalty = 0;
    E.overall_reward += res;
    sc += 1;
    if (sc > K.max_steps) { i0 = FTL_MISSION_FINISHED_BY_TIME; i2 = FTL_LEADER_MOVING; i1 = FTL_AGENT_MOVING; E.done = 1; }
    reward = (K.flags & TK_AGGREGATE) ? E.overall_reward : res;
}

// ---- LeaderPositionsTracker_v2.scan (sensors.py:243-327), one env per group; the sequential parts (numpy pairwise
// sum, border pairs) are evaluated redundantly by every lane of the group, memory is written by lane r == 0 ------------
__device__ __forceinline__ double g_hist_dist(const FtlDevParams& P, int env, int i, bool f64) {     // |hist[i] - hist[i+1]| in the array's dtype
    const double* p = hist_slot(P, env, i); const double* q = hist_slot(P, env, i + 1);
    if (f64) { double dx = p[0] - q[0], dy = p[1] - q[1]; return sqrt(dx * dx + dy * dy); }
    float dx = (float)p[0] - (float)q[0], dy = (float)p[1] - (float)q[1];
    return (double)sqrtf(dx * dx + dy * dy);
}
// numpy pairwise sum of the m-1 consecutive distances of hist[lo..hi) without staging them (a[i] = dist(lo+i, lo+i+1));
// T is the array dtype (float when no seeded float64 point is left).  n <= 128 * 4 (validated on the host).
// The eight interleaved accumulators of numpy's unrolled loop are spread over the G lanes of the env's group (8/G each)
// and combined in numpy's order ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) with width-G shuffles -- IEEE addition commutes, so
// every lane ends with the same bits as the sequential form.  Must be called by all G lanes of the group.
template <typename T, int G>
__device__ __forceinline__ T g_pw128(const FtlDevParams& P, int env, int r, int first, int n, bool f64) {
    if (n < 8) { T s = (T)0; for (int i = 0; i < n; i++) s += (T)g_hist_dist(P, env, first + i, f64); return s; }
    constexpr int APL = 8 / G;                     // accumulators per lane
    T acc[APL];
#pragma unroll
    for (int a = 0; a < APL; a++) acc[a] = (T)g_hist_dist(P, env, first + r * APL + a, f64);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int a = 0; a < APL; a++) acc[a] += (T)g_hist_dist(P, env, first + i + r * APL + a, f64);
    }
    T res = acc[0];
    if constexpr (APL == 2) res = acc[0] + acc[1];                               // r_{2k} + r_{2k+1}
    else res = res + gx(res, 1, G);
    res = res + gx(res, APL == 2 ? 1 : 2, G);                             // (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)
    res = res + gx(res, APL == 2 ? 2 : 4, G);
    for (; i < n; i++) res += (T)g_hist_dist(P, env, first + i, f64);
    return res;
}
// numpy's pairwise_sum recursion above its 128-element leaf, for n <= 512 (two levels; validated on the host), written as
// loops around ONE inlined copy of the leaf (the recursive template form expanded it 7 times per dtype -- instruction cache)
template <typename T, int G>
__device__ __forceinline__ T g_pw(const FtlDevParams& P, int env, int r, int first, int n, bool f64) {
    int hn[2] = {n, 0};
    if (n > 128) { int n2 = n / 2; n2 -= n2 % 8; hn[0] = n2; hn[1] = n - n2; }
    T total = (T)0;
    int hf = first;
#pragma nounroll
    for (int h = 0; h < 2; h++) {
        if (hn[h] == 0) break;
        int qn[2] = {hn[h], 0};
        if (hn[h] > 128) { int n2 = hn[h] / 2; n2 -= n2 % 8; qn[0] = n2; qn[1] = hn[h] - n2; }
        T hs = (T)0;
        int qf = hf;
#pragma nounroll
        for (int q = 0; q < 2; q++) {
            if (qn[q] == 0) break;
            const T v = g_pw128<T, G>(P, env, r, qf, qn[q], f64);
            hs = q == 0 ? v : hs + v;
            qf += qn[q];
        }
        total = h == 0 ? hs : total + hs;
        hf += hn[h];
    }
    return total;
}
template <int G>
__device__ __forceinline__ double g_path_length(const FtlDevParams& P, const GCtx& E, int lo, int hi) {
    int m = hi - lo;
    if (m < 2) return 0.0;
    if (lo < E.seed_end) return g_pw<double, G>(P, E.env, E.r, lo, m - 1, true);
    return (double)g_pw<float, G>(P, E.env, E.r, lo, m - 1, false);
}
__device__ __forceinline__ void g_border_pair(const FtlDevParams& P, const GCtx& E, int i1, int i0, int ia, int at, bool write) {
    const double* p1 = hist_slot(P, E.env, i1); const double* p0 = hist_slot(P, E.env, i0); const double* a = hist_slot(P, E.env, ia);
    double vx, vy;
    if (i1 < E.seed_end || i0 < E.seed_end) {
        vx = p1[0] - p0[0]; vy = p1[1] - p0[1];
        double nrm = sqrt(__builtin_fma(vy, vy, vx * vx));
        double sc = P.cfg.corridor_width / nrm;
        vx *= sc; vy *= sc;
    } else {
        float fx = (float)p1[0] - (float)p0[0], fy = (float)p1[1] - (float)p0[1];
        float nrm = sqrtf(fx * fx + fy * fy);
        float sc = (float)P.cfg.corridor_width / nrm;
        fx *= sc; fy *= sc; vx = (double)fx; vy = (double)fy;
    }
    const double c90 = 6.123233995736766e-17, s90 = 1.0, cm90 = 6.123233995736766e-17, sm90 = -1.0;
    double r0 = (c90 * vx + (-s90) * vy) + a[0], r1 = (s90 * vx + c90 * vy) + a[1];
    double l0 = (cm90 * vx + (-sm90) * vy) + a[0], l1 = (sm90 * vx + cm90 * vy) + a[1];
    if (write) {
        double* q = corr_slot(P, E.env, at); q[0] = r0; q[1] = r1; q[2] = l0; q[3] = l1;
        *corr32_slot(P, E.env, at) = make_float4((float)r0, (float)r1, (float)l0, (float)l1);
    }
}

// Both scans of a step for every env of the wave.  Memory written by one lane is read by the other lanes of its group
// only after a workgroup barrier; barriers sit at wave-uniform points (the phases below are predicated, not branched).
template <int G>
__device__ __forceinline__ void g_sensors(const FtlDevParams& P, GCtx& E) {
    const ftl_config& c = P.cfg;
    if (c.has_tracker != 2) { E.scan_ok = 0; return; }       // no tracker, or the v1 tracker (ftl_tracker1_kernel does its scan and the snapshot)
    int groups = 0, strict = 0;       // dict-order groups with a ray sensor / with one that raises on a corridor of <= 1 points
    for (int k = 0; k < c.n_lasers; k++) {
        groups |= 1 << (c.lasers[k].after_tracker ? 1 : 0);
        if (!c.lasers[k].lenient) strict |= 1 << (c.lasers[k].after_tracker ? 1 : 0);
    }
    int ok = 0, w0lo = 0, w0hi = 0;
    const float lpx = gb_f<G, 0>(E.rb.px), lpy = gb_f<G, 0>(E.rb.py);
    const float fpx = gb_f<G, 1>(E.rb.px), fpy = gb_f<G, 1>(E.rb.py);
    const double fdir = gb_d<G, 1>(E.rb.direction);
    const bool w = E.valid && E.r == 0;
#pragma nounroll
    for (int g = 0; g < 2; g++) {
        // ---- phase A: decide, append the new history point(s) ---------------------------------------------------------
        bool save = E.valid && (E.trk_counter % c.tracker_saving_period == 0);
        bool counted = E.valid;                      // sensors.py:247-251 returns WITHOUT incrementing the counter
        bool first = false;
        if (save) {
            int len = E.corr_hi - E.corr_lo;
            if (len > 0) {
                const double* last = hist_slot(P, E.env, E.corr_hi - 1);
                if (last[0] == (double)lpx && last[1] == (double)lpy) { save = false; counted = false; }
            }
            if (save) {
                first = (len == 0 && E.trk_counter == 0);
                if (first) {
                    const double lmax = c.leader.max_speed;
                    int n;
                    if (c.tracker_start_behind) {                     // sensors.py:257-272 (float64 seed points)
                        double s, co;
                        sincos_bounded(angle_correction(fdir + 180.0) * kDeg2Rad, s, co);
                        double sx = 50 * co + (double)fpx, sy = 50 * s + (double)fpy;
                        double dist = euclid_f64(sx, sy, (double)lpx, (double)lpy);
                        n = (int)(dist / ((double)(c.tracker_saving_period * 5) * lmax));
                        if (n < 2 || n > c.corr_cap) { E.error |= (n < 2) ? FTL_ERR_TRACKER_SEED : FTL_ERR_CORR_OVERFLOW; save = false; }
                        else {
                            double stepx = ((double)lpx - sx) / (n - 1), stepy = ((double)lpy - sy) / (n - 1);
                            for (int i = E.r; i < n; i += G) {
                                double x = (stepx == 0) ? ((double)i / (n - 1)) * ((double)lpx - sx) + sx : (double)i * stepx + sx;
                                double y = (stepy == 0) ? ((double)i / (n - 1)) * ((double)lpy - sy) + sy : (double)i * stepy + sy;
                                if (i == n - 1) { x = (double)lpx; y = (double)lpy; }
                                double* h = hist_slot(P, E.env, i); h[0] = x; h[1] = y;
                            }
                            E.seed_end = n; E.corr_lo = 0; E.corr_hi = n;
                        }
                    } else {                                          // sensors.py:275-284 (np.linspace(f32,f32) is float32)
                        double dist = euclid_f32(fpx, fpy, lpx, lpy);
                        n = (int)(dist / ((double)(c.tracker_saving_period * 5) * lmax));
                        if (n < 2 || n > c.corr_cap) { E.error |= (n < 2) ? FTL_ERR_TRACKER_SEED : FTL_ERR_CORR_OVERFLOW; save = false; }
                        else {
                            float stepx = (lpx - fpx) / (float)(n - 1), stepy = (lpy - fpy) / (float)(n - 1);
                            for (int i = E.r; i < n; i += G) {
                                float x = (stepx == 0) ? ((float)i / (float)(n - 1)) * (lpx - fpx) + fpx : (float)i * stepx + fpx;
                                float y = (stepy == 0) ? ((float)i / (float)(n - 1)) * (lpy - fpy) + fpy : (float)i * stepy + fpy;
                                if (i == n - 1) { x = lpx; y = lpy; }
                                double* h = hist_slot(P, E.env, i); h[0] = (double)x; h[1] = (double)y;
                            }
                            E.seed_end = 0; E.corr_lo = 0; E.corr_hi = n;
                        }
                    }
                } else {                                              // sensors.py:286
                    int oldest = E.corr_lo;               // the ring must still hold every point a stored snapshot refers to
                    const int* sw = rec_field(P.snap_win, P, E.env);
                    int nsnap = E.snap_count < P.hmax ? E.snap_count : P.hmax;
                    for (int j = 0; j < nsnap; j++) { int l0 = sw[4 * j], l1 = sw[4 * j + 2]; oldest = min(oldest, min(l0, l1)); }
                    if (E.corr_hi + 1 - oldest > c.corr_cap) { E.error |= FTL_ERR_CORR_OVERFLOW; save = false; }
                    else {
                        if (w) { double* h = hist_slot(P, E.env, E.corr_hi); h[0] = (double)lpx; h[1] = (double)lpy; }
                        E.corr_hi += 1;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase B: trim to corridor_length, append the border pair(s) ------------------------------------------------
        if (save) {
            double path = g_path_length<G>(P, E, E.corr_lo, E.corr_hi);          // sensors.py:288-297
            while (path > c.corridor_length) {
                if (first) E.error |= FTL_ERR_TRACKER_SEED;                   // reference: popleft on the still-empty corridor deque
                E.corr_lo += 1;
                path = g_path_length<G>(P, E, E.corr_lo, E.corr_hi);
            }
            int m = E.corr_hi - E.corr_lo;
            if (m > 1) {                                                      // sensors.py:299-317
                if (first)
                    for (int i = m - 1 - E.r; i > 0; i -= G)                  // the m-1 initial pairs are independent: strided over the group
                        g_border_pair(P, E, E.corr_lo + i, E.corr_lo + i - 1, E.corr_lo + m - i - 1, E.corr_lo + m - 1 - i, true);
                g_border_pair(P, E, E.corr_hi - 1, E.corr_hi - 2, E.corr_hi - 2, E.corr_hi - 1, w);
            }
        }
        if (counted) E.trk_counter += 1;
        __syncthreads();
        if ((groups >> g) & 1) {
            if (E.corr_hi - E.corr_lo > 1) ok |= 1 << g;
            else if ((strict >> g) & 1) E.error |= FTL_ERR_EMPTY_CORRIDOR;      // SEN:893/962 (LeaderCorridor_lasers_v2 does not raise)
        }
        if (g == 0) { w0lo = E.corr_lo; w0hi = E.corr_hi; }
    }
    if (c.n_aux > 0 && w) { int* ei = rec_field(P.env_int, P, E.env); ei[FTL_EI_HW0_LO] = w0lo; ei[FTL_EI_HW0_HI] = w0hi; }
    E.scan_ok = ok;
    if (ok && E.valid) {             // one snapshot per step: dynamic rects + the corridor window each group saw
        int slot = E.snap_head;
        int4* sr = reinterpret_cast<int4*>(rec_field(P.snap_rects, P, E.env)) + slot * (P.R - 1);
        if (E.r < P.R && E.r != 1) sr[E.r == 0 ? 0 : E.r - 1] = make_int4(E.rb.rx, E.rb.ry, E.rb.rw, E.rb.rh);
        if (E.r == 0) {
            int* sw = rec_field(P.snap_win, P, E.env) + slot * 4;
            bool g0 = ok & 1;
            sw[0] = g0 ? w0lo : E.corr_lo; sw[1] = g0 ? w0hi : E.corr_hi; sw[2] = E.corr_lo; sw[3] = E.corr_hi;
        }
        E.snap_count += 1;
        E.snap_head = (E.snap_head + 1 == P.hmax) ? 0 : E.snap_head + 1;
    }
}

template <int G>
__device__ __forceinline__ void g_write_obs(const FtlDevParams& P, const FtlCall& C, const GCtx& E) {     // ENV:1789-1810
    if (!E.valid) return;
    if (E.r < 2) {
        float* o = C.out.obs_num + (size_t)E.env * FTL_OBS_NUM + 5 * E.r;
        o[0] = E.rb.px; o[1] = E.rb.py; o[2] = (float)E.rb.speed; o[3] = (float)E.rb.direction; o[4] = (float)E.rb.rot_speed;
    }
    if (E.r == 0) {
        double tx = E.cur_tx, ty = E.cur_ty;
        if (E.route_len > 1) {
            const double* rt = P.scen.route + (size_t)E.scen * P.cfg.route_cap * 2;
            if (tx == rt[2 * (E.route_len - 1)] && ty == rt[2 * (E.route_len - 1) + 1]) { tx = rt[2 * (E.route_len - 2)]; ty = rt[2 * (E.route_len - 2) + 1]; }
        }
        C.out.target[2 * (size_t)E.env] = tx; C.out.target[2 * (size_t)E.env + 1] = ty;
    }
}


// ---- env regrouping: counting sort of the envs by P.keys (descending), rebuilt after every frame-kernel launch --------------
#define FTL_NKEYS 64
#define FTL_RG_BLOCK 1024
// pass 1: per block of 1024 envs, count the envs of every key, reserve the block's range inside each key's segment with one
// atomic per (block, key) (which block comes first inside a key is irrelevant) and give every env its rank in that range.
// tot = the key totals of this step (one of two buffers, the other one is cleared by pass 2 for the next step).
__global__ void __launch_bounds__(FTL_RG_BLOCK) ftl_regroup_count_kernel(const FtlDevParams* __restrict__ Pp, int* __restrict__ tot) {
    const FtlDevParams& P = *Pp;
    __shared__ int cnt[FTL_RG_BLOCK / FTL_WAVE][FTL_NKEYS];
    const int t = threadIdx.x, w = t / FTL_WAVE, lane = t % FTL_WAVE;
    const int env = blockIdx.x * FTL_RG_BLOCK + t;
    const int key = env < P.n_envs ? (FTL_NKEYS - 1 - (P.keys[env] & (FTL_NKEYS - 1))) : -1;       // descending: expensive envs first
    int my_rank = 0;
    for (int k = 0; k < FTL_NKEYS; k++) {
        const unsigned long long m = __ballot(key == k);
        if (lane == 0) cnt[w][k] = __popcll(m);
        if (key == k) my_rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    int base = 0;
    if (key >= 0) for (int ww = 0; ww < w; ww++) base += cnt[ww][key];
    if (env < P.n_envs) P.rank[env] = (uint16_t)(base + my_rank);
    if (t < FTL_NKEYS) {
        int n_k = 0;
        for (int ww = 0; ww < FTL_RG_BLOCK / FTL_WAVE; ww++) n_k += cnt[ww][t];
        P.bh[blockIdx.x * FTL_NKEYS + t] = n_k ? atomicAdd(&tot[t], n_k) : 0;      // start of this block's range inside key t
    }
}
// pass 2: slot of an env = (envs of smaller keys) + (start of its block's range inside its key) + its rank
__global__ void __launch_bounds__(FTL_RG_BLOCK) ftl_regroup_scatter_kernel(const FtlDevParams* __restrict__ Pp, const int* __restrict__ tot, int* __restrict__ tot_next) {
    const FtlDevParams& P = *Pp;
    __shared__ int off[FTL_NKEYS];
    const int t = threadIdx.x;
    if (t < FTL_NKEYS) {
        int smaller = 0;
        for (int k = 0; k < t; k++) smaller += tot[k];
        off[t] = smaller + P.bh[blockIdx.x * FTL_NKEYS + t];
        if (blockIdx.x == 0) tot_next[t] = 0;
    }
    __syncthreads();
    const int env = blockIdx.x * FTL_RG_BLOCK + t;
    if (env < P.n_envs) {
        const int key = FTL_NKEYS - 1 - (P.keys[env] & (FTL_NKEYS - 1));
        P.perm[off[key] + P.rank[env]] = env;
    }
}
__global__ void ftl_perm_identity_kernel(int32_t* perm, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) perm[i] = i; }

// ---- episode metrics (include/ftl.h ftl_episode_metrics): fixed-order sum of the per-env "ep_stats" records ---------------------
// pass 1: block b sums envs [b*FTL_MT_BLOCK, (b+1)*FTL_MT_BLOCK) -- every thread walks its envs in index order, then a fixed
// tree over the threads; pass 2: one block adds the partial vectors in block order.  No atomics: the result is bit-reproducible.
#define FTL_MT_THREADS 256
#define FTL_MT_BLOCK 4096
__global__ void __launch_bounds__(FTL_MT_THREADS) ftl_metrics_partial_kernel(const FtlDevParams* __restrict__ Pp, double* __restrict__ part, int* __restrict__ epart, int clear) {
    const FtlDevParams& P = *Pp;
    __shared__ double sm[FTL_MT_THREADS][FTL_N_METRICS + 1];
    __shared__ int se[FTL_MT_THREADS][2];
    const int t = threadIdx.x;
    double acc[FTL_N_METRICS];
#pragma unroll
    for (int k = 0; k < FTL_N_METRICS; k++) acc[k] = 0.0;
    int ecount = 0, ebits = 0;
    const int e0 = blockIdx.x * FTL_MT_BLOCK, e1 = min(e0 + FTL_MT_BLOCK, P.n_envs);
    for (int e = e0 + t; e < e1; e += FTL_MT_THREADS) {
        double* st = P.ep_stats + (size_t)e * FTL_N_METRICS;
#pragma unroll
        for (int k = 0; k < FTL_N_METRICS; k++) { acc[k] += st[k]; if (clear) st[k] = 0.0; }
        int* sticky = rec_field(P.env_int, P, e) + FTL_EI_ERROR_STICKY;
        const int b = *sticky;
        if (b) { ecount += 1; ebits |= b; if (clear) *sticky = 0; }
    }
#pragma unroll
    for (int k = 0; k < FTL_N_METRICS; k++) sm[t][k] = acc[k];
    se[t][0] = ecount; se[t][1] = ebits;
    __syncthreads();
    for (int off = FTL_MT_THREADS / 2; off >= 1; off >>= 1) {
        if (t < off) {
#pragma unroll
            for (int k = 0; k < FTL_N_METRICS; k++) sm[t][k] += sm[t + off][k];
            se[t][0] += se[t + off][0]; se[t][1] |= se[t + off][1];
        }
        __syncthreads();
    }
    if (t < FTL_N_METRICS) part[blockIdx.x * FTL_N_METRICS + t] = sm[0][t];
    if (t < 2) epart[blockIdx.x * 2 + t] = se[0][t];
}
__global__ void ftl_metrics_final_kernel(const double* __restrict__ part, const int* __restrict__ epart, int nb, double* __restrict__ out, int* __restrict__ eout) {
    const int t = threadIdx.x;
    if (t < FTL_N_METRICS) { double s = 0.0; for (int b = 0; b < nb; b++) s += part[b * FTL_N_METRICS + t]; out[t] = s; }
    if (eout && t == FTL_N_METRICS) { int n = 0, bits = 0; for (int b = 0; b < nb; b++) { n += epart[2 * b]; bits |= epart[2 * b + 1]; } eout[0] = n; eout[1] = bits; }
}

}  // namespace ftl

#ifndef FTL_FRAMESG_WPE
#define FTL_FRAMESG_WPE 2
#endif

template <int G, bool REG>
__global__ void __launch_bounds__(FTL_WAVE, FTL_FRAMESG_WPE) ftl_frames_group_kernel(const FtlDevParams* __restrict__ Pp, const FtlCall C) {
    extern __shared__ __align__(16) unsigned char lds[];
    using namespace ftl;
    const FtlDevParams& P = *Pp;
    constexpr int EPW = FTL_WAVE / G;
    GCtx E;
    E.slot = threadIdx.x / G; E.r = threadIdx.x % G;
    // Envs are not bound to wavefronts: slot -> env goes through the permutation that ftl_regroup_* rebuilt after the last
    // launch (envs with similar expected cost share a wavefront, expensive ones first); any permutation gives the same results.
    const int gslot = ((int)blockIdx.x * C.parts + C.part) * EPW + E.slot;     // this launch's share of the slot groups
    E.valid = gslot < P.n_envs;
    E.env = E.valid ? (P.perm ? P.perm[gslot] : gslot) : P.n_envs - 1;       // idle groups shadow the last env (loads only; every store is guarded)
    int4* s_near = reinterpret_cast<int4*>(lds);
    int* s_cnt = reinterpret_cast<int*>(lds + (size_t)EPW * P.cfg.n_static * 16);
    E.scan_ok = 0; E.near_cnt = 0; E.n_search = 0; E.err_acc = 0;
    const Limits L = lane_limits(P.cfg, E.r);
#ifdef FTL_WAVE_TIMES
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_wt[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    bool wt_reset = false; E.dbg_walk = 0; E.dbg_full = 0;
#endif
#ifdef FTL_PROFILE_PATHS
    if (threadIdx.x < 16) s_cyc[threadIdx.x] = 0;
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_wave_t[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
#endif
    FTL_TIC_INIT;
    if (C.mode == 1) {                                   // reset(): ENV:434-543
        if (C.mask && !C.mask[E.env]) E.valid = false;
        g_load<G>(P, E);                                 // keeps the state of masked-out envs intact (nothing is stored for them)
        if (E.valid) E.err_acc |= E.error;               // the episode being replaced may have raised error bits: they stay in the sticky word
        g_reset<G>(P, E, E.valid ? C.scen_idx[E.env] : E.scen, E.valid);
        if (E.valid && E.r == 0) {
            C.out.reward[E.env] = 0.0; C.out.done[E.env] = (uint8_t)E.done;
            C.out.status[3 * (size_t)E.env] = 0; C.out.status[3 * (size_t)E.env + 1] = 0; C.out.status[3 * (size_t)E.env + 2] = 0;
        }
        __syncthreads();
    } else {                                             // step(action): ENV:908-945
#ifdef FTL_NO_LATE_LOAD
        g_load<G>(P, E);
#else
        g_load<G, true, REG>(P, E);
#endif
        FTL_TIC(4);
        // One memory round trip for everything the frames need besides the state: the action and the scenario's static rects
        // (culled into the near list).
        double a0, a1;
        if (C.action_kind == FTL_ACTION_BOX2) { a0 = C.action[2 * (size_t)E.env]; a1 = C.action[2 * (size_t)E.env + 1]; }
        else if (C.action_kind == FTL_ACTION_DISCRETE) {        // ENV:918-922 with the table of ENV:362-367
            const int k = reinterpret_cast<const int32_t*>(C.action)[E.env];
            const double mr = P.cfg.follower.max_rotation_speed;
            a0 = P.cfg.follower.max_speed;
            a1 = k == 0 ? -mr : k == 1 ? -mr / 2 : k == 3 ? mr / 2 : k == 4 ? mr : 0.0;
            if (k < 0 || k > 4) E.error |= FTL_ERR_BAD_ACTION;    // reference: KeyError
        } else { a0 = 0.25; a1 = C.action[E.env]; }            // ENV:924-925

        // (this env's block bounding boxes stay in global memory: the searches that read them are rare -- an LDS copy per
        //  step measured 2 % slower than no copy once the search caches were in place, and cost 0.6 KB of traffic)
        g_build_near<G>(P, E, s_near, s_cnt, reinterpret_cast<float4*>(lds + P.fr_env_off + EPW * 4 + 16));
        if (E.r == 1) {
            command_forward(E.rb, L, a0);                                   // ENV:927
            if (a1 < 0) command_turn(E.rb, L, fabs(a1), -1);                // ENV:928-933
            else if (a1 > 0) command_turn(E.rb, L, a1, 1);
            else command_turn(E.rb, L, 0, 0);
        }
        double reward = 0; int i0 = 0, i1 = 0, i2 = 0;
        const int4* near = s_near + (size_t)E.slot * P.cfg.n_static;
        // frame records, pending position checks (LDS offsets from the host, FtlDevParams::fr_*)
        unsigned char* s_rec = lds + P.fr_rec_off;             // [slot][frame], rec_stride (a multiple of 16) bytes per env
        const int rec_stride = P.fr_rec_stride;
        FrPend* s_pend = reinterpret_cast<FrPend*>(lds + P.fr_pend_off);
        int* s_env = reinterpret_cast<int*>(lds + P.fr_env_off);
        int* s_pcnt = s_env + EPW;
        if (E.r == 0) s_env[E.slot] = E.env;
        if (threadIdx.x == 0) *s_pcnt = 0;
        const bool defer = P.fr_defer != 0;              // the later frames' searches wait for the end of the step
        int sc = E.step_count;                           // step_count as the tail sees it (the frames advance E.step_count themselves)
        double2 lead_cs;                                 // (cos, sin) of this lane's robot's direction as its last move left it: lane 0 = the leader's,
        sincos_bounded(E.rb.direction * kDeg2Rad, lead_cs.y, lead_cs.x);   // which the bears' way-points are built from (g_frame)
        __syncthreads();
        FTL_TIC(5);
        int tick = (E.step_count + 1) % P.cfg.trajectory_saving_period;
        // ENV:935-936; under random_frames_per_step every env has its own frame count this step (group-uniform branch)
#ifdef FTL_ABLATE_FRAMES
        const int f_max = FTL_ABLATE_FRAMES;
#else
        const int f_max = (REG && P.cfg.rand_fps_hi > 0) ? P.cfg.rand_fps_hi - 1 : P.cfg.frames_per_step;
#endif
#pragma nounroll
        for (int f = 0; f < f_max; f++) {
            bool pend = false; int pend_idx = -1; float new_ad = 3.0e38f;
            if (!REG || f < E.fps) g_frame<G, REG>(P, E, L, lead_cs, near, tick, f == 0, f, s_rec, rec_stride, s_pend, s_pcnt, pend, pend_idx, new_ad);
            __syncthreads();          // an appended trajectory point is read by the other lanes of the group next frame
            // The pending searches.  After the first frame (and after every frame when nothing is deferred) every env searches for
            // itself: the next frames live on the caches these searches refresh.  After the last frame of a deferring step: everything
            // the later frames left behind, sixteen items at a time, whichever envs they belong to.  (One call site: one copy of the code.)
            if (f == 0 || !defer || f == f_max - 1) {
                const int cnt = *s_pcnt;
                if (cnt > 0) {        // wave-uniform
                    const bool spread = defer && f > 0;
                    int hint; float hx, hy, cg, ca;
                    const int own = pend ? gb_i<G, 0>(pend_idx) : -1;
                    g_resolve<G>(P, E, s_pend, s_rec, rec_stride, s_env, cnt, spread, own, hint, hx, hy, cg, ca);
                    if (pend && !spread) {
                        E.hint = hint;
                        if (hx < 1.0e38f) { E.hx = hx; E.hy = hy; }
                        if (cg >= 0.0f) E.clr_g = fminf(cg, new_ad);      // (the search did not see a point this frame appended)
                        if (ca >= 0.0f) E.clr_a = fminf(ca, new_ad);
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) *s_pcnt = 0;
                    __syncthreads();
                }
            }
        }
        FTL_TIC(11);
#ifndef FTL_NO_LATE_LOAD
        g_load_late<REG>(P, E);
#endif
        const int done0 = E.done;
        // the tail of every frame (ENV:1077-1141), from the records
        const TailK TK = tail_consts(P.cfg);
#pragma nounroll
        for (int f0 = 0; f0 < f_max; f0 += 4) {        // four records per LDS read
            const unsigned w = *reinterpret_cast<const unsigned*>(s_rec + E.slot * rec_stride + f0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int f = f0 + j;
                if (f < f_max && (!REG || f < E.fps)) g_tail(TK, E, (int)((w >> (8 * j)) & 255u), sc, reward, i0, i1, i2);
            }
        }
        if (REG && P.cfg.rand_fps_hi > 0) E.fps = d_rand_frames(P.cfg, E.env, E.resets, E.step_count);      // ENV:939-940: the next step's frames
        if (E.valid && E.r == 0) {
            C.out.reward[E.env] = reward; C.out.done[E.env] = (uint8_t)E.done;
            C.out.status[3 * (size_t)E.env] = (uint8_t)i0; C.out.status[3 * (size_t)E.env + 1] = (uint8_t)i1; C.out.status[3 * (size_t)E.env + 2] = (uint8_t)i2;
        }
        FTL_TIC(9);
        // Episode metrics: the reference reports overall_reward / step_count when done is raised (ENV:941-944).
        // The record of a finishing episode goes to the env's "ep_stats" slot HERE, before an auto-reset wipes the counters;
        // ftl_episode_metrics() sums the slots.  Rare (one step in a few hundred per env), so one lane's read-modify-write will do.
        if (E.valid && E.r == 0 && E.done && (!done0 || (C.flags & FTL_STEP_AUTO_RESET))) {
            double* st = P.ep_stats + (size_t)E.env * FTL_N_METRICS;
            st[FTL_M_EPISODES] += 1.0; st[FTL_M_RETURN_SUM] += E.overall_reward; st[FTL_M_FRAMES_SUM] += (double)E.step_count;
            if (i0 == FTL_MISSION_SUCCESS) st[FTL_M_SUCCESS] += 1.0;
            if (i1 == FTL_AGENT_CRASH) st[FTL_M_CRASH] += 1.0;
            if (i1 == FTL_AGENT_LOW_REWARD) st[FTL_M_LOW_REWARD] += 1.0;
            if (i1 == FTL_AGENT_TOO_FAR) st[FTL_M_TOO_FAR] += 1.0;
            if (i0 == FTL_MISSION_FINISHED_BY_TIME) st[FTL_M_TIMEOUT] += 1.0;
        }
        bool go = E.valid && E.done && (C.flags & FTL_STEP_AUTO_RESET);
        if (__ballot(go) != 0ull) {
#ifdef FTL_PROFILE_PATHS
            if (threadIdx.x == 0) s_cyc[12] = 1;
#endif
#ifdef FTL_WAVE_TIMES
            wt_reset = true;
#endif
            if (go) { E.episodes += 1; E.err_acc |= E.error; }
            g_reset<G>(P, E, go ? C.win_base + ((E.scen % C.win_count) + C.win_stride) % C.win_count : E.scen, go);
            __syncthreads();
        }
    }
    FTL_TIC(6);
#ifndef FTL_ABLATE_SENSORS
    g_sensors<G>(P, E);                                  // ENV:937 / ENV:541 (tracker part of use_sensors)
#endif
    if (P.cfg.n_lasers > 0) {     // cos / sin of the follower's heading for the ray kernel: one sincos here serves the 16 envs of the wavefront
        double s, co;
        sincos_bounded(E.rb.direction * kDeg2Rad, s, co);
        if (E.valid && E.r == 1) { double* fc = rec_field(P.fol_cs, P, E.env); fc[0] = co; fc[1] = s; }
    }
    FTL_TIC(7);
    g_write_obs<G>(P, C, E);                             // ENV:938
    g_store<G>(P, E);
    if (P.keys && E.valid) {
        // cost class of this env's NEXT step, most expensive = largest: frames (random_frames_per_step), searches likely
        // (caches that will not outlast the step), a tracker scan that saves a point
        const float fpx = gb_f<G, 1>(E.rb.px), fpy = gb_f<G, 1>(E.rb.py);
        bool quiet = true;
        if (E.green_count > 2) (void)g_cache_class(P.cfg, E, fpx, fpy, E.fps, true, quiet);
        // tracker: envs whose counters are at most one scan apart save in the same steps and stay together from step to
        // step (all counters advance by two per step); class 0 = saves in the next step
        int tc = 0;
        if (P.cfg.has_tracker == 2) {
            const int per = P.cfg.tracker_saving_period;
            tc = (((E.trk_counter + 1) % per) * 4) / per;
        }
        int fb = 0;
        if (P.cfg.rand_fps_hi > 0) fb = ((E.fps - P.cfg.rand_fps_lo) * 8) / (P.cfg.rand_fps_hi - P.cfg.rand_fps_lo);
        // how often this env searched in the step that just ended predicts the next one better than the caches alone:
        // envs that hover around a threshold keep doing it
        int key;
        if (P.cfg.rand_fps_hi > 0) key = (fb << 3) | ((quiet && E.n_search <= 1 ? 0 : 1) << 2) | (3 - tc);
#ifdef FTL_KEY_NOSEARCH
        else key = (3 - tc);
#elif defined(FTL_KEY_QUIETONLY)
        else key = ((quiet ? 0 : 1) << 2) | (3 - tc);
#else
        else key = (min(E.n_search, 7) << 3) | ((quiet ? 0 : 1) << 2) | (3 - tc);
#endif
        if (E.r == 0) P.keys[E.env] = (uint8_t)key;
    }
    FTL_TIC(10);
#ifdef FTL_WAVE_TIMES
    {
        int ns = (E.valid && E.r == 0) ? E.n_search : 0, nw = (E.valid && E.r == 0) ? E.dbg_walk : 0, nf = (E.valid && E.r == 0) ? E.dbg_full : 0;
        for (int o = 32; o >= 1; o >>= 1) { ns += __shfl_xor(ns, o); nw += __shfl_xor(nw, o); nf += __shfl_xor(nf, o); }
        if (threadIdx.x == 0 && blockIdx.x < 8192) { g_wt[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); g_wi[blockIdx.x] = (wt_reset ? 1u : 0u) | ((unsigned)min(ns, 255) << 8) | ((unsigned)min(nw, 255) << 16) | ((unsigned)min(nf, 255) << 24); }
    }
#endif
#ifdef FTL_PROFILE_PATHS
    __syncthreads();
    if (threadIdx.x < 16) atomicAdd(&g_cyc[threadIdx.x], s_cyc[threadIdx.x]);
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_wave_t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), t0 = blockIdx.x < 8192 ? g_wave_t[2 * blockIdx.x] : t1;
        const bool heavy = __builtin_amdgcn_readfirstlane((int)(t1 - t0)) > 11500;       // 100 MHz ticks
        if (heavy && threadIdx.x < 16) atomicAdd(&g_cyc_heavy[threadIdx.x], s_cyc[threadIdx.x]);
        if (heavy && threadIdx.x == 0) atomicAdd(&g_cyc_heavy[16], 1ull);
    }
    if (threadIdx.x == 0 && C.mode != 1) {
        unsigned long long tot = 0; for (int i = 0; i < 11; i++) if (i != 9) tot += s_cyc[i];
        int bin = (int)(tot >> 12); bin = bin > 63 ? 63 : bin;
        atomicAdd(&g_whist[s_cyc[12] ? 1 : 0][bin], 1u);
    }
#endif
}

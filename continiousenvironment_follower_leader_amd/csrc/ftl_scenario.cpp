// ftl_generate_scenarios -- the scenario part of the reference's Game.reset(), host side.
//
// Follows, in the reference's order of `random` draws (so that python seed s reproduces `game.seed(s); game.reset()`):
//   _create_robots                ENV:545-595     leader start by randrange, follower placed behind it (first draw)
//   _create_obstacles             ENV:613-677     two bridge walls, obstacle_number 50x50 rocks by rejection sampling
//   generate_finish_point         ENV:1614-1630   rejection sampling against every game object
//   generate_trajectory_dstar     ENV:1493-1612   utils/dstar.py:84-210 -- a first D* run is Dijkstra from the goal
//   _create_dyn_obs/_reset_pose_bear  ENV:687-720, 761-770
//   _pos_follower_behind_leader   ENV:598-611     second follower draw, relative to the leader's new direction
//   initial leader_factual_trajectory  ENV:533-539  np.linspace(float32, float32) -> float32
// Third-party semantics restated here: CPython 3.10 `random` (MT19937 init_by_array, getrandbits, _randbelow_with_getrandbits,
// randrange), pygame.Rect integer truncation (tests/golden/gen/standins, parity unpinned at that boundary as in DESIGN.md 3),
// numpy float32 linspace, scipy euclidean on float32 operands.
// Not reproducible: which of several equal-cost routes dstar.py returns (min() over a set of objects hashed by id).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/ftl.h"

namespace {

// ---- CPython random.Random (Modules/_randommodule.c, Lib/random.py) ---------------------------------------------------
struct PyRandom {
    uint32_t mt[624];
    int idx;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t* key, int len) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > len ? 624 : len); k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    void seed(int64_t a) {                       // random.seed(int): key = 32-bit little-endian digits of abs(a)
        uint64_t u = a < 0 ? (uint64_t)(-(a + 1)) + 1u : (uint64_t)a;
        uint32_t key[2] = {(uint32_t)u, (uint32_t)(u >> 32)};
        init_by_array(key, key[1] ? 2 : 1);
    }
    uint32_t next() {
        if (idx >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            for (; kk < 623; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    uint32_t randbelow(uint32_t n) {             // _randbelow_with_getrandbits, n < 2^32
        int k = 0; for (uint32_t v = n; v; v >>= 1) k++;
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
    // randrange(start, stop, step) with positive step; ok=false where CPython raises ValueError (empty range)
    long randrange(long start, long stop, long step, bool& ok) {
        long width = stop - start;
        long n = step == 1 ? width : (width + step - 1) / step;
        if (n <= 0) { ok = false; return start; }
        return start + step * (long)randbelow((uint32_t)n);
    }
};

struct Rect {
    int x, y, w, h;
    int right() const { return x + w; }
    int bottom() const { return y + h; }
    bool collidepoint(double px, double py) const { return x <= px && px < x + w && y <= py && py < y + h; }
};
// image.get_rect(center=position, width=w, height=h) on an image already scaled to (w, h): CLS:42-50
inline Rect rect_at(float cx, float cy, int w, int h) { return Rect{(int)cx - (w >> 1), (int)cy - (h >> 1), w, h}; }

inline double angle_correction(double a) { return a >= 360 ? a - 360 : (a < 0 ? 360 + a : a); }   // MISC:6-13
inline double angle_to_point(double cx, double cy, double tx, double ty) {                         // MISC:16-26
    const double rx = tx - cx, ry = ty - cy;
    double res;
    if (rx > 0) res = atan(ry / rx) * (180.0 / M_PI);
    else if (rx < 0) res = atan(ry / rx) * (180.0 / M_PI) + 180;
    else res = 0;
    return angle_correction(res);
}
inline double radians(double d) { return d * (M_PI / 180.0); }
// scipy.spatial.distance.euclidean on two float32 vectors (oracle/ftl_oracle.c euclid_f32)
inline double euclid_f32(float ax, float ay, float bx, float by) {
    float dx = ax - bx, dy = ay - by;
    return (double)(float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}

// ---- utils/dstar.py: the first run() on a fresh map is Dijkstra from the goal; the route is the parent chain from start.
// Costs accumulate exactly as there (h_new = x.h + cost, cost = 1.0 or sqrt(2.0) as a double).  Returns false when the walk
// along the parents cannot reach the goal the way the reference's cannot (goal inside an inflated obstacle).
struct Grid {
    int rows, cols;
    std::vector<uint8_t> obst;
    bool in(int x, int y) const { return x >= 0 && x < rows && y >= 0 && y < cols; }
};

bool plan_route(const Grid& g, int sx, int sy, int gx, int gy, int max_iterat, std::vector<int>& rx, std::vector<int>& ry) {
    rx.clear(); ry.clear();
    if (!g.in(sx, sy) || !g.in(gx, gy)) return false;
    if (sx == gx && sy == gy) return true;                   // `while tmp != end` never runs: empty route
    // An obstacle goal or start makes every first move cost sys.maxsize: the reference then wanders through modify()
    // until max_iterat and reports found_target_point = False (pinned against the seeds the golden pool dropped).
    if (g.obst[(size_t)gx * g.cols + gy] || g.obst[(size_t)sx * g.cols + sy]) return false;
    // The open list in the reference's order -- smallest key first, insertion order among equal keys -- without a heap: every move
    // costs at least 1, so a state popped with key k only inserts keys >= k + 1, i.e. into LATER unit-wide buckets than its own.  When
    // bucket i comes up it is therefore complete; a stable sort by key (insertion order = sequence order already) puts it into pop
    // order.  Same pops, same relaxations, same parents as the binary heap this replaces, at a third of the time (the planner is
    // nine tenths of a scenario).  The grid carries a one-cell border of blocked cells, so the neighbour loop needs no bounds test.
    const int W2 = g.cols + 2, N2 = (g.rows + 2) * W2;
    const double INF = 1e300;
    struct Item { double k; int id; };
    static thread_local std::vector<double> h;               // per-thread scratch (thousands of scenarios per thread)
    static thread_local std::vector<int> parent;
    static thread_local std::vector<uint8_t> blocked;        // obstacle | closed | border
    static thread_local std::vector<std::vector<Item>> bucket;
    h.assign((size_t)N2, INF); parent.assign((size_t)N2, -1); blocked.assign((size_t)N2, 1);
    for (int x = 0; x < g.rows; x++) {
        const uint8_t* src = &g.obst[(size_t)x * g.cols];
        uint8_t* dst = &blocked[(size_t)(x + 1) * W2 + 1];
        for (int y = 0; y < g.cols; y++) dst[y] = src[y];
    }
    const int nb = g.rows + g.cols + 8;                      // keys stay below rows + cols (a path never needs more moves than that)
    if ((int)bucket.size() < nb) bucket.resize((size_t)nb);
    for (int i = 0; i < nb; i++) bucket[(size_t)i].clear();
    const int goal = (gx + 1) * W2 + gy + 1, start = (sx + 1) * W2 + sy + 1;
    const double SQ2 = sqrt(2.0);
    // neighbour order of Map.get_neighbors (dstar.py:62-74): i = -1..1 outer, j = -1..1 inner
    const int doff[8] = {-W2 - 1, -W2, -W2 + 1, -1, 1, W2 - 1, W2, W2 + 1};
    const double dcost[8] = {SQ2, 1.0, SQ2, 1.0, 1.0, SQ2, 1.0, SQ2};
    h[(size_t)goal] = 0.0;
    bucket[0].push_back(Item{0.0, goal});
    bool reached = false;
    for (int bi = 0; bi < nb && !reached; bi++) {
        std::vector<Item>& B = bucket[(size_t)bi];
        if (B.empty()) continue;
        std::stable_sort(B.begin(), B.end(), [](const Item& a, const Item& b) { return a.k < b.k; });
        for (size_t t = 0; t < B.size(); t++) {
            const Item it = B[t];
            if (blocked[(size_t)it.id] || it.k != h[(size_t)it.id]) continue;      // closed meanwhile / superseded entry
            blocked[(size_t)it.id] = 1;
            if (it.id == start) { reached = true; break; }
            for (int d = 0; d < 8; d++) {
                const int nid = it.id + doff[d];
                if (blocked[(size_t)nid]) continue;
                const double hn = it.k + dcost[d];
                if (hn < h[(size_t)nid]) {
                    h[(size_t)nid] = hn; parent[(size_t)nid] = it.id;
                    const int to = (int)hn;
                    if (to >= nb) return false;              // (cannot happen on a grid this size)
                    bucket[(size_t)to].push_back(Item{hn, nid});
                }
            }
        }
    }
    if (!reached) return false;                               // unreachable goal: the reference would never return
    int cur = start, iter = 0;
    while (cur != goal) {
        if (++iter > max_iterat) return false;
        rx.push_back(cur / W2 - 1); ry.push_back(cur % W2 - 1);
        cur = parent[(size_t)cur];
        if (cur < 0) return false;
    }
    return true;
}

// ---- utils/astar.py:50-166, literally: nodes ordered by f = g + h with h = SQUARED distance to the goal, CPython heapq (ties keep the
// heap's own order), a closed LIST, duplicates in the open list unless an open node at the same cell has a smaller g, and a cap of
// 1000 expansions after which the path to the LAST expanded node is returned (return_none_on_max_iter=False, ENV:1681-1695).
// Returns false when the open list runs dry (the reference returns None).  Path cells are appended to px / py in grid units.
struct ANode { int x, y, g, f, parent; };
bool astar_route(const std::vector<uint8_t>& maze, int gw, int gh, int sx, int sy, int ex, int ey, int max_iterations,
                 std::vector<int>& px, std::vector<int>& py) {
    std::vector<ANode> nodes;                       // every node ever created (parent = index)
    std::vector<int> heap;                          // open_list as CPython's heapq keeps it (indices into nodes)
    std::vector<int> closed;                        // closed_list (indices), membership by position
    std::vector<uint8_t> in_closed((size_t)gw * gh, 0);
    auto lt = [&](int a, int b) { return nodes[(size_t)a].f < nodes[(size_t)b].f; };
    auto siftdown = [&](int startpos, int pos) {
        const int item = heap[(size_t)pos];
        while (pos > startpos) {
            const int pp = (pos - 1) >> 1;
            if (lt(item, heap[(size_t)pp])) { heap[(size_t)pos] = heap[(size_t)pp]; pos = pp; continue; }
            break;
        }
        heap[(size_t)pos] = item;
    };
    auto siftup = [&](int pos) {
        const int endpos = (int)heap.size(), startpos = pos, item = heap[(size_t)pos];
        int child = 2 * pos + 1;
        while (child < endpos) {
            const int right = child + 1;
            if (right < endpos && !lt(heap[(size_t)child], heap[(size_t)right])) child = right;
            heap[(size_t)pos] = heap[(size_t)child]; pos = child; child = 2 * pos + 1;
        }
        heap[(size_t)pos] = item;
        siftdown(startpos, pos);
    };
    auto push = [&](int n) { heap.push_back(n); siftdown(0, (int)heap.size() - 1); };
    auto pop = [&]() { const int last = heap.back(); heap.pop_back(); if (heap.empty()) return last; const int ret = heap[0]; heap[0] = last; siftup(0); return ret; };
    auto emit = [&](int n) {                       // return_path: goal-to-start chain, reversed
        std::vector<int> chain;
        for (int c = n; c >= 0; c = nodes[(size_t)c].parent) chain.push_back(c);
        for (size_t i = chain.size(); i-- > 0;) { px.push_back(nodes[(size_t)chain[i]].x); py.push_back(nodes[(size_t)chain[i]].y); }
    };
    nodes.push_back(ANode{sx, sy, 0, 0, -1});
    push(0);
    static const int dx8[8] = {0, 0, -1, 1, -1, -1, 1, 1}, dy8[8] = {-1, 1, 0, 0, -1, 1, -1, 1};
    int outer = 0, current = -1;
    while (!heap.empty()) {
        outer++;
        if (outer > max_iterations) { emit(current); return true; }       // "giving up on pathfinding too many iterations"
        current = pop();
        closed.push_back(current);
        const ANode cur = nodes[(size_t)current];
        if (cur.x >= 0 && cur.x < gw && cur.y >= 0 && cur.y < gh) in_closed[(size_t)cur.x * gh + cur.y] = 1;
        if (cur.x == ex && cur.y == ey) { emit(current); return true; }
        for (int d = 0; d < 8; d++) {
            const int nx = cur.x + dx8[d], ny = cur.y + dy8[d];
            if (nx > gw - 1 || nx < 0 || ny > gh - 1 || ny < 0) continue;
            if (maze[(size_t)nx * gh + ny] != 0) continue;
            if (in_closed[(size_t)nx * gh + ny]) continue;                  // child is on the closed list
            const int g = cur.g + 1, h = (nx - ex) * (nx - ex) + (ny - ey) * (ny - ey);
            bool worse = false;                                          // an open node at this cell with a smaller g
            for (int o : heap) if (nodes[(size_t)o].x == nx && nodes[(size_t)o].y == ny && g > nodes[(size_t)o].g) { worse = true; break; }
            if (worse) continue;
            nodes.push_back(ANode{nx, ny, g, g + h, current});
            push((int)nodes.size() - 1);
        }
    }
    return false;                                                        // "Couldn't get a path to destination"
}

struct Obj { Rect r; float px, py; int w, h; };             // GameObject: rectangle, float32 start_position, height/width

void generate_one(const ftl_config& c, const ftl_scen_params& sp, int64_t seed, int idx, const ftl_scenarios& out, uint8_t* status) {
    const int R = 2 + c.n_bears;
    PyRandom rnd; rnd.seed(seed);
    bool ok = true;
    unsigned st = 0;
    const int W = sp.width, H = sp.height, sg = sp.step_grid;
    // ---- _create_robots (ENV:545-595)
    const long lx = rnd.randrange((long)(W / 2.0 + sp.max_distance), (long)(W - sp.max_distance), 10, ok);
    const long ly = rnd.randrange((long)sp.max_distance, (long)(H - sp.max_distance), 10, ok);
    const double ldir0 = angle_to_point((double)lx, (double)ly, (double)(long)(W / 2.0), (double)(long)(H / 2.0));   // np.array(..., dtype=int)
    const float lpx = (float)lx, lpy = (float)ly;
    const Rect lrect = rect_at(lpx, lpy, c.leader.img_w, c.leader.img_h);
    Rect frect0;
    {
        const long d = rnd.randrange((long)(sp.min_distance * 1.1), (long)(sp.max_distance * 0.9), 1, ok);
        const double th = radians(angle_correction(ldir0 + 180));
        const double fx = (double)d * cos(th) + (double)lx, fy = (double)d * sin(th) + (double)ly;
        frect0 = rect_at((float)fx, (float)fy, c.follower.img_w, c.follower.img_h);
    }
    // ---- _create_obstacles (ENV:613-677); game_object_list = [leader, follower, wall1, wall2, rocks...]
    std::vector<Obj> objs;           // statics only, in game_object_list order
    if (sp.add_obstacles) {
        const int boh = (H - sp.bridge_gap) / 2;                              // bridge_obstacle_height
        const float m1x = (float)(W / 2.0), m1y = (float)(boh / 2);
        const float m2y = (float)((H / 2) + (boh / 2) + (sp.bridge_gap / 2));
        Obj w1{rect_at(m1x, m1y, sp.bridge_width, boh), m1x, m1y, sp.bridge_width, boh};
        Obj w2{rect_at(m1x, m2y, sp.bridge_width, boh), m1x, m2y, sp.bridge_width, boh};
        const int wall_start_x = w1.r.x, wall_end_x = w1.r.right();
        // pygame.Rect(...) truncates each float argument toward zero
        const Rect bridge{(int)(wall_start_x - sp.leader_w * 4), (int)(w1.r.bottom() - sp.leader_h * sp.leader_margin),
                          (int)(w1.r.w + 8 * sp.leader_w), (int)(w2.r.y - w1.r.bottom() + 3 * sp.leader_h)};
        const int osz = 50;
        std::vector<Obj> rocks;
        for (int i = 0; i < sp.obstacle_number && ok; i++) {
            long gx2, gy2;
            for (;;) {
                gx2 = rnd.randrange(130, W - 120, sg, ok); gy2 = rnd.randrange(20, H - 20, sg, ok);
                if (!ok) break;
                const double ddx = (double)lpx - (double)gx2, ddy = (double)lpy - (double)gy2;
                const bool busy = lrect.collidepoint((double)gx2, (double)gy2) || frect0.collidepoint((double)gx2, (double)gy2) ||
                                  (gx2 >= wall_start_x && gx2 <= wall_end_x) || bridge.collidepoint((double)gx2, (double)gy2) ||
                                  sqrt(ddx * ddx + ddy * ddy) <= sp.max_distance + osz / 2.0;
                if (!busy) break;
            }
            rocks.push_back(Obj{rect_at((float)gx2, (float)gy2, osz, osz), (float)gx2, (float)gy2, osz, osz});
        }
        objs.push_back(w1); objs.push_back(w2);
        objs.insert(objs.end(), rocks.begin(), rocks.end());
    }
    // ---- generate_finish_point (ENV:1614-1630) against [leader, follower (as first placed), statics]
    auto finish_point = [&](long x0, long y0, long x1, long y1, long& fx, long& fy) {
        std::vector<Rect> all; all.push_back(lrect); all.push_back(frect0);
        for (const Obj& o : objs) all.push_back(o.r);
        for (;;) {
            fx = rnd.randrange(x0, x1, 10, ok); fy = rnd.randrange(y0, y1, 10, ok);
            if (!ok) return;
            bool good = true;
            for (const Rect& r : all) {
                if (r.collidepoint((double)fx, (double)fy)) { good = false; continue; }
                // distance_to_rect (MISC:29-44): corners and edge mid-points, integer coordinates
                const int qx[8] = {r.x, r.x, r.x + r.w, r.x + r.w, r.x + (r.w >> 1), r.x, r.x + (r.w >> 1), r.x + r.w};
                const int qy[8] = {r.y, r.y + r.h, r.y, r.y + r.h, r.y, r.y + (r.h >> 1), r.y + r.h, r.y + (r.h >> 1)};
                double md = INFINITY;
                for (int k = 0; k < 8; k++) { double dx = (double)(fx - qx[k]), dy = (double)(fy - qy[k]); md = fmin(md, sqrt(dx * dx + dy * dy)); }
                if (md < sp.leader_pos_epsilon) good = false;
            }
            if (good) return;
        }
    };
    long f1x = 0, f1y = 0, f2x = 0, f2y = 0, f3x = 0, f3y = 0;
    const bool fixed = sp.planner == 2;                                        // trajectory= of the constructor: ENV:470 skips all of this
    if (!fixed) finish_point(20, 20, (long)(W / 2.0), H - 20, f1x, f1y);
    if (!fixed && sp.multiple_end_points && ok) {                                       // ENV:470-481
        if (f1y >= H / 2.0) finish_point(20, 20, W - 20, (long)(H / 2.0), f2x, f2y);
        else finish_point(20, (long)(H / 2.0), W - 20, H - 20, f2x, f2y);
        if (ok) {
            if (f2y >= H / 2.0) finish_point(20, 20, W - 20, (long)(H / 2.0), f3x, f3y);
            else finish_point(20, (long)(H / 2.0), W - 20, H - 20, f3x, f3y);
        }
    }
    std::vector<double> route_x, route_y;
    bool found = ok;
    if (ok && fixed)
        for (int i = 0; i < sp.fixed_route_len; i++) { route_x.push_back(sp.fixed_route[2 * i]); route_y.push_back(sp.fixed_route[2 * i + 1]); }
    if (ok && sp.planner == 1) {
        // ---- generate_trajectory_astar (ENV:1632-1711): a 20 px grid, obstacles inflated by 2 x the leader's larger side, the bridge row
        // cleared, one leg to the near end of the bridge and one from its far end to the finish point
        const int a_sg = 20;
        const int sx = (int)(lpx / (float)a_sg), sy = (int)(lpy / (float)a_sg);
        const int ex = (int)((double)f1x / a_sg), ey = (int)((double)f1y / a_sg);
        const int gw = (int)((double)W / a_sg), gh = (int)((double)H / a_sg);
        std::vector<uint8_t> maze((size_t)gw * gh, 0);
        const int lsf = (int)(fmax(sp.leader_w, sp.leader_h) * 2);
        for (const Obj& o : objs) {
            const int x0 = std::max((int)((double)(o.r.x - lsf) / a_sg), 0), x1 = std::min((int)((double)(o.r.right() + lsf) / a_sg), gw - 1);
            const int y0 = std::max((int)((double)(o.r.y - lsf) / a_sg), 0), y1 = std::min((int)((double)(o.r.bottom() + lsf) / a_sg), gh - 1);
            for (int x = x0; x < x1; x++) for (int y = y0; y < y1; y++) maze[(size_t)x * gh + y] = 1;
        }
        std::vector<int> px, py;
        bool got = true;
        if (sp.add_obstacles && objs.size() >= 2) {
            const Obj& w1 = objs[0]; const Obj& w2 = objs[1];
            // self.bridge_point: float32 mean of the two wall centres (ENV:636-637), / 20 in float32, truncated
            const float bpx = (float)(((double)w1.px + (double)w2.px) / 2), bpy = (float)(((double)w1.py + (double)w2.py) / 2);
            const int bx = (int)(bpx / (float)a_sg), by = (int)(bpy / (float)a_sg);
            auto clear = [&](int x, int y) {              // numpy indexing: a negative index wraps around
                if (x < 0) x += gw; if (y < 0) y += gh;
                if (x >= 0 && x < gw && y >= 0 && y < gh) maze[(size_t)x * gh + y] = 0;
            };
            clear(bx, by);
            for (int i = (int)(((double)w1.r.x / a_sg) - ((double)lsf / a_sg)); i < (int)(((double)w1.r.right() / a_sg) + ((double)lsf / a_sg)); i++) clear(i, by);
            const int fbx = (int)(((double)w1.r.right() + sp.leader_pos_epsilon) / a_sg), sbx = (int)(((double)w1.r.x - sp.leader_pos_epsilon) / a_sg);
            got = astar_route(maze, gw, gh, sx, sy, fbx, by, 1000, px, py);
            if (got) {
                for (size_t i = 0; i < px.size(); i++) { route_x.push_back((long)px[i] * a_sg); route_y.push_back((long)py[i] * a_sg); }
                // `if path[-1] != first_bridge_point` compares a pixel pair with a grid pair (ENV:1684): they only coincide at the origin
                if (!(route_x.back() == fbx && route_y.back() == by)) { route_x.push_back((long)((double)w1.r.right() + sp.leader_pos_epsilon)); route_y.push_back((long)a_sg * by); }
                px.clear(); py.clear();
                if (astar_route(maze, gw, gh, sbx, by, ex, ey, 1000, px, py))
                    for (size_t i = 0; i < px.size(); i++) { route_x.push_back((long)px[i] * a_sg); route_y.push_back((long)py[i] * a_sg); }
            }
        } else {
            got = astar_route(maze, gw, gh, sx, sy, ex, ey, 1000, px, py);
            if (got) for (size_t i = 0; i < px.size(); i++) { route_x.push_back((long)px[i] * a_sg); route_y.push_back((long)py[i] * a_sg); }
        }
        found = route_x.size() >= 2;                  // the reference leaves found_target_point False (ENV:1537 is D*-only): "usable" = it can be stepped
    }
    // ---- generate_trajectory_dstar (ENV:1493-1612)
    if (ok && sp.planner == 0) {
        Grid g; g.rows = W / sg; g.cols = H / sg; g.obst.assign((size_t)g.rows * g.cols, 0);
        const int margin = (int)floor(sp.leader_margin * fmax(sp.leader_w, sp.leader_h) / sg);
        // order of the reference: rocks, then the two walls (irrelevant for a set of cells)
        for (const Obj& o : objs) {
            const int pmx = (int)floorf(o.px / (float)sg), pmy = (int)floorf(o.py / (float)sg);
            const int hh = (int)floor((o.h / 2.0) / sg) + margin, hw = (int)floor((o.w / 2.0) / sg) + margin;
            for (int i = pmx - hw; i < pmx + hw; i++)
                for (int j = pmy - hh; j < pmy + hh; j++)
                    if (g.in(i, j)) g.obst[(size_t)i * g.cols + j] = 1;
        }
        std::vector<int> rx, ry;
        int sx = (int)(lpx / (float)sg), sy = (int)(lpy / (float)sg);
        const long gxs[3] = {f1x, f2x, f3x}, gys[3] = {f1y, f2y, f3y};
        const int runs = sp.multiple_end_points ? 3 : 1;
        for (int k = 0; k < runs; k++) {
            const int gx = (int)((double)gxs[k] / sg), gy = (int)((double)gys[k] / sg);
            // the 2nd and 3rd planner of the reference use the default max_iterat (Dstar(m2), dstar.py:85)
            const bool f = plan_route(g, sx, sy, gx, gy, k == 0 ? sp.path_finding_iterations : 15000, rx, ry);
            found = found && f;
            for (size_t i = 0; i < rx.size(); i++) { route_x.push_back((long)rx[i] * sg); route_y.push_back((long)ry[i] * sg); }
            sx = gx; sy = gy;
        }
    }
    if (found) st |= FTL_SCEN_FOUND;
    const int rl = (int)route_x.size();
    if (rl == 0) st |= FTL_SCEN_DONE_AT_RESET;
    if (rl == 1) st |= FTL_SCEN_REF_RAISES;
    // ---- leader direction, follower behind the leader (ENV:506-525, 598-611)
    double ldir = ldir0;
    float fpx = lpx, fpy = lpy; double fdir = 0;
    if (ok) {
        double tx = (double)lpx, ty = (double)lpy;              // len(trajectory) == 0: cur_target_point = leader.start_position
        if (rl >= 2) { tx = route_x[1]; ty = route_y[1]; }
        ldir = angle_to_point((double)lpx, (double)lpy, tx, ty);
        const long d = rnd.randrange((long)(sp.min_distance * 1.1), (long)(sp.max_distance * 0.9), 1, ok);
        const double th = angle_correction(ldir + 180);
        const double fx = (double)d * cos(radians(th)) + (double)lpx, fy = (double)d * sin(radians(th)) + (double)lpy;
        fdir = angle_to_point(fx, fy, (double)lpx, (double)lpy);
        fpx = (float)fx; fpy = (float)fy;
    }
    // ---- outputs
    int32_t* srect = const_cast<int32_t*>(out.static_rects) + (size_t)idx * c.n_static * 4;
    for (int s = 0; s < c.n_static; s++) {
        Rect r = s < (int)objs.size() ? objs[(size_t)s].r : Rect{0, 0, 0, 0};
        srect[4 * s] = r.x; srect[4 * s + 1] = r.y; srect[4 * s + 2] = r.w; srect[4 * s + 3] = r.h;
    }
    float* rp = const_cast<float*>(out.robot_pos) + (size_t)idx * R * 2;
    double* rd = const_cast<double*>(out.robot_dir) + (size_t)idx * R;
    int32_t* rr = const_cast<int32_t*>(out.robot_rect) + (size_t)idx * R * 4;
    auto put = [&](int r, float x, float y, double dir, Rect q) {
        rp[2 * r] = x; rp[2 * r + 1] = y; rd[r] = dir; rr[4 * r] = q.x; rr[4 * r + 1] = q.y; rr[4 * r + 2] = q.w; rr[4 * r + 3] = q.h;
    };
    put(0, lpx, lpy, ldir, lrect);
    put(1, fpx, fpy, fdir, rect_at(fpx, fpy, c.follower.img_w, c.follower.img_h));
    for (int b = 0; b < c.n_bears; b++) {                     // _reset_pose_bear (ENV:761-770); float32 arithmetic on leader.position
        const float bx = (b % 2 == 0) ? lpx + 150.0f : lpx - 150.0f, by = (b % 2 == 0) ? lpy - 150.0f : lpy + 150.0f;
        put(2 + b, bx, by, 0.0, rect_at(bx, by, c.bear.img_w, c.bear.img_h));
    }
    double* ro = const_cast<double*>(out.route) + (size_t)idx * c.route_cap * 2;
    if (rl > c.route_cap) st |= FTL_SCEN_ROUTE_OVERFLOW;
    const int rn = rl < c.route_cap ? rl : c.route_cap;
    for (int i = 0; i < rn; i++) { ro[2 * i] = route_x[(size_t)i]; ro[2 * i + 1] = route_y[(size_t)i]; }
    for (int i = rn; i < c.route_cap; i++) { ro[2 * i] = 0; ro[2 * i + 1] = 0; }
    const_cast<int32_t*>(out.route_len)[idx] = rn;
    // ---- initial leader_factual_trajectory (ENV:533-539): float32 linspace follower -> leader
    float* it = const_cast<float*>(out.init_traj) + (size_t)idx * c.init_traj_cap * 2;
    int n = (int)(euclid_f32(fpx, fpy, lpx, lpy) / (sp.trajectory_saving_period * sp.leader_max_speed));
    if (n < 0) n = 0;
    if (n > c.init_traj_cap) { st |= FTL_SCEN_TRAJ_OVERFLOW; n = c.init_traj_cap; }
    if (n == 1) { it[0] = fpx; it[1] = fpy; }
    else if (n > 1) {
        const float div = (float)(n - 1);
        const float dxx = lpx - fpx, dyy = lpy - fpy;
        const float stepx = dxx / div, stepy = dyy / div;
        for (int i = 0; i < n; i++) {
            it[2 * i] = (stepx == 0) ? ((float)i / div) * dxx + fpx : (float)i * stepx + fpx;
            it[2 * i + 1] = (stepy == 0) ? ((float)i / div) * dyy + fpy : (float)i * stepy + fpy;
        }
        it[2 * (n - 1)] = lpx; it[2 * (n - 1) + 1] = lpy;
    }
    for (int i = n; i < c.init_traj_cap; i++) { it[2 * i] = 0; it[2 * i + 1] = 0; }
    const_cast<int32_t*>(out.init_traj_len)[idx] = n;
    if (!ok) st = FTL_SCEN_REF_RAISES;                          // an empty randrange: CPython raises ValueError
    status[idx] = (uint8_t)st;
}

}  // namespace

extern "C" int ftl_generate_scenarios(const ftl_config* cfg, const ftl_scen_params* sp, const int64_t* seeds, int32_t n,
                                      int32_t n_threads, const ftl_scenarios* out, uint8_t* status) {
    if (!cfg || !sp || !seeds || !out || !status || n < 0) return FTL_E_INVALID;
    if (!out->static_rects || !out->robot_pos || !out->robot_dir || !out->robot_rect || !out->route || !out->route_len ||
        !out->init_traj || !out->init_traj_len) return FTL_E_INVALID;
    if (sp->step_grid <= 0 || sp->width <= 0 || sp->height <= 0 || sp->trajectory_saving_period <= 0 || !(sp->leader_max_speed > 0)) return FTL_E_INVALID;
    if (cfg->n_static != (sp->add_obstacles ? sp->obstacle_number + 2 : 0)) return FTL_E_INVALID;
    if (cfg->n_bears != (sp->add_bear ? sp->bear_number : 0)) return FTL_E_INVALID;
    if (sp->planner < 0 || sp->planner > 2 || (sp->planner == 2 && (sp->fixed_route_len < 0 || (sp->fixed_route_len > 0 && !sp->fixed_route)))) return FTL_E_INVALID;
    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > n) T = n > 0 ? n : 1;
    std::atomic<int> next(0);
    auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) generate_one(*cfg, *sp, seeds[i], i, *out, status); };
    if (T == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    return FTL_OK;
}

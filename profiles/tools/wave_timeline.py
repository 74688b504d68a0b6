"""Diagnostic (build with -DFTL_WAVE_TIMES, FTL_LIB=that .so): start / end of every frame-kernel wavefront of one launch of a bench
workload (FTL_TIMELINE_WORKLOAD, default B) in its steady state -- how the wavefronts pack onto the 2,048 slots.  Not part of the product or the tests."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
W = os.environ.get("FTL_TIMELINE_WORKLOAD", "B")
from continiousenvironment_follower_leader_amd import shard
n = int(os.environ.get("FTL_DIAG_N", "0")) or shard.plan(W, 0, 1)[0].n
cfg, pool, *_ = bench.build_workload(W, 0, 0, torch.device("cuda:0"))
env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(pool)
env.reset((torch.arange(n) % pool.n).to(torch.int32))
acts = bench.make_actions(cfg, n, 16, 0, torch.device("cuda:0"))
for k in range(320): env.step(acts[k % 16], auto_reset=True)
T = (C.c_ulonglong * (2 * 8192))(); I = (C.c_uint * 8192)()
for rep in range(3):
    env.step(acts[rep], auto_reset=True)
    env.lib.ftl_debug_wave_timeline(T, I)
    nw = (n + 15) // 16 if (cfg.n_robots <= 4 and n > 8192) else (n + 7) // 8      # (small batches run 8 envs per wavefront)
    t = np.array(list(T), dtype=np.int64).reshape(8192, 2)[:nw]; info = np.array(list(I), dtype=np.int64)[:nw]
    t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0; d = en - st
    rs = (info & 1) == 1; ns = (info >> 8) & 255; nw = (info >> 16) & 255; nf = (info >> 24) & 255
    print("launch %d: kernel %.1f us; wave lifetime mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; sum/2048 slots = %.1f us" % (
        rep, en.max(), d.mean(), *np.percentile(d, [10, 50, 90, 99]), d.max(), d.sum() / 2048))
    first = st < 5.0
    if (~first).any():
        print("  first round: %d waves, lifetime mean %.1f (ends p50 %.1f p90 %.1f max %.1f); second round: %d waves, lifetime mean %.1f, start p10 %.1f p50 %.1f p90 %.1f" % (
            first.sum(), d[first].mean(), *np.percentile(en[first], [50, 90]), en[first].max(), (~first).sum(), d[~first].mean(), *np.percentile(st[~first], [10, 50, 90])))
    else:
        print("  one round: every wavefront starts within 5 us")
    print("  waves with a reset: %d, lifetime mean %.1f vs %.1f without; searches per wave mean %.1f; lifetime by searches: %s" % (
        rs.sum(), d[rs].mean() if rs.any() else 0, d[~rs].mean(), ns.mean(),
        " ".join("%d:%.0f(%d)" % (k, d[ns == k].mean(), (ns == k).sum()) for k in range(0, 40, 4) if (ns == k).any())))
    print("  busy slots per 10 us:", [int(((st <= x) & (en > x)).sum()) for x in np.arange(0, en.max(), 10)])
    slow = np.argsort(-d)[:24]
    print("  slowest waves (lifetime us, start us, reset, searches, exact green walks, whole-trajectory searches):", [(round(float(d[i]), 0), round(float(st[i]), 0), int(rs[i]), int(ns[i]), int(nw[i]), int(nf[i])) for i in slow])
    print("  lifetime by exact walks: %s; by whole-trajectory searches: %s" % (" ".join("%d:%.0f(%d)" % (k, d[nw == k].mean(), (nw == k).sum()) for k in range(0, 12) if (nw == k).any()), " ".join("%d:%.0f(%d)" % (k, d[nf == k].mean(), (nf == k).sum()) for k in range(0, 24, 2) if (nf == k).any())))
    order = np.argsort(st); print("  launch order vs lifetime (blocks of 512 waves in start order): ", [round(float(d[order[i:i + 512]].mean()), 1) for i in range(0, len(order), 512)])

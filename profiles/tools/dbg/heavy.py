"""Diagnostic (-DFTL_PROFILE_PATHS -DFTL_PROFILE_NOCOUNT build): per-section cycles of ALL frame-kernel wavefronts against those of the
wavefronts that run longer than FTL_HEAVY_TICKS x 10 ns (115 us by default), steady state of a bench workload (FTL_TIMELINE_WORKLOAD)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
W = os.environ.get("FTL_TIMELINE_WORKLOAD", "B")
n = bench.DEFAULT_ENVS[W]
cfg, pool, *_ = bench.build_workload(W, n, 0, 0, torch.device("cuda:0"))
env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(pool)
env.reset((torch.arange(n) % pool.n).to(torch.int32))
acts = bench.make_actions(cfg, n, 16, 0, torch.device("cuda:0"))
for k in range(320): env.step(acts[k % 16], auto_reset=True)
A = (C.c_ulonglong * 48)(); H = (C.c_ulonglong * 17)()
env.lib.ftl_debug_prof(A, 1); env.lib.ftl_debug_heavy(H, 1)
steps = 50
for k in range(steps): env.step(acts[k % 16], auto_reset=True)
env.lib.ftl_debug_prof(A, 0); env.lib.ftl_debug_heavy(H, 0)
cn16 = {13: "episode metrics record", 14: "g_reset", 15: "g_reset: probe"}
cn = ["frame:move+collide", "frame:green", "frame:agent check (after hint)", "frame:tail", "load", "near+bb stage", "auto-reset", "sensors(tracker)", "frame:hint window", "[frames loop total]", "obs+store", "frame:green search"]
allc = np.array(list(A)[16:28], dtype=np.float64); hv = np.array(list(H)[:12], dtype=np.float64); nh = max(int(H[16]), 1); nall = steps * n // 16
print("waves %d, heavy %d (%.1f %%)" % (nall, nh, 100.0 * nh / nall))
print("%-32s %12s %12s" % ("section (kcycles per wave-step)", "all", "heavy"))
for i in range(12): print("%-32s %12.1f %12.1f" % (cn[i], allc[i] / nall / 1e3, hv[i] / nh / 1e3))
for i, nm in cn16.items(): print("%-32s %12.1f %12.1f" % (nm, list(A)[16 + i] / nall / 1e3, list(H)[i] / nh / 1e3))

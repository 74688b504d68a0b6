"""Diagnostic (build with -DFTL_PROFILE_PATHS, FTL_LIB=that .so): branch frequencies of the frame kernel in the
steady state of the bench workload.  Not part of the product or the tests."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
n = int(os.environ.get("FTL_DIAG_N", "65536"))
env = VecGame(n, device="cuda:0", config=cfg); pool = ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"); env.load_scenarios(pool)
GROUP = int(os.environ.get("FTL_DIAG_GROUP", "1"))      # >1: make every GROUP consecutive envs identical (no divergence inside a wavefront)
env.reset(((torch.arange(n) // GROUP) % pool.n).to(torch.int32))
acts = bench.make_actions(cfg, n // GROUP, 16, 0, torch.device("cuda:0")).repeat_interleave(GROUP, dim=1).contiguous()
out = (C.c_ulonglong * 48)()
names = ["frames", "Gc>2", "fast path", "green search", "not quiet @frame0: on trace", "not quiet @frame0: dev band", "not quiet @frame0: nothing in reach", "not quiet @frame0: old trajectory",
         "later-frame search by an env predicted quiet", "first-frame search by an env predicted quiet", "search: in eps", "search: in dev", "undecided @frame0", "search in the first frame of a step", "search in a later frame", "first frames"]
for phase, steps in (("steps 0-30", 30), ("steps 30-150", 120), ("steps 150-250", 100)):
    env.lib.ftl_debug_prof(out, 1); env.lib.ftl_debug_whist((C.c_uint * 128)(), 1)
    for k in range(steps): env.step(acts[k % 16], auto_reset=True)
    env.lib.ftl_debug_prof(out, 1)
    v = list(out); fr = max(v[0], 1)
    cn = ["frame:collide", "frame:green", "frame:agent check", "frame:rest", "load", "near+bb stage", "auto-reset+metrics", "sensors(tracker)", "-", "tail+outputs", "obs+store+keys",
          "[frame loop total, resolves included]", "(reset flag)", "frame:way-point switch", "frame:bears+steer", "frame:robot_move"]
    cy = v[16:32]; per = steps * n / 16
    inloop = cy[0] + cy[1] + cy[2] + cy[3] + cy[13] + cy[14] + cy[15]
    tot = cy[4] + cy[5] + cy[6] + cy[7] + cy[9] + cy[10] + cy[11]
    print(phase, "cycles per wave-step:", " ".join("%s=%.0f" % (cn[i], cy[i] / per) for i in range(16) if i not in (8, 12)), "| resolve+sync (loop - frame sections)=%.0f" % ((cy[11] - inloop) / per),
          "| total=%.0f" % (tot / per))
    rc = v[32:48]
    if sum(rc[:8]):
        rn = ["stage corridor (2nd trip)", "phase1 table", "phase2 ray ends", "phase3 decode", "phase3 arcs", "phase3 ray tests", "phase4 rows", "setup (1st trip)"]
        rt = sum(rc[:8])
        print(phase, "RAYS cycles share:", " ".join("%s=%.3f" % (rn[i], rc[i] / rt) for i in range(8)), "cycles/env-step=%.0f" % (rt / (steps * n)),
              "chunks/env-step=%.2f max-cnt/chunk=%.2f sum-cnt/chunk=%.1f items/chunk=%.1f" % (rc[8] / (steps * n), rc[9] / max(rc[8], 1), rc[10] / max(rc[8], 1), rc[11] / max(rc[8], 1)),
              "chunks by max cnt: <=2 %.3f  3-4 %.3f  5-8 %.3f  >8 %.3f" % tuple(rc[12 + j] / max(rc[8], 1) for j in range(4)))
    print(phase, " ".join("%s=%.4f" % (names[i], v[i] / fr) for i in range(1, 16)), "waves/frame-wave: green %.3f full %.3f" % (v[8] / (fr / 16), v[9] / (fr / 16)))
wh = (C.c_uint * 128)()
env.lib.ftl_debug_whist(wh, 0)
wh = np.array(list(wh)).reshape(2, 64)
for k, nm in enumerate(("no reset", "with reset")):
    tot = wh[k].sum(); cs = np.cumsum(wh[k]) / max(tot, 1)
    print("frame-kernel wave lifetime (last phase), %s: waves=%d mean=%.0f kcyc p10=%d p50=%d p90=%d p99=%d max=%d (x4096 cycles)" % (
        nm, tot, (wh[k] * (np.arange(64) + 0.5) * 4.096).sum() / max(tot, 1), np.searchsorted(cs, 0.1), np.searchsorted(cs, 0.5), np.searchsorted(cs, 0.9), np.searchsorted(cs, 0.99), np.nonzero(wh[k])[0].max() if tot else 0))
wt = (C.c_ulonglong * (2 * 8192))()
env.lib.ftl_debug_wave_times(wt)
wt = np.array(list(wt), dtype=np.int64).reshape(8192, 2)[:n // 16]
t0 = wt[:, 0].min()
st, en = (wt[:, 0] - t0) / 100.0, (wt[:, 1] - t0) / 100.0            # microseconds
print("frame-kernel launch timeline (last step, us): first start 0, start p50=%.1f p90=%.1f last=%.1f | wave duration mean=%.1f p90=%.1f | end p50=%.1f last=%.1f" % (
    np.percentile(st, 50), np.percentile(st, 90), st.max(), (en - st).mean(), np.percentile(en - st, 90), np.percentile(en, 50), en.max()))
print("  starts per 10 us:", np.histogram(st, bins=np.arange(0, en.max() + 10, 10))[0].tolist())
print("  ends   per 10 us:", np.histogram(en, bins=np.arange(0, en.max() + 10, 10))[0].tolist())
ei = env.state_field("env_int").cpu().numpy()
from continiousenvironment_follower_leader_amd import abi
print("traj_len pct", np.percentile(ei[:, abi.EI_TRAJ_LEN], [5, 50, 95]), "green_count pct", np.percentile(ei[:, abi.EI_GREEN_COUNT], [5, 50, 95]), "step_count pct", np.percentile(ei[:, abi.EI_STEP_COUNT], [5, 50, 95]))
# do wave durations persist from one step to the next (would longest-first launch order help)?
durs = []
for k in range(6):
    env.step(acts[k % 16], auto_reset=True)
    env.lib.ftl_debug_wave_times(wt_buf := (C.c_ulonglong * (2 * 8192))())
    a = np.array(list(wt_buf), dtype=np.int64).reshape(8192, 2)[:n // 16]
    durs.append((a[:, 1] - a[:, 0]) / 100.0)
durs = np.array(durs)
print("wave duration correlation between consecutive steps:", [round(float(np.corrcoef(durs[i], durs[i + 1])[0, 1]), 3) for i in range(5)],
      "lag 2:", round(float(np.corrcoef(durs[0], durs[2])[0, 1]), 3), "lag 4:", round(float(np.corrcoef(durs[0], durs[4])[0, 1]), 3),
      "std/mean %.3f" % (durs.std() / durs.mean()))
np.save(os.path.join(os.path.dirname(__file__), "..", "..", "gpurun_out", "wave_durs.npy"), durs)

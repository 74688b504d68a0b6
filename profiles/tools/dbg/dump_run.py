"""Diagnostic: run N config-B envs for T steps with auto-reset and dump per-step output checksums + final state fields to an .npz
(compare two library builds: FTL_LIB=... python dump_run.py out.npz N T)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
out, n, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
env = VecGame(n, device="cuda:0", config=cfg); pool = ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"); env.load_scenarios(pool)
env.reset((torch.arange(n) % pool.n).to(torch.int32))
acts = bench.make_actions(cfg, n, 16, 0, torch.device("cuda:0"))
rec = {}
for t in range(T):
    env.step(acts[t % 16], auto_reset=True)
    torch.cuda.synchronize()
    rec["num%d" % t] = env.obs_num.cpu().numpy(); rec["rew%d" % t] = env.reward.cpu().numpy(); rec["done%d" % t] = env.done.cpu().numpy()
    rec["las%d" % t] = env.lasers.cpu().numpy()
for f in ("rb_pos", "rb_dbl", "rb_int", "env_int", "env_dbl"):
    rec["f_" + f] = env.state_field(f).cpu().numpy()
np.savez_compressed(out, **rec)
print("ok", n, T, env.error_report())

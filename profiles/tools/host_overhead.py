import sys, os, json, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
for n in (1024, 65536):
    env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"))
    env.reset(); acts = bench.make_actions(cfg, n, 16, 0, "cuda:0"); torch.cuda.synchronize()
    for k in range(20): env.step(acts[k % 16], auto_reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200): env.step(acts[k % 16], auto_reset=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(n, "host enqueue per step %.1f us, total per step %.1f us" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
    env.close()

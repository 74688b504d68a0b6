import sys, os, json, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
from continiousenvironment_follower_leader_amd import abi
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
for n, same in ((8192, False), (65536, True)):
    env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"))
    idx = (torch.arange(n, dtype=torch.int64)) % env.pool.n
    if same: idx = idx * 0 + 7          # every env runs the same scenario
    env.reset(idx.to(torch.int32)); acts = bench.make_actions(cfg, n, 16, 0, "cuda:0"); torch.cuda.synchronize()
    ts = []
    for t in range(125):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.step(acts[t % 16], auto_reset=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    print(n, same, [round(x) for x in ts[60:125:3]])
    env.close()

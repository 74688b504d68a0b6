"""Upper bound of what env regrouping could buy: run the bench workload with every 16 consecutive envs (= one wavefront of
the frame kernel) made identical (same scenario, same actions), i.e. zero divergence inside a wavefront, and time the
kernels with HIP events.  Diagnostic only."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
n = 65536
for group in (1, 16):
    env = VecGame(n, device="cuda:0", config=cfg); pool = ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"); env.load_scenarios(pool)
    e = torch.arange(n) // group
    env.reset((e % pool.n).to(torch.int32))
    acts = bench.make_actions(cfg, n // group, 16, 0, torch.device("cuda:0")).repeat_interleave(group, dim=1).contiguous()
    for k in range(150): env.step(acts[k % 16], auto_reset=True)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for k in range(200): env.step(acts[(150 + k) % 16], auto_reset=True)
    ev1.record(); torch.cuda.synchronize()
    print("identical envs per group of %2d: %.4f ms/step" % (group, ev0.elapsed_time(ev1) / 200))
    env.close(); del env

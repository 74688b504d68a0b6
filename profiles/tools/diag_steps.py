import sys, os, json, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
from continiousenvironment_follower_leader_amd import abi
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
n = 65536
env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"))
idx = (torch.arange(n, dtype=torch.int64)) % env.pool.n
env.reset(idx.to(torch.int32)); acts = bench.make_actions(cfg, n, 16, 0, "cuda:0"); torch.cuda.synchronize()
ei = env.state_field("env_int")
for t in range(130):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(acts[t % 16], auto_reset=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
    if t % 5 == 4 or 84 <= t <= 96:
        e = ei.cpu().numpy()
        print("t=%3d %6.0fus  traj %5.1f green %5.1f(max %d) inbox %.2f ontrace %.2f close %.2f done %.3f lfin %.3f epis %d hintgap %.1f steps %.0f" % (
            t, dt, e[:, abi.EI_TRAJ_LEN].mean(), e[:, abi.EI_GREEN_COUNT].mean(), e[:, abi.EI_GREEN_COUNT].max(), e[:, abi.EI_IN_BOX].mean(), e[:, abi.EI_ON_TRACE].mean(),
            e[:, abi.EI_TOO_CLOSE].mean(), e[:, abi.EI_DONE].mean(), e[:, abi.EI_LEADER_FINISHED].mean(), e[:, abi.EI_EPISODES].sum(),
            np.abs(e[:, abi.EI_TRAJ_LEN] - e[:, abi.EI_HINT]).mean(), e[:, abi.EI_STEP_COUNT].mean()))

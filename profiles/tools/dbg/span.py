import json, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from golden_util import GOLDEN, config_for, load_episode
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
from continiousenvironment_follower_leader_amd import abi
import bench
for name in ["B", "E", "F"]:
    n = 16384
    cfg, pool, *_ = bench.build_workload(name, n, 0, 0, torch.device("cuda:0"))
    env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(pool)
    env.reset((torch.arange(n) % pool.n).to(torch.int32))
    acts = bench.make_actions(cfg, n, 16, 0, torch.device("cuda:0"))
    mx = 0; mxw = 0
    for k in range(2500):
        env.step(acts[k % 16], auto_reset=True)
        if k % 5 == 0:
            ei = env.state_field("env_int")
            span = (ei[:, abi.EI_CORR_HI] - ei[:, abi.EI_CORR_LO]).max().item(); mx = max(mx, span)
            sw = env.state_field("snap_win").reshape(n, -1, 4)
            w = (sw[:, :, [1, 3]].amax(dim=(1, 2)) - sw[:, :, [0, 2]].clamp(min=0).amin(dim=(1, 2))).max().item(); mxw = max(mxw, w)
    print(name, "corr_cap", cfg.c.corr_cap, "max live corridor points", mx, "max span over the snapshot windows (rough)", mxw, "errors", env.error_report())

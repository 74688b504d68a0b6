"""Probe: do the frame kernel of one half of the envs and the ray kernel of the other half overlap when the halves are
stepped on two streams?  Two VecGame handles of n/2 envs each on their own stream against one handle of n envs."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
dev = torch.device("cuda:0")

def run(parts, n_total, steps=300, warm=120, stagger=False):
    n = n_total // parts
    envs, acts, streams = [], [], []
    for p in range(parts):
        e = VecGame(n, device=dev, config=cfg); e.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", dev))
        e.reset(((torch.arange(n) + p * n) % e.pool.n).to(torch.int32))
        envs.append(e); acts.append(bench.make_actions(cfg, n, 16, p, dev)); streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    def step(k):
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                envs[p].step(acts[p][k % 16], auto_reset=True)
    for k in range(warm): step(k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps): step(warm + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    for e in envs: e.close()
    return n_total * steps / dt

for parts in (1, 2, 4):
    print("handles/streams: %d -> %.1f M env-steps/s" % (parts, run(parts, 65536) / 1e6), flush=True)

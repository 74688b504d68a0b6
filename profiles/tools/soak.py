"""Soak: many steps of the bench workload (config B pool) and of config E (regimes, generated pool) at full batch size;
checks that no env raises an error flag, observations stay finite and in range, and episode statistics look sane."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from golden_util import GOLDEN, config_for, load_episode
from continiousenvironment_follower_leader_amd import abi
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
import bench

def run(name, cfg, pool_fn, n, steps):
    env = VecGame(n, device="cuda:0", config=cfg)
    pool = pool_fn(cfg)
    env.load_scenarios(pool)
    env.reset((torch.arange(n) % pool.n).to(torch.int32))
    acts = bench.make_actions(cfg, n, 16, 3, torch.device("cuda:0"))
    done_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
    st_hist = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    t = time.time()
    for k in range(steps):
        env.step(acts[k % 16], auto_reset=True)
        d = env.done.bool()
        done_sum += d.sum()
        st_hist += torch.bincount(env.status[:, 1][d].to(torch.int64), minlength=8)[:8]
        if k % 500 == 499:
            ei = env.state_field("env_int")
            err = env.error_report()[0]      # sticky words: errors of episodes that auto-reset wiped count too
            fin = bool(torch.isfinite(env.obs_num).all().item()) and bool(torch.isfinite(env.lasers).all().item())
            print("%s step %5d: error envs %d, finite %s, episodes %.0f, agent-status histogram of finished episodes %s, max traj_len %d" % (
                name, k + 1, err, fin, done_sum.item(), st_hist.cpu().numpy().astype(int).tolist(), int(ei[:, abi.EI_TRAJ_LEN].max().item())), flush=True)
            assert err == 0 and fin
    torch.cuda.synchronize()
    print("%s: %d steps x %d envs in %.1f s; episode metrics %s" % (name, steps, n, time.time() - t, env.episode_metrics().tolist()))
    assert abs(float(env.episode_metrics()[0]) - done_sum.item()) < 0.5
    w, h = cfg.c.width, cfg.c.height
    o = env.obs_num
    assert float(o[:, [0, 1, 5, 6]].min()) > -60 and float(o[:, [0, 5]].max()) < w + 60 and float(o[:, [1, 6]].max()) < h + 60
    env.close()

z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfgB = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
run("B", cfgB, lambda c: ScenarioPool.from_npz(c, GOLDEN + "/pool_B.npz", "cuda:0"), 65536, 3000)
zE, mE = load_episode("E_s3_chase")
cfgE = config_for(mE, scen_route_len=256, rng_seed=11)
run("E", cfgE, lambda c: ScenarioPool.generate(c, np.arange(2048), "cuda:0"), 16384, 4000)

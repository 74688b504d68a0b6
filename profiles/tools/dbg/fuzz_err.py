import sys, os, warnings
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_fuzz import draw_config
from test_gpu_configs import _actions, _vec
from continiousenvironment_follower_leader_amd import make_config, abi
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool
from oracle_batch import OracleBatch, pool_scenarios
seed = int(sys.argv[1])
kw = draw_config(seed)
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); cfg = make_config(route_cap=256, **kw)
print({k: v for k, v in kw.items() if k != "follower_sensors"}); [print("  ", k, v) for k, v in kw["follower_sensors"].items()]
n, steps = 96, 60
pool = ScenarioPool.generate(cfg, np.arange(16) + 50 * seed, "cuda:0")
env = _vec(n, cfg, pool); scen = pool_scenarios(pool); idx = np.arange(n) % pool.n
env.reset(torch.from_numpy(idx.astype(np.int32)))
ora = OracleBatch(cfg, n, env_id_base=100 * seed); ora.reset(scen, idx)
err_seen = np.zeros(n, np.int64)
def dev_err():
    ei = env.state_field("env_int").cpu().numpy(); return ei[:, abi.EI_ERROR].copy(), ei[:, abi.EI_ERROR_STICKY].copy()
for t in range(steps):
    a = _actions(cfg, n, t, "mixed" if t % 3 == 2 else "random", seed=seed)
    env.step(torch.tensor(a, dtype=torch.float64, device="cuda:0")); ora.step(a)
    oe = ora.counters()[2]; err_seen |= oe
    de, ds = dev_err()
    diff = np.nonzero((de != 0) != (oe != 0))[0]
    if len(diff): print("step", t, "current-error mismatch envs", diff[:10], "dev", de[diff[:10]], "ora", oe[diff[:10]], "done", ora.done[diff[:10]])
    d = ora.done.astype(bool)
    if d.any() and t % 12 == 11:
        idx = np.where(d, (idx + n) % pool.n, idx)
        env.reset(torch.from_numpy(idx.astype(np.int32)), mask=torch.from_numpy(d.astype(np.uint8))); ora.reset(scen, idx, mask=d)
        oe = ora.counters()[2]; de, ds = dev_err()
        diff = np.nonzero((de != 0) != (oe != 0))[0]
        if len(diff): print("after masked reset at", t, "mismatch envs", diff[:10], "dev", de[diff[:10]], "ora", oe[diff[:10]])
        err_seen |= oe
de, ds = dev_err()
print("sticky envs", (ds != 0).sum(), "oracle seen", (err_seen != 0).sum(), "envs sticky-but-not-seen", np.nonzero((ds != 0) & (err_seen == 0))[0][:10], "seen-but-not-sticky", np.nonzero((ds == 0) & (err_seen != 0))[0][:10])

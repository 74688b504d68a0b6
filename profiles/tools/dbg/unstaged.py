import sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_configs import _cfg_pool, _vec, _actions, _compare_with_oracle
from oracle_batch import OracleBatch, pool_scenarios
from golden_util import close
n = 300
cfg, pool = _cfg_pool("E_s3_chase", 512, rng_seed=9, env_id_base=9000)
for cap in (None, "8", "32", "128"):
    if cap: os.environ["FTL_DEBUG_CORR_LDS_CAP"] = cap
    elif "FTL_DEBUG_CORR_LDS_CAP" in os.environ: del os.environ["FTL_DEBUG_CORR_LDS_CAP"]
    env = _vec(n, cfg, pool)
    scen = pool_scenarios(pool); idx = (np.arange(n) * 5) % pool.n
    env.reset(torch.from_numpy(idx.astype(np.int32)))
    ora = OracleBatch(cfg, n, env_id_base=9000); ora.reset(scen, idx)
    las = env.lasers.cpu().numpy(); L = cfg.lasers_len
    bad = np.argwhere(~close(las[:, :L], ora.lasers[:, :L]))
    print("cap", cap, "mismatches at reset:", bad[:6].tolist(), [(float(las[e, c]), float(ora.lasers[e, c])) for e, c in bad[:3]])
    if len(bad):
        e = bad[0][0]; s = scen[int(idx[e])]
        print("  env", e, "scenario", int(idx[e]), "follower", s["robot_pos"][1], "dir", s["robot_dir"][1], "rects near:", [r.tolist() for r in np.concatenate([s["static_rects"], s["robot_rect"][[0, 2, 3]]]) if abs(r[0] - s["robot_pos"][1][0]) < 200 and abs(r[1] - s["robot_pos"][1][1]) < 200][:8])
    env.close()

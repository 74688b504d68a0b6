"""A batch of oracle envs stepped through ``ftlo_step_batch`` (OpenMP over envs) with numpy outputs shaped like the
device tensors -- the checker of the GPU parity tests at batch sizes where a Python loop per env would dominate.
TEST INFRASTRUCTURE ONLY (see the header of oracle/ftl_oracle.c)."""
import ctypes as C
import os

import numpy as np

from oracle import OracleEnv, load_oracle


def pool_scenarios(pool):
    """Per-scenario dicts (the keyword arguments of OracleEnv.reset) from a ``ScenarioPool``'s device arrays."""
    t = {k: v.cpu().numpy() for k, v in pool.t.items()}
    return [dict(static_rects=t["static_rects"][i], robot_pos=t["robot_pos"][i], robot_dir=t["robot_dir"][i],
                 robot_rect=t["robot_rect"][i], route=t["route"][i, :t["route_len"][i]],
                 init_traj=t["init_traj"][i, :t["init_traj_len"][i]]) for i in range(pool.n)]


class OracleBatch:
    def __init__(self, cfg, n, env_id_base=None):
        self.cfg, self.n = cfg, n
        self.lib = load_oracle()
        self.envs = [OracleEnv(cfg, env_id=None if env_id_base is None else env_id_base + e) for e in range(n)]
        self.arr = (C.c_void_p * n)(*[o.h for o in self.envs])
        self.L = max(cfg.lasers_len, 1)
        self.obs_num = np.zeros((n, 10), np.float32)
        self.lasers = np.zeros((n, self.L), np.float32)
        self.target = np.zeros((n, 2))
        self.reward = np.zeros(n)
        self.done = np.zeros(n, np.uint8)
        self.status = np.zeros((n, 3), np.uint8)
        self.threads = min(len(os.sched_getaffinity(0)), 16)

    def reset(self, scen, idx, mask=None):
        """env e <- scen[idx[e]] for every e with mask[e] (all when mask is None)."""
        for e, o in enumerate(self.envs):
            if mask is not None and not mask[e]:
                continue
            ob = o.reset(**scen[int(idx[e])])
            self.obs_num[e] = ob["num"]
            self.lasers[e, :self.cfg.lasers_len] = o.lasers[:self.cfg.lasers_len]
            self.target[e] = ob["target"]
            self.reward[e] = 0.0; self.done[e] = 0; self.status[e] = 0

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float64)
        p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        self.lib.ftlo_step_batch(self.arr, self.n, p(a), p(self.obs_num), p(self.lasers), p(self.target), p(self.reward),
                                 p(self.done), p(self.status), self.threads)

    def counters(self):
        """(step_count[n], overall_reward[n], error[n]) of every env."""
        sc = np.zeros(self.n, np.int64); ret = np.zeros(self.n); err = np.zeros(self.n, np.int64)
        cnt = np.zeros(32, np.int64); acc = np.zeros(2)
        for e, o in enumerate(self.envs):
            self.lib.ftlo_get_counters(o.h, cnt.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p))
            sc[e] = cnt[0]; ret[e] = acc[1]; err[e] = cnt[14]
        return sc, ret, err

    def robot_ints(self):
        R = self.cfg.n_robots
        out = np.zeros((self.n, R, 6), np.int32)
        pos = np.zeros((R, 2), np.float32); dbl = np.zeros((R, 5))
        for e, o in enumerate(self.envs):
            self.lib.ftlo_get_robots(o.h, pos.ctypes.data_as(C.c_void_p), dbl.ctypes.data_as(C.c_void_p), out[e].ctypes.data_as(C.c_void_p))
        return out

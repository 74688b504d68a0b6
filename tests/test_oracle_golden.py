"""The CPU oracle (oracle/ftl_oracle.c) against the golden episodes produced by the UNMODIFIED reference
(tests/golden/gen/make_golden.py).  This is what pins the oracle: every observable of every step --
numerical_features, ray-sensor rows, leader_target_point, reward, done, info codes -- plus the internal
state the generator dumped (robot poses / controller state / integer hitboxes, counters, tracker history
and corridor) must agree.  Integer / flag state bit-exactly, floats to the north_star tolerance (they are in
fact bit-identical in this container; the test keeps the contract tolerance so that it stays meaningful on
a host with a different libm)."""
import numpy as np
import pytest

from golden_util import close, config_for, episode_names, load_episode, scenario_arrays
from oracle import OracleEnv


@pytest.mark.parametrize("name", episode_names())
def test_oracle_matches_reference_episode(name):
    z, meta = load_episode(name)
    cfg = config_for(meta, scen_route_len=len(z["scen:route"]))
    env = OracleEnv(cfg)
    obs = env.reset(**scenario_arrays(z))
    lnames = meta["laser_names"]

    def check_obs(tag, t, obs):
        ref = z[tag + ":num"] if t is None else z[tag + ":num"][t]
        assert close(obs["num"], ref).all(), (name, t, "num", obs["num"] - ref)
        for ln in lnames:
            ref = z[tag + ":laser:" + ln] if t is None else z[tag + ":laser:" + ln][t]
            assert obs[ln].shape == ref.shape
            assert close(obs[ln], ref).all(), (name, t, ln, np.abs(obs[ln] - ref).max())
        for a in cfg.aux:                      # LaserSensor / LeaderTrackDetector_vector / _radar (float32 arrays)
            ref = z[tag + ":aux:" + a.name] if t is None else z[tag + ":aux:" + a.name][t]
            assert obs[a.name].shape == ref.shape, (name, a.name, obs[a.name].shape, ref.shape)
            assert close(obs[a.name], ref).all(), (name, t, a.name, np.abs(obs[a.name] - ref).max(), np.argwhere(~close(obs[a.name], ref))[:4])

    check_obs("reset", None, obs)
    assert np.array_equal(obs["target"], z["reset:target"])
    for t in range(len(z["actions"])):
        obs, rew, done, st = env.step(z["actions"][t])
        check_obs("obs", t, obs)
        assert abs(rew - z["reward"][t]) <= 1e-5, (name, t, rew, z["reward"][t])
        assert done == bool(z["done"][t]), (name, t)
        assert tuple(st) == tuple(z["info"][t]), (name, t, st, z["info"][t])
        assert np.array_equal(obs["target"], z["obs:target"][t]), (name, t)
        d = env.debug()
        assert np.array_equal(d["counters"][:11], z["dbg:counters"][t]), (name, t, d["counters"][:11], z["dbg:counters"][t])
        assert np.array_equal(d["robot_i32"], z["dbg:robot_i32"][t]), (name, t, "hitboxes")
        assert close(d["robot_pos"], z["dbg:robot_pos"][t]).all(), (name, t)
        assert np.allclose(d["robot_f64"], z["dbg:robot_f64"][t], rtol=0, atol=1e-9), (name, t)
        assert abs(d["acc"] - z["dbg:acc"][t]).max() <= 1e-9
        if "dbg:trk" in z:
            tr = z["dbg:trk"][t]
            assert tuple(tr) == tuple(d["counters"][11:14]), (name, t, tr, d["counters"][11:14])
            assert np.allclose(d["hist"], z["dbg:hist"][t][:int(tr[1])], rtol=0, atol=1e-9)
            assert np.allclose(d["corr"], z["dbg:corr"][t][:int(tr[2])], rtol=0, atol=1e-9)
            assert np.array_equal(d["hist_isf64"], z["dbg:hist_isf64"][t][:int(tr[1])])
        if "dbg:dyn_index" in z:
            nb = z["dbg:dyn_index"].shape[1]
            assert np.array_equal(d["counters"][15:15 + nb], z["dbg:dyn_index"][t])
        assert d["counters"][14] == 0, "oracle raised an error flag"


def test_golden_covers_terminal_modes():
    """The fixture set must exercise every way an episode ends (ENV:960-964, 1077-1107, 1129-1134)."""
    seen = set()
    for name in episode_names():
        z, _ = load_episode(name)
        for row, d in zip(z["info"], z["done"]):
            if d:
                seen.add(tuple(int(v) for v in row))
    assert (1, 1, 0) in seen      # fail / crash
    assert (2, 4, 2) in seen      # success
    assert (3, 0, 0) in seen      # finished_by_time
    assert (1, 2, 0) in seen      # low_reward

"""GPU tests of the episode-metrics vector (ENV:941-944) and of the sticky per-env error words:
both must survive the in-kernel auto-reset that wipes the episode counters."""
import json

import numpy as np
import pytest
import torch

from continiousenvironment_follower_leader_amd import _lib, abi, shard
from golden_util import GOLDEN, config_for
from oracle_batch import OracleBatch, pool_scenarios

pytestmark = pytest.mark.gpu


def _pool_cfg(**over):
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    kw = dict(meta["kwargs"])
    kw.update(over)
    return config_for(dict(kwargs=kw, post=None), scen_route_len=int(z["route_len"].max()))


def _vec(n, cfg, limit=None):
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
    env = VecGame(n, device="cuda:0", config=cfg)
    env.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0", limit=limit))
    return env


def _actions(cfg, n, t):
    rng = np.random.default_rng(900 + t)
    ms, mr = cfg.c.follower.max_speed, cfg.c.follower.max_rotation_speed
    return np.stack([rng.uniform(0.5, 1.0, n) * ms, np.clip(rng.normal(0, 0.2 * mr, n), -mr, mr)], 1)


@pytest.mark.parametrize("auto_reset", [True, False])
def test_metrics_vector_matches_oracle_episodes(auto_reset):
    """[episodes, sum return, sum frames, success, crash, low_reward, too_far, timeout] accumulated on the device against
    overall_reward / step_count / info of the oracle's episodes at the step they end (ENV:941-944).  With auto-reset the
    oracle env is reset to the scenario the kernel picks, (scen + n_envs) mod P; without it an episode counts once, at the
    step that raises `done`, although the reference keeps simulating afterwards (ENV:935-936)."""
    n, P, steps = 192, 256, 40
    cfg = _pool_cfg(max_steps=180, warm_start=20, early_stopping={"max_distance_coef": 1.1, "low_reward": -30})
    env = _vec(n, cfg, limit=P)
    scen = pool_scenarios(env.pool)
    idx = np.arange(n) % P
    env.reset(torch.from_numpy(idx.astype(np.int32)))
    ora = OracleBatch(cfg, n)
    ora.reset(scen, idx)
    want = torch.zeros(8, dtype=torch.float64)
    finished = np.zeros(n, bool)
    for t in range(steps):
        a = _actions(cfg, n, t)
        env.step(torch.tensor(a, dtype=torch.float64, device="cuda:0"), auto_reset=auto_reset)
        ora.step(a)
        assert np.array_equal(env.done.cpu().numpy(), ora.done) and np.array_equal(env.status.cpu().numpy(), ora.status), t
        sc, ret, _ = ora.counters()
        ends = ora.done.astype(bool) & (~finished if not auto_reset else True)
        want += shard.episode_metrics_from_outputs(torch.from_numpy(ends), torch.from_numpy(ora.status), torch.from_numpy(ret),
                                                   torch.from_numpy(sc.astype(np.float64)))
        if auto_reset:
            idx = np.where(ora.done.astype(bool), (idx + n) % P, idx)
            ora.reset(scen, idx, mask=ora.done.astype(bool))
        else:
            finished |= ora.done.astype(bool)
    got = env.episode_metrics().cpu()
    assert want[abi.M_EPISODES] >= n / 2, "the sample must finish episodes"
    assert want[abi.M_CRASH] > 0 and want[abi.M_TIMEOUT] + want[abi.M_LOW_REWARD] + want[abi.M_TOO_FAR] > 0
    assert torch.equal(got[[0, 2, 3, 4, 5, 6, 7]], want[[0, 2, 3, 4, 5, 6, 7]]), (got, want)      # counts and frames: exact
    assert abs(float(got[1] - want[1])) <= 1e-9 * max(1.0, abs(float(want[1]))), (got[1], want[1])
    # the vector is what the kernel wrote per env: it equals the sum of the "ep_stats" records, and a second read is identical
    assert torch.equal(env.state_field("ep_stats").sum(0).cpu()[[0, 2, 3, 4, 5, 6, 7]], got[[0, 2, 3, 4, 5, 6, 7]])
    assert torch.equal(env.episode_metrics().cpu(), got)
    assert env.error_report() == (0, 0)
    # clear: the next read starts from zero
    env.episode_metrics(clear=True)
    assert float(env.episode_metrics().abs().sum()) == 0.0
    env.close()


def test_metrics_sum_is_order_independent_and_reproducible():
    """The fixed-order reduction gives the same bits whatever the slot -> env permutation did (regrouping on / off)."""
    import os
    n = 8192 + 11
    cfg = _pool_cfg(max_steps=150, warm_start=10)
    try:
        os.environ["FTL_NO_REGROUP"] = "0"          # forced on (the default switches it on beyond one round of wavefronts only)
        a = _vec(n, cfg)
        os.environ["FTL_NO_REGROUP"] = "1"
        b = _vec(n, cfg)
    finally:
        del os.environ["FTL_NO_REGROUP"]
    idx = torch.arange(n, dtype=torch.int32) % a.pool.n
    a.reset(idx); b.reset(idx)
    for t in range(30):
        act = torch.tensor(_actions(cfg, n, t), dtype=torch.float64, device="cuda:0")
        a.step(act, auto_reset=True); b.step(act, auto_reset=True)
    ma, mb = a.episode_metrics().cpu(), b.episode_metrics().cpu()
    assert float(ma[0]) >= n and torch.equal(ma, mb), (ma, mb)
    assert int(a.state_field("env_int")[:, abi.EI_EPISODES].sum()) == int(ma[0])
    a.close(); b.close()


def test_error_bits_survive_auto_reset_and_are_surfaced():
    """A trajectory capacity that is too small overflows in every episode; the per-episode word EI_ERROR is wiped by the
    auto-reset of the same launch, the sticky word and the report are not."""
    from continiousenvironment_follower_leader_amd.vec_game import error_for_bits
    n = 128
    base = _pool_cfg()
    cfg = _pool_cfg(max_steps=400, warm_start=10, traj_cap=((base.c.init_traj_cap + 31) // 32) * 32, init_traj_cap=base.c.init_traj_cap)
    env = _vec(n, cfg)
    env.reset(torch.arange(n, dtype=torch.int32))
    assert env.error_report() == (0, 0)
    for t in range(90):          # episodes last 40 steps = 80 appended points: more than the 8..16 spare slots of any scenario
        env.step(torch.tensor(_actions(cfg, n, t), dtype=torch.float64, device="cuda:0"), auto_reset=True)
    ei = env.state_field("env_int")
    assert int(ei[:, abi.EI_EPISODES].sum()) > n, "every env should have been auto-reset at least once"
    live = ei[:, abi.EI_ERROR].cpu().numpy()
    sticky = ei[:, abi.EI_ERROR_STICKY].cpu().numpy()
    n_sticky = int((sticky != 0).sum())
    assert n_sticky > n // 2, "most envs overflowed at least once (the others crashed before their trajectory filled up)"
    assert (((live & abi.FTL_ERR_TRAJ_OVERFLOW) == 0) & ((sticky & abi.FTL_ERR_TRAJ_OVERFLOW) != 0)).any(), \
        "some env that overflowed is in a fresh episode whose own word is clean again"
    assert ((live & ~sticky) == 0).all(), "every live bit is in the sticky word"
    cnt, bits = env.error_report()
    assert cnt == n_sticky and bits == abi.FTL_ERR_TRAJ_OVERFLOW
    with pytest.raises(_lib.FtlError, match="traj_cap"):
        env.step(torch.tensor(_actions(cfg, n, 99), dtype=torch.float64, device="cuda:0"), auto_reset=True, check_errors=True)
    # an explicit (masked) reset keeps the sticky word too
    env.reset(torch.arange(n, dtype=torch.int32), mask=torch.ones(n, dtype=torch.uint8))
    assert env.error_report()[0] >= n_sticky
    env.episode_metrics(clear=True)
    assert env.error_report() == (0, 0)
    # the reference's exception types for the conditions it raises on
    assert isinstance(error_for_bits(abi.FTL_ERR_EMPTY_CORRIDOR), UnboundLocalError)
    assert isinstance(error_for_bits(abi.FTL_ERR_TRACKER_SEED), IndexError)
    env.close()


def test_game_facade_raises_what_the_reference_raises():
    """A tracker whose history cannot hold the seeded points leaves the corridor empty: the reference's ray sensor raises
    UnboundLocalError at the first scan (SEN:893/962); the single-env facade raises the same type from reset()."""
    from continiousenvironment_follower_leader_amd.game import Game
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    kw = dict(meta["kwargs"])
    sensors = {k: dict(v) for k, v in kw["follower_sensors"].items()}
    sensors["LeaderPositionsTracker_v2"]["corridor_length"] = 1        # every seeded point is trimmed away: corridor of <= 1 points
    kw["follower_sensors"] = sensors
    g = Game(scenarios=GOLDEN + "/pool_B.npz", route_cap=int(z["route_len"].max()), **kw)
    g.seed(0)
    with pytest.raises((UnboundLocalError, IndexError)):
        g.reset()
    g.close()


def test_stop_and_go_soak_keeps_every_capacity():
    """Whole episodes (max_steps 5000 frames = 500 steps, with auto-reset into second episodes) under a stop-and-go follower on configs B
    and D: the tracker ring is sized from the leader's point spacing with 2.5x head-room (config.py), the trajectory slot from max_steps --
    neither may overflow (FTL_ERR_CORR_OVERFLOW / FTL_ERR_TRAJ_OVERFLOW would make the results diverge from the reference silently)."""
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
    from golden_util import load_episode
    for ep, n in (("B_s1_chase", 2048), ("D_s2_chase", 512)):
        _, meta = load_episode(ep)
        cfg = config_for(meta, scen_route_len=256)
        env = VecGame(n, device="cuda:0", config=cfg)
        env.load_scenarios(ScenarioPool.generate(cfg, np.arange(300), "cuda:0"))
        env.reset()
        ms, mr = cfg.c.follower.max_speed, cfg.c.follower.max_rotation_speed
        gen = torch.Generator(device="cpu"); gen.manual_seed(3)
        phase = torch.arange(n) % 7
        for t in range(560):
            v = (0.6 + 0.4 * torch.rand(n, generator=gen, dtype=torch.float64)) * ms
            v[((t // 12) + phase) % 3 == 0] = 0.0                      # a third of the envs stand still for 12 steps at a time
            w = torch.clamp(torch.randn(n, generator=gen, dtype=torch.float64) * 0.15 * mr, -mr, mr)
            env.step(torch.stack([v, w], 1).to("cuda:0"), auto_reset=True)
        assert env.error_report() == (0, 0), (ep, env.error_report())
        m = env.episode_metrics().tolist()
        assert m[0] >= n                                               # every slot finished at least one episode on average
        env.close()

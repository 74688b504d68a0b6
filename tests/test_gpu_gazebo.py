"""Row f4 on the GPU: the batched follower-relative tracker / ray sensors (ftl_gz_kernel behind include/ftl_gazebo.h) against the
vectors of the UNMODIFIED reference classes (gazebo_tracker.py) and, on a batch of independent robots, against the oracle."""
import numpy as np
import pytest
import torch

from golden_util import close
from test_gazebo_oracle import NAMES, load

pytestmark = pytest.mark.gpu


def _dev(a, dt=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")


@pytest.mark.parametrize("name", NAMES)
def test_gazebo_hip_matches_reference(name):
    from continiousenvironment_follower_leader_amd.gazebo import GazeboTrackerBatch
    z, meta = load(name)
    n = 3                      # the same robot three times: multi-workgroup launches, env indexing
    g = GazeboTrackerBatch(n, lasers=meta["lasers"], max_pts=meta["max_pts"])
    for t in range(meta["steps"]):
        tile = lambda a: np.tile(np.asarray(a)[None], (n,) + (1,) * np.asarray(a).ndim)      # noqa: E731
        g.step(_dev(tile(z["leader"][t])), _dev(np.full(n, z["yaw"][t])), _dev(tile(z["delta"][t])), _dev(tile(z["pts1"][t])),
               _dev(tile(z["pts2"][t])), _dev(np.full(n, z["n_pts"][t]), torch.int32))
        for e in (0, n - 1):
            counter, hist, corr, err = g.tracker_state(e)
            assert err == 0 and counter == int(z["counter"][t]), (name, t, counter, z["counter"][t])
            assert len(hist) == int(z["n_hist"][t]) and len(corr) == int(z["n_corr"][t]), (name, t)
            assert np.allclose(hist, z["hist"][t][:len(hist)], rtol=0, atol=1e-12), (name, t, "history")
            assert np.allclose(corr.reshape(-1, 4), z["corr"][t][:len(corr)], rtol=0, atol=1e-9), (name, t, "corridor")
        for k in range(len(meta["lasers"])):
            got = g.laser_view(k).cpu().numpy()
            ref = z["laser%d" % k][t]
            assert got.shape[1:] == ref.shape
            for e in range(n):
                assert close(got[e], ref).all(), (name, t, k, e, np.abs(got[e] - ref).max())
    g.close()


def test_gazebo_batch_matches_oracle_with_masked_reset():
    """64 independent robots on their own random motion; a masked reset in the middle restarts some of them (tracker.reset() +
    laser.reset()): history, corridor and both sensors' rows against one oracle per robot."""
    from continiousenvironment_follower_leader_amd.gazebo import ARCTIC_ENV_LASERS, GazeboTrackerBatch, make_gz_config
    from oracle import OracleGazebo
    n, steps, mp = 64, 70, 32
    g = GazeboTrackerBatch(n, lasers=ARCTIC_ENV_LASERS, max_pts=mp)
    oras = [OracleGazebo(make_gz_config(ARCTIC_ENV_LASERS, mp)) for _ in range(n)]
    for o in oras:
        o.reset()
    rng = np.random.default_rng(3)
    lead = np.stack([rng.uniform(5, 9, n), rng.uniform(-3, 3, n)], 1)
    yaw = rng.uniform(-1, 1, n)
    for t in range(steps):
        delta = np.stack([rng.uniform(0.0, 0.45, n), rng.normal(0, 0.05, n)], 1)
        lead = lead + np.stack([rng.uniform(0.0, 0.5, n), rng.normal(0, 0.15, n)], 1) - delta
        yaw = yaw + rng.normal(0, 0.05, n)
        n_pts = rng.integers(0, mp + 1, n).astype(np.int32)
        p1 = rng.uniform(-14, 14, (n, mp, 2)); p1[:, 1::2] = p1[:, 0::2] + rng.uniform(-0.4, 0.4, (n, mp // 2, 2))
        p2 = p1 + rng.uniform(0.3, 1.5, p1.shape)
        if t == 33:
            mask = (np.arange(n) % 3 == 0)
            g.reset(mask=torch.from_numpy(mask.astype(np.uint8)))
            for e in np.nonzero(mask)[0]:
                oras[e].reset()
                lead[e] = [7.0, 0.5]
        las = g.step(_dev(lead), _dev(yaw), _dev(delta), _dev(p1), _dev(p2), _dev(n_pts, torch.int32)).cpu().numpy()
        for e, o in enumerate(oras):
            ref = o.step(lead[e], yaw[e], delta[e], p1[e, :n_pts[e]], p2[e, :n_pts[e]])
            assert close(las[e, :len(ref)], ref).all(), (t, e, np.abs(las[e, :len(ref)] - ref).max())
            if e % 7 == 0:
                st = o.state()
                counter, hist, corr, err = g.tracker_state(e)
                assert counter == st["counter"] and err == st["error"] and len(hist) == len(st["hist"]) and len(corr) == len(st["corr"]), (t, e)
                # (the seed points go through sin / cos: the device's differ from glibc's in the last bit)
                assert np.allclose(hist, st["hist"], rtol=0, atol=1e-12) and np.allclose(corr.reshape(-1, 4), st["corr"], rtol=0, atol=1e-11), (t, e)
    g.close()


@pytest.mark.parametrize("seed", range(10))
def test_gazebo_random_sensor_sets_match_oracle(seed):
    """Randomised sensor sets (ray count, length, history 1-12, every react_to_* combination, pad_sectors) on 40 robots x 45 steps of
    random motion with obstacle points: the candidate-arc pruning of ftl_gz_kernel must never drop a hit the oracle finds."""
    from continiousenvironment_follower_leader_amd.gazebo import GazeboTrackerBatch, make_gz_config
    from oracle import OracleGazebo
    rng = np.random.default_rng(500 + seed)
    lasers = []
    for _ in range(int(rng.integers(1, 3))):
        react = dict(react_to_green_zone=bool(rng.integers(2)), react_to_safe_corridor=bool(rng.integers(2)), react_to_obstacles=bool(rng.integers(2)))
        if not any(react.values()):
            react["react_to_obstacles"] = True
        lasers.append(dict(lasers_count=int(rng.choice([12, 20, 24, 36])), laser_length=float(rng.choice([4, 8, 10, 15, 20])),
                           max_prev_obs=int(rng.integers(1, 13)), pad_sectors=bool(rng.integers(3) == 0), **react))
    lasers = tuple(lasers)
    n, steps, mp = 40, 45, 24
    g = GazeboTrackerBatch(n, lasers=lasers, max_pts=mp)
    oras = [OracleGazebo(make_gz_config(lasers, mp)) for _ in range(n)]
    for o in oras:
        o.reset()
    lead = np.stack([rng.uniform(4, 9, n), rng.uniform(-3, 3, n)], 1)
    yaw = rng.uniform(-3, 3, n)
    for t in range(steps):
        delta = np.stack([rng.uniform(0.0, 0.45, n), rng.normal(0, 0.08, n)], 1)
        lead = lead + np.stack([rng.uniform(0.0, 0.5, n), rng.normal(0, 0.2, n)], 1) - delta
        yaw = yaw + rng.normal(0, 0.1, n)
        n_pts = rng.integers(0, mp + 1, n).astype(np.int32)
        p1 = rng.uniform(-12, 12, (n, mp, 2)); p1[:, 1::2] = p1[:, 0::2] + rng.uniform(-0.4, 0.4, (n, mp // 2, 2))
        p1[::5, :4] *= 0.05                       # some obstacle segments right next to the robot (every ray is a candidate)
        p2 = p1 + rng.uniform(0.3, 1.5, p1.shape)
        las = g.step(_dev(lead), _dev(yaw), _dev(delta), _dev(p1), _dev(p2), _dev(n_pts, torch.int32)).cpu().numpy()
        for e, o in enumerate(oras):
            ref = o.step(lead[e], yaw[e], delta[e], p1[e, :n_pts[e]], p2[e, :n_pts[e]])
            assert close(las[e, :len(ref)], ref).all(), (seed, lasers, t, e, np.abs(las[e, :len(ref)] - ref).max())
    g.close()

"""Host logic: the constructor mirror (config.make_config) against the reference's documented defaults / unit
conversions (ENV:45-105, 283-357; SURVEY.md 8(c) sanity anchors) and its error behaviour."""
import math

import pytest

from continiousenvironment_follower_leader_amd import make_config
from continiousenvironment_follower_leader_amd.game import Game, _spaces
from golden_util import config_for, load_episode


def test_defaults_match_reference_units():
    c = make_config().c
    assert (c.width, c.height, c.frames_per_step, c.max_steps, c.warm_start) == (1500, 1000, 10, 5000, 500)
    assert c.n_static == 37 and c.n_bears == 3 and c.trajectory_saving_period == 5
    assert (c.min_distance, c.max_distance, c.max_dev, c.leader_pos_epsilon) == (50, 200, 50, 25)
    f, l, b = c.follower, c.leader, c.bear
    assert f.max_speed == 0.25 and f.min_speed == 0 and math.isclose(f.max_rotation_speed, 0.57296)
    assert f.max_speed_change == 0.0025 and f.max_rotation_speed_change == 0.2
    assert (f.img_w, f.img_h) == (17, 25) and (l.img_w, l.img_h) == (19, 26) and (b.img_w, b.img_h) == (25, 25)
    assert math.isclose(b.max_speed, 0.275) and b.max_speed_change == 0.25
    assert (c.reward_in_box, c.reward_in_dev, c.reward_on_track) == (1.0, 0.5, 0.1)
    assert (c.not_on_track_penalty, c.crash_penalty, c.too_close_penalty, c.leader_movement_reward) == (-1, -10, -5, 0)


def test_gazebo_preset_units():
    # TestGameManual_gazebo numbers (ENV:2015-2040) without the regimes
    c = make_config(pixels_to_meter=10, min_distance=8, max_distance=15, max_dev=1, follower_size=(1, 1), leader_size=(4, 2),
                    bear_size=(1.5, 1.5), follower_max_speed=2, leader_max_speed=1, negative_speed=True, bear_number=2,
                    follower_max_rotation_speed=28.65, leader_max_rotation_speed=28.65, follower_acceleration=1,
                    leader_acceleration=1, obstacle_number=20, frames_per_step=5, max_steps=30000, warm_start=0,
                    early_stopping={"max_distance_coef": 4, "low_reward": -300}).c
    assert c.min_distance == 80 and c.max_distance == 150 and c.follower.min_speed == -0.2 and c.follower.max_speed == 0.2
    assert c.leader.max_speed == 0.1 and (c.leader.img_w, c.leader.img_h) == (40, 20) and c.n_static == 22
    assert c.has_low_reward == 1 and c.low_reward == -300 and c.has_max_distance_coef == 1 and c.max_distance_coef == 4


def test_sensor_registry_and_dict_order():
    z, meta = load_episode("B_s1_chase")
    cfg = config_for(meta)
    assert cfg.tracker_name == "LeaderPositionsTracker_v2" and cfg.c.tracker_saving_period == 8
    assert [(l.name, l.count, l.length, l.history, l.after_tracker) for l in cfg.lasers] == [
        ("LeaderCorridor_lasers_all", 12, 100.0, 5, True), ("LeaderCorridor_lasers_obstacles", 24, 150.0, 5, True)]
    assert [l.react_obstacles for l in cfg.lasers] == [1, 1] and cfg.lasers[0].react_green and not cfg.lasers[1].react_corridor
    assert cfg.lasers[0].angle_offset == -45
    # tracker LAST in the dict (as in server/config/3c1bc/params.json): lasers are scanned before its 2nd scan
    sens = dict(meta["kwargs"]["follower_sensors"])
    trk = sens.pop("LeaderPositionsTracker_v2")
    sens["LeaderPositionsTracker_v2"] = trk
    cfg2 = make_config(bear_number=1, follower_sensors=sens)
    assert [l.after_tracker for l in cfg2.lasers] == [False, False]


@pytest.mark.parametrize("kw,exc", [
    (dict(path_finding_algorythm="rrt"), ValueError),                               # ENV:423-425
    (dict(add_bear=True, bear_number=0), ValueError),                               # ENV:426-427
    (dict(multiple_end_points=True, path_finding_algorythm="astar"), NotImplementedError),   # ENV:239-243
    (dict(follower_sensors={"x": {"sensor_class": "LeaderCorridor_Prev_lasers_v2", "lasers_count": 13, "max_prev_obs": 5}}), ValueError),  # SEN:761-762
    (dict(follower_sensors={"mystery": {}}), ValueError),                            # CLS:249
    (dict(follower_sensors={"LeaderTrackDetector_radar": {}}), ValueError),          # CLS:240-243: no tracker registered
    (dict(follower_sensors={"c": {"sensor_class": "LeaderCorridor_lasers_compas", "max_prev_obs": 5}}), ValueError),   # SEN:1148-1151: flags
    (dict(follower_sensors={"c": {"sensor_class": "LeaderCorridor_Prev_lasers_v3", "max_prev_obs": 5}}), ValueError),  # SEN:993
    (dict(manual_control=True), NotImplementedError),
    (dict(bear_number=7), NotImplementedError),
])
def test_constructor_errors(kw, exc):
    with pytest.raises(exc):
        make_config(**kw)


def test_shipped_training_config_is_accepted():
    """server/config/3c1bc/params.json: random frames per step, ten snapshots, the v2 tracker under the key
    'LeaderPositionsTracker' registered last (so the ray sensors scan before its second scan, CLS:255-288)."""
    z, meta = load_episode("F_s7_chase")
    with pytest.warns(UserWarning):                      # ENV:399-401 warns that both frame settings are given
        cfg = config_for(meta)
    c = cfg.c
    assert (c.rand_fps_lo, c.rand_fps_hi) == (30, 70) and c.frames_per_step == 10
    assert cfg.tracker_name == "LeaderPositionsTracker" and [l.after_tracker for l in cfg.lasers] == [False, False]
    assert [l.history for l in cfg.lasers] == [10, 10] and c.n_speed_regime == 9
    assert [c.speed_key[i] for i in range(9)] == [0, 1000, 1500, 200, 2300, 2500, 3000, 4000, 5000]      # the file's key order
    assert c.traj_cap >= (c.max_steps + 2 * 70) // 5                                                        # room for the longest step
    with pytest.raises(ValueError):
        make_config(random_frames_per_step=[70, 30])
    with pytest.raises(NotImplementedError):
        make_config(random_frames_per_step=[30, 70], frames_per_step=None)


def test_corridor_ring_capacity_follows_the_point_spacing():
    """corr_cap (the tracker rings, and through them the LDS footprint of the ray kernel): 2.5x the corridor's point count at full
    leader speed -- a point every saving_period / 2 steps of frames_per_step frames --, the flat 4x over the seeded spacing when
    regimes can slow the leader down, always a power of two, never below the corridor's point count."""
    import warnings
    _, mb = load_episode("B_s5_random")
    _, me = load_episode("E_s3_chase")
    b = config_for(mb).c
    assert b.corr_cap == 64 and b.frames_per_step == 10
    per_point = b.tracker_saving_period / 2 * b.frames_per_step * b.leader.max_speed        # px between saved points at full speed
    assert b.corr_cap >= 2.5 * b.corridor_length / per_point > b.corr_cap / 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert config_for(me).c.corr_cap == 256                                             # speed + acceleration regimes
        kw = dict(mb["kwargs"])
        assert make_config(**dict(kw, frames_per_step=5)).c.corr_cap == 128                 # points twice as dense
        assert make_config(**dict(kw, frames_per_step=3)).c.corr_cap == 256
        assert make_config(**dict(kw, corr_cap=100)).c.corr_cap == 128                      # an explicit value is rounded up to a power of two


def test_lidar_return_all_points_block():
    """LaserSensor(return_all_points=True) (SEN:112-113, 131-134): the batched block is [K][K rows][zeros], sized for every marching point."""
    cfg = make_config(follower_sensors={"LaserSensor": {"return_all_points": True, "available_angle": 90, "angle_step": 30, "points_number": 6}})
    a = cfg.aux[0]
    assert a.params["return_all_points"] == 1 and a.params["n_angles"] == 5 and a.shape == (1 + 5 * 6 * 2,) and cfg.lasers_len == 61
    cfg = make_config(follower_sensors={"LaserSensor": {"return_all_points": True, "return_only_distances": True, "points_number": 4}})
    assert cfg.aux[0].shape == (1 + 37 * 4,)


def test_unknown_kwargs_are_swallowed_like_the_reference():
    make_config(some_future_flag=1)        # ENV:104 **kwargs


def test_spaces():
    act, obs = _spaces(make_config())
    assert act.shape == (2,) and act.low[0] == 0 and math.isclose(float(act.high[1]), 0.57296, rel_tol=1e-6)   # Env_demo.ipynb cell 8
    assert obs.shape == (10,) and obs.high[0] == 1500 and obs.high[3] == 360
    act, _ = _spaces(make_config(discrete_action_space=True))
    assert act.n == 5
    act, _ = _spaces(make_config(constant_follower_speed=True))
    assert act.shape == (1,)
    act, _ = _spaces(make_config(negative_speed=True))
    assert act.low[0] == -0.25


def test_game_facade_fails_loudly_without_a_gpu():
    """No CPU fallback: on a box without a ROCm device reset() raises instead of computing anything on the host."""
    import torch
    from continiousenvironment_follower_leader_amd._lib import FtlError
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    g = Game(bear_number=1)
    with pytest.raises(FtlError):
        g.reset()


def test_registry_and_params_json(tmp_path):
    import json
    from continiousenvironment_follower_leader_amd.game import kwargs_from_params_json, make
    z, meta = load_episode("F_s7_chase")
    # the layout of the reference's shipped params.json (server/config/3c1bc): env_config.base_env_config + name + wrappers
    doc = {"env": "continuous-grid", "env_config": {"base_env_config": meta["kwargs"], "name": "Test-Cont-Env-Auto-v0",
                                                     "wrappers": ["ContinuousObserveModifier_sensorPrev", "SkipBadSeeds"]}}
    path = tmp_path / "params.json"
    path.write_text(json.dumps(doc))
    kw, env_id, wrappers = kwargs_from_params_json(str(path))
    assert env_id == "Test-Cont-Env-Auto-v0" and wrappers[0] == "ContinuousObserveModifier_sensorPrev"
    assert list(kw["leader_speed_regime"].keys()) == ["0", "1000", "1500", "200", "2300", "2500", "3000", "4000", "5000"]
    with pytest.warns(UserWarning):
        g = make(env_id, **kw)                                  # constructing needs no GPU; reset() would
    assert g.cfg.c.rand_fps_hi == 70 and g.cfg.tracker_name == "LeaderPositionsTracker"
    assert make("Test-Game-Neat-v0").action_space.n == 5
    with pytest.raises(NotImplementedError):
        make("Test-Cont-Env-Manual-v0")
    with pytest.raises(KeyError):
        make("no-such-env")

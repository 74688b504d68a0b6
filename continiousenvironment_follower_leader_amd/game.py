"""``Game`` -- the single-env, gym-style facade over the batched HIP path: same constructor kwargs, ``seed`` /
``reset`` / ``step`` signatures, observation dict keys / shapes / dtypes, ``info`` strings and spaces as the
reference's ``class Game(gym.Env)`` (follow_the_leader_continuous_env.py:44-105, 429-543, 908-945, 1789-1824).

``reset()`` builds its scenario with the host-side generator (scenario.py / ``ftl_generate_scenarios``): after
``seed(v)`` the first ``reset()`` yields the scenario of the reference's ``game.seed(v); game.reset()`` (rocks, robots,
route -- up to the choice among equal-cost routes).  Differences: later resets without a new ``seed()`` draw
from a fresh stream keyed on (v, reset number) instead of continuing the global Mersenne twister, and a scenario the
reference would start with a broken route (finish point inside an inflated obstacle, ``found_target_point`` False) is
re-drawn.  With ``scenarios=`` (a ``ScenarioPool`` or an .npz captured from the reference) ``seed(v)`` selects pool entry
``v mod P`` instead."""
from collections import OrderedDict

import numpy as np
import torch

from . import abi
from .config import make_config
from .vec_game import ScenarioPool, VecGame


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency of this package)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
            low, high = np.full(shape, low, dtype=dtype), np.full(shape, high, dtype=dtype)
        self.low, self.high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
        self.shape, self.dtype = self.low.shape, np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete:
    def __init__(self, n):
        self.n, self.shape, self.dtype = n, (), np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n


def _spaces(cfg):
    c = cfg.c
    if cfg.discrete_action_space:                                   # ENV:360-367
        act = Discrete(5)
    elif cfg.constant_follower_speed:                               # ENV:368-372
        act = Box(low=-c.follower.max_rotation_speed, high=c.follower.max_rotation_speed, shape=(1,), dtype=np.float32)
    else:                                                           # ENV:373-378
        act = Box(np.array(cfg.action_low, dtype=np.float32), np.array(cfg.action_high, dtype=np.float32))
    lr, fr = c.leader.max_rotation_speed, c.follower.max_rotation_speed
    obs = Box(low=np.array((0, 0, 0, 0, -lr, 0, 0, 0, 0, -fr), dtype=np.float32),      # ENV:1812-1824
              high=np.array((c.width, c.height, c.leader.max_speed, 360, lr,
                             c.width, c.height, c.follower.max_speed, 360, fr), dtype=np.float32))
    return act, obs


class Game:
    metadata = {"render.modes": ["rgb_array"]}

    def __init__(self, scenarios=None, device="cuda:0", **kwargs):
        self.cfg = make_config(**kwargs)
        self.action_space, self.observation_space = _spaces(self.cfg)
        self._scenarios = scenarios
        self._device = device
        self._vec = None
        self._seed = 0
        self._resets_since_seed = 0
        self.simulation_number = 0
        self.done = False

    # ------------------------------------------------------------------ gym API
    def seed(self, seed_value):                                    # ENV:429-432
        self._seed = int(seed_value)
        self._resets_since_seed = 0

    def _ensure(self):
        if self._vec is None:
            self._vec = VecGame(1, device=self._device, config=self.cfg)
            pool = self._scenarios
            if isinstance(pool, str):
                pool = ScenarioPool.from_npz(self.cfg, pool, self._device)
            if pool is not None:
                self._vec.load_scenarios(pool)
            self._act = torch.zeros(1, 2, dtype=torch.float64, device=self._device)

    def _generated_pool(self):
        """One-entry pool for this reset: python seed v for the first reset after seed(v), a derived seed afterwards;
        unusable scenarios (see the module docstring) are re-drawn."""
        k = self._resets_since_seed
        for attempt in range(64):
            s = self._seed if (k == 0 and attempt == 0) else int(abi.mix64((self._seed << 20) ^ (k << 8) ^ attempt) >> 2)
            try:
                return ScenarioPool.generate(self.cfg, [s], self._device, n_threads=1)
            except ValueError:
                continue
        raise RuntimeError("no usable scenario in 64 draws")

    def reset(self):                                               # ENV:434-543
        self._ensure()
        if self._scenarios is None:
            self._vec.load_scenarios(self._generated_pool())
        self._resets_since_seed += 1
        idx = torch.tensor([self._seed % self._vec.pool.n], dtype=torch.int32)
        # a reference reset() that raises (SEN:893, 288-297) raises here too; the check reads the error word of the episode in
        # progress, which reset() clears like the sensors the reference rebuilds -- an episode that raised does not poison the next one
        self._vec.reset(idx, check_errors=True, live_errors=True)
        self.simulation_number += 1
        self.done = bool(self._vec.done[0].item())
        return self._obs()

    def step(self, action):                                        # ENV:908-945
        cfg = self.cfg
        if cfg.discrete_action_space:                              # ENV:918-922, decoded on the device (ftl_step_encoded); combined with
            if type(action) is np.ndarray:                         # constant_follower_speed ENV:925 prepends 0.25 to the decoded pair and the
                assert action.shape[0] == 1 and action.shape[1] == 1   # follower's max_speed becomes the rotation: VecGame.step does that too
                action = action[0, 0]
            if action not in cfg.discrete_rotation_speed_to_value:
                raise KeyError(action)
            act = torch.tensor([int(action)], dtype=torch.int32, device=self._device)
        elif cfg.constant_follower_speed:                          # ENV:924-925: np.concatenate([[0.25], action]) -- the speed command of
            a = np.concatenate([[0.25], action])                   # ENV:910-911 is overwritten by ENV:927
            act = torch.tensor([float(a[1])], dtype=torch.float64, device=self._device)
        else:
            self._act[0, 0] = float(action[0])
            self._act[0, 1] = float(action[1])
            act = self._act
        self._vec.step(act, check_errors=True, live_errors=True)   # the reference's exceptions instead of silent error bits
        st = self._vec.status[0].cpu().numpy()
        info = {"mission_status": abi.MISSION[st[0]], "agent_status": abi.AGENT[st[1]], "leader_status": abi.LEADER[st[2]]}
        self.done = bool(self._vec.done[0].item())
        return self._obs(), float(self._vec.reward[0].item()), self.done, info

    def render(self, *a, **k):
        raise NotImplementedError("rendering (pygame display, ENV:1196-1202) is outside the accelerated path")

    def close(self):
        if self._vec is not None:
            self._vec.close()
            self._vec = None

    # ------------------------------------------------------------------ observation dict (ENV:1789-1810)
    def _obs(self):
        v = self._vec
        obs = OrderedDict()
        obs["numerical_features"] = v.obs_num[0].cpu().numpy().copy()
        t = v.target[0].cpu().numpy()
        obs["leader_target_point"] = (float(t[0]), float(t[1]))
        for name, cls in self.cfg.sensor_order:
            if cls == "LeaderPositionsTracker_v2":
                obs[name] = v.tracker_obs(0)            # (leader_positions_hist, corridor), SEN:324-325
            elif cls == "LeaderPositionsTracker":
                continue                                 # the v1 tracker's own dict entry is skipped by use_sensors (CLS:269-270)
            elif cls == "FollowerInfo":
                obs[name] = v.follower_info(name)[0].cpu().numpy().copy()   # [speed / max_speed, direction / 360] float32, SEN:834-842
            elif cls in ("LaserSensor", "LeaderTrackDetector_vector", "LeaderTrackDetector_radar"):
                a = v.aux_view(name)[0].cpu().numpy().copy()                # float32 arrays, SEN:131-134, 381, 476
                spec = next(x for x in self.cfg.aux if x.name == name)
                if spec.params.get("return_all_points"):                    # [K][K rows][zeros] -> the K rows the reference's list holds
                    k, w = int(a[0]), (1 if spec.params["return_only_distances"] else 2)
                    a = a[1:1 + k * w].reshape((k,) if w == 1 else (k, 2))
                obs[name] = a
            else:
                a = v.laser_view(name)[0].cpu().numpy().copy()            # [max_prev_obs, lasers_count] float32, SEN:958
                spec = next(l for l in self.cfg.lasers if l.name == name)
                if spec.lenient:
                    obs[name] = a[0]                                      # LeaderCorridor_lasers_v2: [lasers_count] float32, SEN:803-807
                else:
                    obs[name] = a.astype(np.float64) if spec.pad_sectors else a   # pad_sectors rows are float64 (SEN:933-953)
        return obs


# ---------------------------------------------------------------------------------------------- registry / config files
# The reference registers its env classes with gym (ENV:2132-2172).  The ids whose classes only fix constructor kwargs map
# to Game(**kwargs) here; the manual-control ids have no batched counterpart.
REGISTRY = {
    "Test-Cont-Env-Auto-v0": dict(),                                                               # TestGameAuto, ENV:1962-1964
    "Test-Cont-Env-Auto-Follow-no-obstacles-v0": dict(manual_control=False, add_obstacles=False,    # TestGameBaseAlgoNoObst
                                                      game_width=1500, game_height=1000,
                                                      early_stopping={"max_distance_coef": 1.2, "low_reward": -100}),
    "Test-Game-Neat-v0": dict(manual_control=False, add_obstacles=False, discrete_action_space=True,    # TestGameNEAT
                              early_stopping={"max_distance_coef": 1.2, "low_reward": -100}),
}
MANUAL_IDS = ("Test-Cont-Env-Manual-v0", "Test-Cont-Env-Manual-gazebo-v0", "Test-Cont-Env-Manual-hardcore-v0",
              "Test-Cont-Env-Manual-gazebo-hardcore-v0")


def make(env_id, device="cuda:0", scenarios=None, **kwargs):
    """``gym.make(env_id, **kwargs)`` for the reference's registered ids (same kwargs precedence: the id's own constructor
    kwargs are fixed, the caller's are passed through where the reference class accepts ``**kwargs``)."""
    if env_id in MANUAL_IDS:
        raise NotImplementedError("%s is a manual-control env (pygame event loop); it has no batched counterpart" % env_id)
    if env_id == "Test-Cont-Env-Auto-Follow-with-obstacles-v0":
        raise ValueError("To use it, you need to uncomment the call self._get_green_zone_border_points(). Commented out "
                         "because it slows down the simulation")       # GreenBoxBorderSensor.__init__, SEN:493-495
    if env_id not in REGISTRY:
        raise KeyError(env_id)
    kw = dict(REGISTRY[env_id])
    if env_id == "Test-Cont-Env-Auto-v0":
        kw.update(kwargs)                    # TestGameAuto(**kwargs)
    elif kwargs:
        raise TypeError("%s takes no constructor arguments in the reference" % env_id)
    return Game(scenarios=scenarios, device=device, **kw)


def kwargs_from_params_json(path):
    """Game kwargs of an RLlib ``params.json`` of the reference (``env_config.base_env_config``, e.g.
    src/arctic_gym/server/config/3c1bc/params.json) + the env id and wrapper names next to it.  JSON objects keep their file
    order, which is the dict order the reference's regimes and sensors depend on."""
    import json
    with open(path, "r") as fh:
        d = json.load(fh, object_pairs_hook=OrderedDict)
    ec = d.get("env_config", d)
    return OrderedDict(ec.get("base_env_config", {})), ec.get("name"), list(ec.get("wrappers", []))

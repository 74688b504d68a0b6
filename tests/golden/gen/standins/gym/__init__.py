"""Stand-in for ``gym`` -- FIXTURE-GENERATION INFRASTRUCTURE ONLY (see ../pygame).

gym is not installed in the build container; the reference only needs the
class hierarchy, ``spaces.Box/Discrete`` as plain records and a registry."""
from . import spaces  # noqa: F401
from .envs import registration  # noqa: F401


class Env:
    metadata = {}
    action_space = None
    observation_space = None

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i

    def observation(self, obs):
        raise NotImplementedError


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))


class RewardWrapper(Wrapper):
    pass


def make(id, **kwargs):
    import importlib
    spec = registration.registry[id]
    mod, cls = spec["entry_point"].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)

"""Helpers shared by the parity tests: load the golden episodes emitted by the reference
(tests/golden/gen/make_golden.py) and rebuild the matching config."""
import glob
import json
import os

import numpy as np

from continiousenvironment_follower_leader_amd import make_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def episode_names():
    return sorted(os.path.basename(p)[len("episode_"):-4] for p in glob.glob(os.path.join(GOLDEN, "episode_*.npz")))


def load_episode(name):
    z = np.load(os.path.join(GOLDEN, "episode_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    return z, meta


def config_for(meta, scen_route_len=None, **over):
    kw = dict(meta["kwargs"])
    sensors = kw.get("follower_sensors")
    if sensors and meta.get("post"):
        # config D: the generator sets lasers_count on the constructed sensor; mirror it here
        for sname, attrs in meta["post"].items():
            sensors[sname] = dict(sensors[sname])
            if "lasers_count" in attrs:
                sensors[sname]["lasers_count"] = attrs["lasers_count"]
                sensors[sname]["_allow_any_lasers_count"] = True
    if scen_route_len is not None:
        kw["route_cap"] = max(128, int(scen_route_len))
    kw.update(over)
    return make_config(**kw)


def scenario_arrays(z):
    return dict(static_rects=z["scen:static_rects"], robot_pos=z["scen:robot_pos"],
                robot_dir=z["scen:robot_f64"][:, 0], robot_rect=z["scen:robot_i32"][:, :4],
                route=z["scen:route"], init_traj=z["scen:init_traj"])


# tolerance of BASELINE.json north_star: 1e-5 for float positions / sensor readings / reward.
# Observations are float32: for |x| >= 128 one f32 ulp already exceeds 1e-5, so the check is
# "1e-5 absolute OR one float32 ulp" (hard part 3).
def close(a, b, atol=1e-5):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    ulp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
    return np.abs(a - b) <= np.maximum(atol, ulp)

#!/usr/bin/env python3
"""CPU timing of the UNMODIFIED reference ``Game.step()`` -- build container only (needs /root/reference and the
stand-ins of make_golden.py; never shipped to or run on the GPU box).

N = nproc independent processes, each one reference ``Game`` of config B (35 rocks, 1 dynamic obstacle, tracker_v2 + two
LeaderCorridor_Prev_lasers_v2 sensors) or config A (defaults, no sensors), a warm-up of 50 steps, then ``--steps`` steps under the random
policy of bench.py (v ~ U[0.5, 1] max_speed, w ~ N(0, 0.2 max_rot) clipped) with reset() on done.  reset() time (the Python D* planner,
~1 s) is reported apart and excluded from steps/s: the GPU leg takes its scenarios from a pool as well.

usage: python tests/golden/gen/time_reference.py [--steps 1000] [--procs 8] [--config B] > profiles/r03_reference_python_timing.json
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def worker(args):
    config, steps, seed = args
    import numpy as np
    import make_golden as MG
    r = MG.Runner(config)
    t0 = time.perf_counter()
    r.reset(seed)
    t_reset, n_reset = time.perf_counter() - t0, 1
    g = r.game
    rng = np.random.default_rng(seed)
    t_step, n = 0.0, 0
    for k in range(50 + steps):
        a = MG.random_action(g, rng)
        t0 = time.perf_counter()
        _, _, done, _ = r.step(a)
        dt = time.perf_counter() - t0
        if k >= 50:
            t_step += dt
            n += 1
        if done:
            t0 = time.perf_counter()
            seed += 1000
            r.reset(seed)
            t_reset += time.perf_counter() - t0
            n_reset += 1
    return dict(steps=n, step_seconds=t_step, resets=n_reset, reset_seconds=t_reset)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--config", default="B")
    a = ap.parse_args()
    with mp.Pool(a.procs) as p:
        t0 = time.perf_counter()
        res = p.map(worker, [(a.config, a.steps, 11 + i) for i in range(a.procs)])
        wall = time.perf_counter() - t0
    per_core = [r["steps"] / r["step_seconds"] for r in res]
    out = dict(what="unmodified reference Game.step() of config %s on build-authored pygame / gym stand-ins (tests/golden/gen/standins), "
                    "%d processes x %d steps after 50 warm-up steps; reset() excluded from steps/s" % (a.config, a.procs, a.steps),
               host=platform.processor() or platform.machine(), cores_used=a.procs,
               steps_per_s_per_core=dict(min=min(per_core), mean=sum(per_core) / len(per_core), max=max(per_core)),
               steps_per_s_aggregate=sum(per_core), wall_seconds=wall,
               reset_seconds_mean=sum(r["reset_seconds"] for r in res) / sum(r["resets"] for r in res),
               resets=sum(r["resets"] for r in res), python=platform.python_version())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

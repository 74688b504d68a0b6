"""Throughput of the follower-relative ("Gazebo") tracker + ray sensors batch (row f4): robot-steps/s of ftl_gz_step at N robots, the two
sensors of arctic_env.py:69-90, synthetic motion and 32 lidar point pairs per robot and call.  Information only (not the headline metric)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from continiousenvironment_follower_leader_amd.gazebo import GazeboTrackerBatch

n, mp, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 32, 200
g = GazeboTrackerBatch(n, max_pts=mp)
rng = np.random.default_rng(0)
dev = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")      # noqa: E731
sets = []
lead = np.stack([rng.uniform(5, 9, n), rng.uniform(-3, 3, n)], 1)
for k in range(8):
    delta = np.stack([rng.uniform(0.0, 0.45, n), rng.normal(0, 0.05, n)], 1)
    lead = lead + np.stack([rng.uniform(0.0, 0.5, n), rng.normal(0, 0.15, n)], 1) - delta
    p1 = rng.uniform(-14, 14, (n, mp, 2)); p2 = p1 + rng.uniform(0.3, 1.5, p1.shape)
    sets.append((dev(lead), dev(rng.uniform(-1, 1, n)), dev(delta), dev(p1), dev(p2), dev(np.full(n, mp), torch.int32)))
for k in range(50):
    g.step(*sets[k % 8])
torch.cuda.synchronize()
t = time.perf_counter()
for k in range(steps):
    g.step(*sets[k % 8])
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("gazebo batch: %d robots x %d steps in %.3f s = %.2f M robot-steps/s (%.3f ms per step)" % (n, steps, dt, n * steps / dt / 1e6, dt / steps * 1e3))

#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Game.step() hot path on MI355X.

A "step" is one pass of the hot path over one batch: ONE ftl_step (frame kernel + ray kernel [+ the regroup kernels])
advancing every env of this rank by one Game.step() (frames_per_step frames + one sensor scan), with auto-reset of
finished episodes from the scenario pool inside the same launch.

Workloads (BASELINE.json configs):
  B (default, the configuration the metric is quoted on): 35 rocks + 2 walls + 1 dynamic obstacle, tracker_v2 +
     LeaderCorridor_Prev_lasers_v2 x2 (12 rays all edges, 24 rays obstacles, H=5), 65,536 envs IN TOTAL (all of them on the one GPU at N = 1);
  D: 100 rocks, one 180-ray Prev_lasers_v2 (the LaserPrevSensor replacement), 4,096 envs;
  E: the "hardcore" parameter set (2 bears, leader speed / acceleration regimes, 5 frames per step), 32,768 envs per GPU;
  F: the reference's shipped training config (random 30-70 frames per step, H=10) -- information only.

The env population is AGED in a fixed untimed phase before the counted warm-up (--age steps, default 300), so that the
timed region sees the steady-state mix of episode ages whatever --warmup says.

Multi-GPU: independent contiguous env shards per rank (shard.py), no data-path collective; ONE all_reduce of the 8-entry
episode-metrics vector (accumulated on the device by the frame kernel) after the timed region.  Workload B keeps BASELINE's
65,536 envs IN TOTAL at every N ("scaling": "strong"; 8,192 per GPU at N = 8 = config C of SURVEY.md 8(d)); workload E is defined
per GPU (32,768 each, "weak").  --scaling / --total-envs / --envs-per-gpu override.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W] [--workload B|D|E|F]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# The batch of a rank is stepped as this many independent sub-batches of consecutive envs, each on its own stream of the process and never
# joined inside the timed region (PipelinedVecGame; --parts 1: one batch on one stream).  Measured on one MI355X (profiles/r03_z_parts_sweep.txt):
# 2 parts give +14 % on workload B at 65,536 envs, +7-9 % at 8,192-32,768, +6-12 % on D, E, F, C, L, T; 3 give no more; 4 side streams collapse.
DEFAULT_PARTS = 2


def bytes_per_env_step(cfg, G, Cn):
    """Algorithmic HBM bytes per env-step, SURVEY.md 8(d): B = 104 R + 16 S + 8 G + 24 C + H (16 D + 8) + 4 H sum(N) + 130
    (R robots, S static rects, G green-window points, C corridor points, D dynamic rects seen by the sensors, H history,
    sum(N) rays).  G and C are the survey's measured figures per config (161 / 32 for B and D, 330 / 103 for E and F)."""
    R, S = cfg.n_robots, cfg.c.n_static
    H = max([l.history for l in cfg.lasers] or [0])
    rays = sum(l.count for l in cfg.lasers)
    return 104 * R + 16 * S + 8 * G + 24 * Cn + H * (16 * (R - 1) + 8) + 4 * H * rays + 130


def make_actions(cfg, n, n_sets, seed, device):
    """Synthetic policy output, resident in HBM before the timed region: v ~ U[0.5,1]*max_speed,
    w ~ N(0, 0.2*max_rot) clipped to the action box, float64 as the reference's Python floats."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ms, mr = cfg.c.follower.max_speed, cfg.c.follower.max_rotation_speed
    v = (0.5 + 0.5 * torch.rand(n_sets, n, generator=g, dtype=torch.float64)) * ms
    w = torch.clamp(torch.randn(n_sets, n, generator=g, dtype=torch.float64) * (0.2 * mr), -mr, mr)
    return torch.stack([v, w], dim=-1).contiguous().to(device)


def cpu_baseline(cfg, pool, n_envs, steps, age, seed, label):
    """The CPU oracle (oracle/ftl_oracle.c, a port of the reference's algorithm; the reference itself is Python
    and cannot travel to the GPU box) timed on this host's cores with OpenMP over envs -- a REPORTED baseline on a
    bounded sample of the same workload, not the thing measured above.  Like the GPU leg it runs on an AGED population with
    finished episodes restarted from the next pool entry (the oracle has no in-kernel auto-reset: the host loop resets the done
    envs between steps, inside the timed region)."""
    import ctypes as C
    from oracle import OracleEnv, load_oracle
    t = {k: v.cpu().numpy() for k, v in pool.t.items()}
    P = pool.n
    # the one-GPU box grants a 16-core CPU share even though more cores are visible: use all of the share, never oversubscribe it
    visible = len(os.sched_getaffinity(0))
    cores = min(visible, int(os.environ.get("FTL_CPU_THREADS", "16")))
    lib = load_oracle()
    envs, scen = [], np.arange(n_envs) % P

    def reset(e):
        i = scen[e]
        envs[e].reset(static_rects=t["static_rects"][i], robot_pos=t["robot_pos"][i], robot_dir=t["robot_dir"][i],
                      robot_rect=t["robot_rect"][i], route=t["route"][i, :t["route_len"][i]],
                      init_traj=t["init_traj"][i, :t["init_traj_len"][i]])
    for e in range(n_envs):
        envs.append(OracleEnv(cfg, env_id=e))
        reset(e)
    arr = (C.c_void_p * n_envs)(*[o.h for o in envs])
    L = max(cfg.lasers_len, 1)
    obs = np.zeros((n_envs, 10), np.float32); las = np.zeros((n_envs, L), np.float32); tg = np.zeros((n_envs, 2))
    rew = np.zeros(n_envs); done = np.zeros(n_envs, np.uint8); st = np.zeros((n_envs, 3), np.uint8)
    acts = make_actions(cfg, n_envs, 16, seed, "cpu").numpy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    n_resets = 0

    def run(k0, k1):
        nonlocal n_resets
        for k in range(k0, k1):
            a = np.ascontiguousarray(acts[k % 16])
            lib.ftlo_step_batch(arr, n_envs, p(a), p(obs), p(las), p(tg), p(rew), p(done), p(st), cores)
            for e in np.nonzero(done)[0]:
                scen[e] = (scen[e] + n_envs) % P
                reset(e)
                n_resets += 1
    run(0, age)
    n_resets = 0
    t0 = time.perf_counter()
    run(age, age + steps)
    dt = time.perf_counter() - t0
    return dict(value=n_envs * steps / dt, unit="env-steps/s", cores=cores, kind="port",
                sample="%d config-%s envs x %d steps of oracle/ftl_oracle.c (C restatement of the reference, OpenMP over envs on %d threads = "
                       "the box's CPU share, %d cores visible) after %d ageing steps, %d finished episodes restarted from the pool by the host "
                       "loop inside the timed region, %.1f s wall" % (n_envs, label, steps, cores, visible, age, n_resets, dt))


def build_workload(name, base, seed, device):
    """(cfg, pool, workload text, bytes per env-step, kernel names) of a workload; base = global index of this rank's first env
    (env_id_base: the keys of the per-env random streams of row a12 stay distinct across ranks)."""
    from golden_util import GOLDEN, config_for, load_episode
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool
    if name == "B":
        pool_path = os.path.join(GOLDEN, "pool_B.npz")
        z = np.load(pool_path)
        meta = json.loads(str(z["meta"]))
        cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()), env_id_base=base)
        pool = ScenarioPool.from_npz(cfg, pool_path, device)
        text = ("config B: %d envs/GPU, 35 rocks + 2 walls + 1 dynamic obstacle, tracker_v2 + LeaderCorridor_Prev_lasers_v2 x2 "
                "(12 rays all edges L=100, 24 rays obstacles L=150, H=5), 10 frames/step, auto-reset from a %d-scenario pool "
                "captured from the reference's reset()")
        return cfg, pool, text, 4096, ("ftl_frames_group_kernel<4, false>", "ftl_rays_kernel<5, false, false, false>")     # 4,010 B by the formula, the survey's rounded figure
    ep = {"D": "D_s2_chase", "E": "E_s3_chase", "F": "F_s7_chase", "C": "C_s1_chase", "L": "L_s2_chase", "T": "T_s3_chase"}[name]
    _, m = load_episode(ep)
    cfg = config_for(m, scen_route_len=256, env_id_base=base, rng_seed=seed)
    # the same seed list on every rank (env e of rank r starts from pool entry (seed*1000003 + r*n + e) mod P, shard.scenario_index)
    pool = ScenarioPool.generate(cfg, np.arange(2048 if name != "D" else 1024), device)
    if name == "D":
        text = ("config D: %d envs/GPU, 100 rocks + 2 walls + 1 dynamic obstacle, tracker_v2 + one LeaderCorridor_Prev_lasers_v2 with 180 rays "
                "(obstacles only, L=200, H=5; the LaserPrevSensor replacement), 10 frames/step, auto-reset from a %d-scenario pool built by the host generator")
        return cfg, pool, text, bytes_per_env_step(cfg, 161, 32), ("ftl_frames_group_kernel<4, false>", "ftl_rays_kernel<5, false, false, false>")
    if name == "E":
        text = ("config E (hardcore, ENV:2015-2105 with manual_control=False): %d envs/GPU, 20 rocks + 2 walls + 2 dynamic obstacles, sensors of B, "
                "5 frames/step, leader speed + acceleration regimes on per-env counter streams, early stopping, auto-reset from a %d-scenario "
                "pool built by the host generator")
        return cfg, pool, text, bytes_per_env_step(cfg, 330, 103), ("ftl_frames_group_kernel<4, true>", "ftl_rays_kernel<5, false, false, true>")
    if name in ("C", "L", "T"):
        what = {"C": "tracker_v2 + Prev_lasers_v2 (12 rays) + two LeaderCorridor_lasers_compas (12 and 20 rays, H=5)",
                "L": "tracker_v2 + two LaserSensor lidars (37 x 20 and 13 x 10 marching points) + three leader-track detectors + Prev_lasers_v2 (12 rays)",
                "T": "the v1 LeaderPositionsTracker + LeaderCorridor_lasers + LeaderCorridor_lasers_v2 + two leader-track detectors"}[name]
        text = "config " + name + " (row f3): %d envs/GPU, config B's world (" + ("2" if name == "L" else "1") + " dynamic obstacle) with " + what + \
               ", 10 frames/step, auto-reset from a %d-scenario pool built by the host generator"
        bpe = bytes_per_env_step(cfg, 161, 32) + 4 * sum(a.out_len for a in cfg.aux)
        kn = ("ftl_frames_group_kernel<4, %s>" % ("false"), "ftl_rays_kernel<5, true, false, %s> + ftl_aux_kernel" % ("true" if name == "T" else "false") + (" + ftl_tracker1_kernel" if name == "T" else ""))
        return cfg, pool, text, bpe, kn
    text = ("config F (server/config/3c1bc): %d envs/GPU, 20 rocks + 2 walls + 2 dynamic obstacles, same sensors with H=10, "
            "random 30-70 frames/step, random leader speed regimes, auto-reset from a %d-scenario pool built by the host generator")
    return cfg, pool, text, bytes_per_env_step(cfg, 330, 103), ("ftl_frames_group_kernel<4, true>", "ftl_rays_kernel<10, false, true, true>")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--age", type=int, default=300,
                    help="untimed steps before the counted warm-up that bring the env population to its steady-state mix of "
                         "episode ages (independent of --warmup)")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="fixed batch per GPU (implies weak scaling); 0: the workload's BASELINE size")
    ap.add_argument("--total-envs", type=int, default=0, help="fixed total batch split over the GPUs (implies strong scaling); 0: the workload's "
                                                              "BASELINE size (B: 65,536 in total)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="default: strong for workload B (BASELINE: 65,536 envs in total on 1/2/4/8 GPUs), weak for the per-GPU workloads (E: 32,768 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--cpu-age", type=int, default=150, help="untimed ageing steps of the CPU baseline's population")
    ap.add_argument("--kernel-steps", type=int, default=100, help="steps of the per-kernel HIP-event pass after the timed region")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--parts", type=int, default=0,
                    help="step the batch as this many independent sub-batches, each on its own stream of the process (PipelinedVecGame: one "
                         "part's ray kernel runs beside another part's frame kernel); 1: one batch on one stream; 0: the workload's default")
    ap.add_argument("--ring", type=int, default=0, metavar="HALF",
                    help="draw the scenarios from a ScenarioRing of two halves of HALF entries that generator threads refill while the batch "
                         "steps (fresh worlds, as the reference's reset() builds them) instead of the fixed pool; reports the window moves")
    ap.add_argument("--gen-sample", type=int, default=4096, help="scenarios generated after the timed region to report the generator's rate (0: skip)")
    ap.add_argument("--workload", default="B", choices=["B", "D", "E", "F", "C", "L", "T"],
                    help="B: the configuration the metric is quoted on.  D, E: the other BASELINE configs.  F: the shipped training config.  "
                         "C, L, T (information only): config B's world with the row-f3 sensors -- compas ray sensors / lidars + leader-track "
                         "detectors / the v1 tracker -- which add ftl_aux_kernel (and ftl_tracker1_kernel) to every step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the metrics all-reduce: nccl (= RCCL over xGMI, one GPU per rank) or gloo -- a rehearsal "
                         "of the N > 1 path on a box with fewer GPUs than ranks (every rank then uses GPU rank %% n_visible)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from continiousenvironment_follower_leader_amd import shard
    from continiousenvironment_follower_leader_amd.vec_game import VecGame
    sh, scaling = shard.plan(a.workload, rank, world, a.scaling, a.total_envs, a.envs_per_gpu)
    n = sh.n
    cfg, pool, workload_text, bpe, knames = build_workload(a.workload, sh.lo, a.seed, device)
    parts = a.parts if a.parts > 0 else DEFAULT_PARTS
    parts = max(1, min(parts, n))
    if parts > 1:
        from continiousenvironment_follower_leader_amd.vec_game import PipelinedVecGame
        env = PipelinedVecGame(n, parts=parts, device=device, config=cfg)
    else:
        env = VecGame(n, device=device, config=cfg)
    ring = None
    if a.ring > 0:
        from continiousenvironment_follower_leader_amd.scenario import ScenarioRing
        import itertools
        # (two cores stay with the thread that launches the kernels: at ~3,000 steps/s it is the other thing this process does)
        gthreads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("FTL_CPU_THREADS", "16"))) - 2)
        ring = ScenarioRing(cfg, a.ring, device, itertools.count(1000003 * (rank + 1)), n_threads=gthreads)
        ring.attach(env)
        pool = ring.pool
        env.reset(shard.scenario_index(a.seed, sh.lo, n, a.ring))
    else:
        env.load_scenarios(pool)
        # env e of this rank is GLOBAL env sh.lo + e and starts from scenario (seed*1000003 + sh.lo + e) mod P; auto-reset walks on by n_envs
        env.reset(shard.scenario_index(a.seed, sh.lo, n, pool.n))
    n_sets = 16
    acts = make_actions(cfg, n, n_sets, a.seed * 7919 + rank, device)
    torch.cuda.synchronize()

    k0 = 0
    for k in range(a.age):                     # ageing: untimed, uncounted
        if ring is not None:
            ring.poll(env, k)
        env.step(acts[(k0 + k) % n_sets], auto_reset=True)
    k0 += a.age
    torch.cuda.synchronize()
    env.episode_metrics(clear=True)            # the metrics vector covers warm-up + timed steps only
    for k in range(a.warmup):
        env.step(acts[(k0 + k) % n_sets], auto_reset=True)
    k0 += a.warmup
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    swaps0 = ring.swaps if ring is not None else 0
    for k in range(a.steps):
        if ring is not None:
            ring.poll(env, k0 + k)
        env.step(acts[(k0 + k) % n_sets], auto_reset=True)
    if parts > 1:
        env.join()                             # the launch stream waits for every part's stream: ev1 closes the step of ALL envs
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k0 += a.steps
    step_ms = ev0.elapsed_time(ev1) / a.steps          # HIP events on the launch stream: avg launch-to-launch duration

    # episode metrics: the 8 x f64 vector the frame kernel accumulated on the device (before auto-reset wipes the counters);
    # its all-reduce is the only collective of the path, off the timed region
    metrics = env.episode_metrics().clone()
    n_err, err_bits = env.error_report()
    tmax = torch.tensor([dt, float(n_err), float(n)], dtype=torch.float64, device=device)
    ebits = torch.tensor([(err_bits >> b) & 1 for b in range(16)], dtype=torch.float64, device=device)
    if dist is not None:
        if a.backend == "gloo":                      # gloo reduces host tensors
            tmax, metrics, ebits = tmax.cpu(), metrics.cpu(), ebits.cpu()
        tm = tmax[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tmax[0] = tm[0]
        shard.reduce_metrics(metrics)
        sums = tmax[1:].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        tmax[1:] = sums
        dist.all_reduce(ebits, op=dist.ReduceOp.MAX)   # bitwise OR of the ranks' error words, one bit per element
    dt = float(tmax[0].item())
    n_err = int(tmax[1].item())
    err_bits = sum(1 << b for b in range(16) if ebits[b].item() > 0)
    total_envs = int(tmax[2].item())                   # the units all ranks processed per step
    value = total_envs * a.steps / dt

    # per-kernel durations: a separate pass on the same (still steady-state) population with HIP events around each launch
    ktimes, kmode = None, "HIP events around every launch of %d further steps on the same population" % a.kernel_steps
    if parts > 1:
        kmode += " with the %d parts run one after the other on one stream (each kernel timed alone; averages per launch = per part)" % parts
    if rank == 0 and a.kernel_steps > 0:
        tenv = env
        try:
            env.kernel_timing(True)
        except NotImplementedError:
            # two-stream mode (random_frames_per_step: the halves of the batch overlap on two streams, so no kernel has a duration of
            # its own).  The per-kernel figures then come from a single-stream handle (FTL_SPLIT=0) continuing the SAME population.
            os.environ["FTL_SPLIT"] = "0"
            tenv = VecGame(n, device=device, config=cfg)
            del os.environ["FTL_SPLIT"]
            tenv.load_scenarios(pool)
            torch.cuda.synchronize()
            tenv.state.view(-1)[tenv._state_off:tenv._state_off + env.state.numel() - 256].copy_(
                env.state.view(-1)[env._state_off:env._state_off + env.state.numel() - 256])
            tenv.kernel_timing(True)
            kmode = "a single-stream pass (FTL_SPLIT=0) of %d steps continuing the same population; the timed region itself runs the two halves " \
                    "of the batch on two streams" % a.kernel_steps
        for k in range(a.kernel_steps):
            tenv.step(acts[(k0 + k) % n_sets], auto_reset=True)
        ktimes = tenv.kernel_times()
        tenv.kernel_timing(False)
        if tenv is not env:
            tenv.close()

    supply = None
    if rank == 0 and a.gen_sample > 0:           # reset-time scenario generation (row f2): the host generator's rate on this box's cores
        from continiousenvironment_follower_leader_amd.scenario import generate_scenarios
        threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("FTL_CPU_THREADS", "16")))
        tg = time.perf_counter()
        gsc = generate_scenarios(cfg, np.arange(5000000, 5000000 + a.gen_sample), threads)
        tg = time.perf_counter() - tg
        supply = {"generator_scenarios_per_s": a.gen_sample / tg, "threads": threads, "usable_fraction": float(gsc["usable"].mean())}
    if ring is not None:
        supply = dict(supply or {}, ring_half=a.ring, window_moves_in_timed_region=ring.swaps - swaps0, scenarios_generated=ring.generated)
        ring.close()
    if rank == 0:
        m = metrics.tolist()
        if ktimes:
            # the dominant kernel of the step
            dom = max(("frames", "rays", "aux"), key=lambda k: ktimes[k + "_us"])
        # One step = every env advanced once: the launches of one ftl_step per part -- the frame kernel, the ray kernel (+ ftl_aux_kernel, + the
        # regroup kernels every k-th step) -- on the part's stream.  SURVEY 8(d)'s algorithmic bytes per env-step cover the whole step, so the
        # roofline figure is bytes of one step / duration of one step, the duration from the HIP events that bracket the timed region on the
        # launch stream (the part streams start after the first event and are joined before the second: step_ms_events).  With more than one
        # part the launches of different streams overlap, so no kernel has a share of the step of its own; kernels_us gives what each LAUNCH
        # takes alone, from a separate pass that runs the parts one after the other with an event after every launch (diagnostics, not the
        # denominator; each such event adds ~4.5 us of its own).
        kernel_ms = step_ms
        launch_bytes = bpe * n
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        prof = {}
        ppath = os.path.join(ROOT, "profiles", "pmc_current.json")
        if os.path.exists(ppath):
            try:
                prof = json.load(open(ppath)).get(a.workload, {})
            except Exception:
                prof = {}
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": prof.get("hbm_bytes_per_step"), "traffic_source": prof.get("source"),
                "kernel": "%s + %s (one step = both launches for every one of the %d part(s) + the regroup kernels every 4th / 8th step; achieved = "
                          "algorithmic bytes of a step / step duration from HIP events over the timed region; HBM is the contract roofline, the ray "
                          "kernel is VALU-issue bound and the frame kernel latency bound -- see valu)" % (knames[0], knames[1], parts),
                "kernel_ms": kernel_ms, "step_ms_events": step_ms, "bytes_per_env_step": bpe,
                "kernels_us": ktimes, "kernels_us_from": kmode if ktimes else None, "valu": prof.get("valu")}
        if ktimes:
            roof["dominant_kernel"] = {"rays": knames[1], "frames": knames[0], "aux": "ftl_aux_kernel"}[dom]
        line = {
            "metric": "env-steps/sec at 65,536 parallel envs; 1/2/4/8 MI355X scaling",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text % (n, pool.n), "workload_id": a.workload,
                       "envs_per_gpu": n, "total_envs": total_envs,
                       "parts": parts, "stepping": ("%d independent sub-batches of consecutive envs, each on its own stream of the process, never joined inside the "
                                                    "timed region (PipelinedVecGame; bit-identical to one batch, tests/test_gpu_api.py)" % parts) if parts > 1
                                                   else "one batch on one stream",
                       "parallelism": "independent contiguous env shards x%d, %s%s" % (world, "the same total at every N" if scaling == "strong" else "a fixed batch per GPU",
                                                                                 " (gloo rehearsal)" if (world > 1 and a.backend == "gloo") else ""),
                       "age_steps": a.age,
                       "episode_metrics": dict(zip(shard.METRIC_NAMES, m)),
                       "mean_return": m[1] / m[0] if m[0] else None, "mean_episode_frames": m[2] / m[0] if m[0] else None,
                       "resets_per_s": m[0] / (dt * (a.steps + a.warmup) / a.steps) if dt > 0 else None,     # episodes that ended (= auto-resets) per second
                       "scenario_supply": supply,
                       "envs_with_error_flags": n_err, "error_bits": err_bits},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, pool, min(a.cpu_envs, n), a.cpu_steps, a.cpu_age, a.seed, a.workload)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Game.step() hot path on MI355X.

A "step" is one pass of the hot path over one batch: ONE ftl_step (frame kernel + ray kernel [+ the regroup kernels])
advancing every env of this rank by one Game.step() (frames_per_step frames + one sensor scan), with auto-reset of
finished episodes from the scenario pool inside the same launch.

Workloads (BASELINE.json configs):
  B (default, the configuration the metric is quoted on): 35 rocks + 2 walls + 1 dynamic obstacle, tracker_v2 +
     LeaderCorridor_Prev_lasers_v2 x2 (12 rays all edges, 24 rays obstacles, H=5), 65,536 envs per GPU;
  D: 100 rocks, one 180-ray Prev_lasers_v2 (the LaserPrevSensor replacement), 4,096 envs;
  E: the "hardcore" parameter set (2 bears, leader speed / acceleration regimes, 5 frames per step), 32,768 envs per GPU;
  F: the reference's shipped training config (random 30-70 frames per step, H=10) -- information only.

The env population is AGED in a fixed untimed phase before the counted warm-up (--age steps, default 300), so that the
timed region sees the steady-state mix of episode ages whatever --warmup says.

Multi-GPU: independent env shards per rank (weak scaling; shard.py), no data-path collective; ONE all_reduce of the
8-entry episode-metrics vector (accumulated on the device by the frame kernel) after the timed region.

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

DEFAULT_ENVS = {"B": 65536, "D": 4096, "E": 32768, "F": 65536, "C": 65536, "L": 65536, "T": 65536}


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


def cpu_baseline(cfg, pool, n_envs, steps, seed, label):
    """The CPU oracle (oracle/ftl_oracle.c, a port of the reference's algorithm; the reference itself is Python
    and cannot travel to the GPU box) timed on this host's cores with OpenMP over envs -- a REPORTED baseline on a
    bounded sample of the same workload, not the thing measured above."""
    import ctypes as C
    from oracle import OracleEnv, load_oracle
    t = {k: v.cpu().numpy() for k, v in pool.t.items()}
    P = pool.n
    # the one-GPU box grants a 16-core share even though more cores are visible: never oversubscribe it
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("FTL_CPU_THREADS", "16")))
    lib = load_oracle()
    envs = []
    for e in range(n_envs):
        i = e % P
        o = OracleEnv(cfg)
        o.reset(static_rects=t["static_rects"][i], robot_pos=t["robot_pos"][i], robot_dir=t["robot_dir"][i],
                robot_rect=t["robot_rect"][i], route=t["route"][i, :t["route_len"][i]],
                init_traj=t["init_traj"][i, :t["init_traj_len"][i]])
        envs.append(o)
    arr = (C.c_void_p * n_envs)(*[o.h for o in envs])
    L = max(cfg.lasers_len, 1)
    obs = np.zeros((n_envs, 10), np.float32); las = np.zeros((n_envs, L), np.float32); tg = np.zeros((n_envs, 2))
    rew = np.zeros(n_envs); done = np.zeros(n_envs, np.uint8); st = np.zeros((n_envs, 3), np.uint8)
    acts = make_actions(cfg, n_envs, steps, seed, "cpu").numpy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    t0 = time.perf_counter()
    for k in range(steps):
        a = np.ascontiguousarray(acts[k])
        lib.ftlo_step_batch(arr, n_envs, p(a), p(obs), p(las), p(tg), p(rew), p(done), p(st), cores)
    dt = time.perf_counter() - t0
    return dict(value=n_envs * steps / dt, unit="env-steps/s", cores=cores, kind="port",
                sample="%d config-%s envs x %d steps of oracle/ftl_oracle.c (C restatement of the reference, OpenMP over envs, "
                       "no auto-reset), %.1f s wall" % (n_envs, label, steps, dt))


def build_workload(name, n, rank, seed, device):
    """(cfg, pool, workload text, bytes per env-step, kernel names) of a workload."""
    from golden_util import GOLDEN, config_for, load_episode
    from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool
    base = rank * n           # env_id_base: global env ids (keys of the per-env random streams of row a12) stay distinct across ranks
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
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="0: the workload's BASELINE size (B 65,536; D 4,096; E 32,768)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--kernel-steps", type=int, default=100, help="steps of the per-kernel HIP-event pass after the timed region")
    ap.add_argument("--seed", type=int, default=0)
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
    n = a.envs_per_gpu or DEFAULT_ENVS[a.workload]
    cfg, pool, workload_text, bpe, knames = build_workload(a.workload, n, rank, a.seed, device)
    env = VecGame(n, device=device, config=cfg)
    env.load_scenarios(pool)
    # env e of rank r is GLOBAL env r*n + e and starts from scenario (seed*1000003 + r*n + e) mod P; auto-reset walks on by n_envs
    env.reset(shard.scenario_index(a.seed, rank * n, n, pool.n))
    n_sets = 16
    acts = make_actions(cfg, n, n_sets, a.seed * 7919 + rank, device)
    torch.cuda.synchronize()

    k0 = 0
    for k in range(a.age):                     # ageing: untimed, uncounted
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
    for k in range(a.steps):
        env.step(acts[(k0 + k) % n_sets], auto_reset=True)
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
    tmax = torch.tensor([dt, float(n_err)], dtype=torch.float64, device=device)
    if dist is not None:
        if a.backend == "gloo":                      # gloo reduces host tensors
            tmax, metrics = tmax.cpu(), metrics.cpu()
        tm = tmax[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tmax[0] = tm[0]
        shard.reduce_metrics(metrics)
        errs = tmax[1:].clone()
        dist.all_reduce(errs, op=dist.ReduceOp.SUM)
        n_err = int(errs.item())
    dt = float(tmax[0].item())
    total_envs = n * world
    value = total_envs * a.steps / dt

    # per-kernel durations: a separate pass on the same (still steady-state) population with HIP events around each launch
    ktimes = None
    if rank == 0 and a.kernel_steps > 0:
        try:
            env.kernel_timing(True)
            for k in range(a.kernel_steps):
                env.step(acts[(k0 + k) % n_sets], auto_reset=True)
            ktimes = env.kernel_times()
            env.kernel_timing(False)
        except NotImplementedError:
            ktimes = None                               # two-stream mode (workload F)

    if rank == 0:
        m = metrics.tolist()
        if ktimes:
            # the dominant kernel's roofline: algorithmic bytes of one launch / its average duration
            dom = "rays" if ktimes["rays_us"] >= ktimes["frames_us"] else "frames"
        kernel_ms = (ktimes["frames_us"] + ktimes["rays_us"] + ktimes["regroup_us"]) * 1e-3 if ktimes else step_ms
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
                "kernel": "%s + %s (one step = both launches on one stream + the regroup kernels every 2nd step; HBM is the "
                          "contract roofline, the ray kernel is VALU-issue bound -- see valu)" % knames,
                "kernel_ms": kernel_ms, "step_ms_events": step_ms, "bytes_per_env_step": bpe,
                "kernels_us": ktimes, "valu": prof.get("valu")}
        if ktimes:
            roof["dominant_kernel"] = knames[1] if dom == "rays" else knames[0]
        line = {
            "metric": "env-steps/sec at 65,536 parallel envs; 1/2/4/8 MI355X scaling",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text % (n, pool.n), "workload_id": a.workload,
                       "envs_per_gpu": n, "total_envs": total_envs, "parallelism": "independent env shards x%d%s" % (world, " (gloo rehearsal)" if (world > 1 and a.backend == "gloo") else ""),
                       "age_steps": a.age,
                       "episode_metrics": dict(zip(shard.METRIC_NAMES, m)),
                       "mean_return": m[1] / m[0] if m[0] else None, "mean_episode_frames": m[2] / m[0] if m[0] else None,
                       "envs_with_error_flags": n_err, "error_bits": err_bits},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, pool, min(a.cpu_envs, n), a.cpu_steps, a.seed, a.workload)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Multi-GPU sharding of the env batch.

Envs are fully independent: rank r of W owns a contiguous block of env indices, there is no exchange inside
``step``.  The only collective is an all-reduce of the episode-metrics vector (64 B) every K steps -- RCCL over xGMI
on GPUs (backend "nccl" is RCCL on ROCm), gloo in the CPU tests.

``bench.py`` uses exactly these functions: ``scenario_index`` for the initial scenario of every global env,
``VecGame.episode_metrics()`` (the vector the frame kernel accumulates on the device, include/ftl.h FTL_M_*) as the
per-rank vector and ``reduce_metrics`` for the collective.  ``episode_metrics_from_outputs`` is the host-side
definition of the same vector from step outputs: the tests use it to check the device accumulator."""
from dataclasses import dataclass

import torch

METRIC_NAMES = ("episodes", "return_sum", "frames_sum", "success", "crash", "low_reward", "too_far", "timeout")


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    lo: int
    hi: int

    @property
    def n(self):
        return self.hi - self.lo


def shard_range(n_total, rank, world):
    """Contiguous env range of ``rank``; the first ``n_total % world`` ranks take one extra env."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return Shard(rank, world, lo, lo + base + (1 if rank < extra else 0))


# bench.py's batch sizes (BASELINE.json configs).  Workload B is quoted on 65,536 envs IN TOTAL: on N GPUs the same
# 65,536 envs are split into N contiguous shards (config C of the survey: 8,192 per GPU at N = 8) -- strong scaling.  Config E is
# defined per GPU (32,768 each, 262,144 at N = 8) -- weak scaling; the information-only workloads keep a fixed size per GPU too.
TOTAL_ENVS = {"B": 65536}
ENVS_PER_GPU = {"D": 4096, "E": 32768, "F": 65536, "C": 65536, "L": 65536, "T": 65536}


def plan(workload, rank, world, scaling=None, total_envs=0, envs_per_gpu=0):
    """(Shard of this rank, "strong" | "weak") for a bench workload.  ``total_envs`` / ``envs_per_gpu`` override the BASELINE size;
    ``scaling`` overrides the workload's own mode (strong = a fixed total split over the ranks, weak = a fixed size per rank)."""
    if total_envs and envs_per_gpu:
        raise ValueError("give total_envs or envs_per_gpu, not both")
    if scaling is None:
        scaling = "weak" if envs_per_gpu else ("strong" if (total_envs or workload in TOTAL_ENVS) else "weak")
    if scaling == "strong":
        total = total_envs or TOTAL_ENVS.get(workload) or envs_per_gpu or ENVS_PER_GPU[workload]
        return shard_range(int(total), rank, world), "strong"
    if scaling != "weak":
        raise ValueError("scaling must be 'strong' or 'weak'")
    per = envs_per_gpu or total_envs or ENVS_PER_GPU.get(workload) or TOTAL_ENVS[workload]
    return Shard(rank, world, rank * int(per), (rank + 1) * int(per)), "weak"


def scenario_index(seed, env_lo, n, pool_size):
    """Scenario of global env e: pool[(seed*1000003 + e) mod P]; int32 tensor for this shard."""
    e = torch.arange(env_lo, env_lo + n, dtype=torch.int64)
    return ((e + seed * 1000003) % pool_size).to(torch.int32)


def episode_metrics_from_outputs(done, status, reward_sum, frames):
    """Metrics vector [episodes, sum return, sum frames, n_success, n_crash, n_low_reward, n_too_far, n_timeout] (f64[8])
    of the episodes that END in one step: ``done`` = the env's episode finished in this step, status = [n,3] codes of
    include/ftl.h at that step, ``reward_sum`` / ``frames`` = overall_reward and step_count of every env after the step
    (ENV:941-944)."""
    d = done.bool()
    st = status
    f64 = torch.float64
    zero = torch.zeros((), dtype=f64, device=done.device)
    return torch.stack([
        d.sum().to(f64), (reward_sum * d).sum().to(f64) if reward_sum is not None else zero,
        (frames * d).sum().to(f64) if frames is not None else zero,
        (d & (st[:, 0] == 2)).sum().to(f64), (d & (st[:, 1] == 1)).sum().to(f64), (d & (st[:, 1] == 2)).sum().to(f64),
        (d & (st[:, 1] == 3)).sum().to(f64), (d & (st[:, 0] == 3)).sum().to(f64)])


def reduce_metrics(vec, group=None):
    """Sum the metrics vector over all ranks (in place) -- the only collective of the path."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec

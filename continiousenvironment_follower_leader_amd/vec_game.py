"""``VecGame`` -- N independent follow-the-leader environments advanced by one HIP kernel launch per
``step()`` (the batched counterpart of the reference's ``Game.reset/step``,
follow_the_leader_continuous_env.py:434-543, 908-945).

PyTorch is used for plumbing only: it owns the device buffers (state blob, scenario pool, outputs) and the
stream; all arithmetic happens in ``libftl_hip.so`` behind the C-ABI of ``include/ftl.h``."""
import ctypes as C
from contextlib import nullcontext as _nullcontext

import numpy as np
import torch

from . import _lib, abi
from .config import GameConfig, make_config

_DT = {0: torch.int32, 1: torch.float32, 2: torch.float64}


class ScenarioPool:
    """Post-reset scenarios (the output of the reference's reset-time generation, ENV:434-539) as device arrays."""

    @staticmethod
    def pack(cfg: GameConfig, static_rects, robot_pos, robot_dir, robot_rect, routes, init_trajs):
        """Host arrays in the layout of ``ftl_scenarios`` (routes / initial trajectories padded to route_cap / init_traj_cap)."""
        c = cfg.c
        P = len(robot_pos)
        R = cfg.n_robots
        sr = np.asarray(static_rects, np.int32).reshape(P, -1, 4)
        if sr.shape[1] != c.n_static:
            raise ValueError("scenario has %d static rects, config expects %d" % (sr.shape[1], c.n_static))
        route = np.zeros((P, c.route_cap, 2), np.float64)
        route_len = np.zeros(P, np.int32)
        it = np.zeros((P, c.init_traj_cap, 2), np.float32)
        it_len = np.zeros(P, np.int32)
        for i in range(P):
            r = np.asarray(routes[i], np.float64).reshape(-1, 2)
            t = np.asarray(init_trajs[i], np.float32).reshape(-1, 2)
            if len(r) > c.route_cap:
                raise ValueError("route of scenario %d has %d way-points > route_cap %d" % (i, len(r), c.route_cap))
            if len(r) == 1:
                raise ValueError("a one-point route makes the reference's reset raise IndexError (ENV:513)")
            if len(t) > c.init_traj_cap:
                raise ValueError("initial trajectory of scenario %d has %d points > init_traj_cap %d"
                                 % (i, len(t), c.init_traj_cap))
            route[i, :len(r)] = r
            route_len[i] = len(r)
            it[i, :len(t)] = t
            it_len[i] = len(t)
        return dict(static_rects=np.ascontiguousarray(sr),
                    robot_pos=np.ascontiguousarray(np.asarray(robot_pos, np.float32).reshape(P, R, 2)),
                    robot_dir=np.ascontiguousarray(np.asarray(robot_dir, np.float64).reshape(P, R)),
                    robot_rect=np.ascontiguousarray(np.asarray(robot_rect, np.int32).reshape(P, R, 4)),
                    route=route, route_len=route_len, init_traj=it, init_traj_len=it_len)

    def __init__(self, cfg: GameConfig, static_rects, robot_pos, robot_dir, robot_rect, routes, init_trajs, device):
        host = self.pack(cfg, static_rects, robot_pos, robot_dir, robot_rect, routes, init_trajs)
        dev = torch.device(device)
        self.n = len(host["robot_pos"])
        self.t = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
        self._bind()

    def _bind(self):
        s = abi.Scenarios()
        s.n_scenarios = self.n
        for k, v in self.t.items():
            setattr(s, k, v.data_ptr())
        self.c_struct = s

    @classmethod
    def empty(cls, cfg: GameConfig, capacity, device):
        """A pool of ``capacity`` zeroed entries to be filled with ``write`` (ScenarioRing)."""
        c, R = cfg.c, cfg.n_robots
        shapes = dict(static_rects=((capacity, c.n_static, 4), torch.int32), robot_pos=((capacity, R, 2), torch.float32),
                      robot_dir=((capacity, R), torch.float64), robot_rect=((capacity, R, 4), torch.int32),
                      route=((capacity, c.route_cap, 2), torch.float64), route_len=((capacity,), torch.int32),
                      init_traj=((capacity, c.init_traj_cap, 2), torch.float32), init_traj_len=((capacity,), torch.int32))
        self = cls.__new__(cls)
        self.n = int(capacity)
        self.t = {k: torch.zeros(sh, dtype=dt, device=device) for k, (sh, dt) in shapes.items()}
        self._bind()
        return self

    def write(self, base, host, stream=None):
        """Copy host arrays (``pack`` layout, pinned for an asynchronous copy) over entries ``[base, base + n)`` on ``stream``."""
        n = len(host["robot_pos"])
        if base < 0 or base + n > self.n:
            raise ValueError("entries outside the pool")
        with torch.cuda.stream(stream) if stream is not None else _nullcontext():
            for k, v in host.items():
                self.t[k][base:base + n].copy_(v if isinstance(v, torch.Tensor) else torch.from_numpy(v), non_blocking=True)

    @classmethod
    def generate(cls, cfg, seeds, device, n_threads=0):
        """Pool of the usable scenarios among python seeds ``seeds`` from the host-side generator (scenario.py): what
        ``game.seed(s); game.reset()`` builds in the reference, without the reference."""
        from .scenario import generate_scenarios
        g = generate_scenarios(cfg, seeds, n_threads)
        keep = np.nonzero(g["usable"])[0]
        if len(keep) == 0:
            raise ValueError("no usable scenario among the given seeds")
        pool = cls.__new__(cls)
        pool.n = len(keep)
        pool.t = {k: torch.from_numpy(np.ascontiguousarray(g[k][keep])).to(device) for k in
                  ("static_rects", "robot_pos", "robot_dir", "robot_rect", "route", "route_len", "init_traj", "init_traj_len")}
        pool._bind()
        pool.seeds = g["seed"][keep]
        return pool

    @classmethod
    def from_npz(cls, cfg, path, device, limit=None):
        z = np.load(path)
        n = len(z["seed"]) if limit is None else min(limit, len(z["seed"]))
        routes = [z["route"][i, :z["route_len"][i]].astype(np.float64) for i in range(n)]
        trajs = [z["init_traj"][i, :z["init_traj_len"][i]] for i in range(n)]
        return cls(cfg, z["static_rects"][:n].astype(np.int32), z["robot_pos"][:n], z["robot_dir"][:n],
                   z["robot_rect"][:n].astype(np.int32), routes, trajs, device)


def error_for_bits(bits, n_envs=1):
    """Exception object for a set of FTL_ERR_* bits: the type the reference raises where it has one."""
    where = "%d env(s)" % n_envs
    if bits & abi.FTL_ERR_BAD_ACTION:           # ENV:922: discrete_rotation_speed_to_value[action] with an action outside 0..4
        return KeyError("Discrete(5) action outside 0..4 in %s" % where)
    if bits & abi.FTL_ERR_TRACKER_SEED:         # SEN:264-297 (the tracker is scanned before every ray sensor, CLS:263-267)
        return IndexError("pop from an empty deque (tracker seeded with fewer than 2 points or trimmed before the corridor "
                          "exists, sensors.py:288-297; %s)" % where)
    if bits & abi.FTL_ERR_EMPTY_CORRIDOR:       # SEN:893/962: `all_obs_arr` is unbound when len(corridor) <= 1
        return UnboundLocalError("local variable 'all_obs_arr' referenced before assignment (ray sensor scanned with a "
                                 "corridor of <= 1 points, sensors.py:893-962; %s)" % where)
    names = [n for b, n in ((abi.FTL_ERR_TRAJ_OVERFLOW, "traj_cap"), (abi.FTL_ERR_CORR_OVERFLOW, "corr_cap"), (abi.FTL_ERR_HIST1_OVERFLOW, "hist1_cap"),
                            (abi.FTL_ERR_LIDAR_OVERFLOW, "objects within a lidar's range")) if bits & b]
    return _lib.FtlError("capacity overflow of the batched state (%s) in %s: results after the overflow differ from the "
                         "reference -- raise the capacity in make_config()" % (", ".join(names) or hex(bits), where))


class VecGame:
    """N parallel envs on one GPU.

    ``reset(scen_idx, mask)`` / ``step(action, auto_reset)`` return views of persistent device tensors:
    ``obs_num`` f32[N,10] (numerical_features, ENV:1793-1802), ``lasers`` f32[N, sum_k H_k*N_k] (one
    ``[H_k, N_k]`` block per ray sensor, ``laser_view(name)``), ``target`` f64[N,2], ``reward`` f64[N],
    ``done`` u8[N], ``status`` u8[N,3] (mission/agent/leader codes of ``abi.MISSION/AGENT/LEADER``)."""

    def __init__(self, n_envs, device="cuda:0", config: GameConfig = None, policy_obs=False, _outputs=None, **game_kwargs):
        self.cfg = config if config is not None else make_config(**game_kwargs)
        self.n = int(n_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.FtlError("VecGame needs a ROCm device (got %s): there is no CPU path" % self.device)
        if not torch.cuda.is_available():
            raise _lib.FtlError("no ROCm device is visible: the batched env runs on the GPU only (there is no CPU path)")
        self.lib = _lib.load()
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.ftl_create(C.byref(self.cfg.c), self.n, dev_index, C.byref(h)), self.lib)
        self.h = h
        nbytes = self.lib.ftl_state_bytes(self.h)
        with torch.cuda.device(self.device):
            self.state = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = self.state.data_ptr()
        self._state_off = (-base) % 256
        _lib.check(self.lib.ftl_bind_state(self.h, base + self._state_off, nbytes), self.lib)
        L = max(self.cfg.lasers_len, 1)
        z = dict(device=self.device)
        def out(name, *shape, dtype):          # an output tensor of this batch, or the rows of a larger one that the caller owns (PipelinedVecGame)
            if _outputs is None:
                return torch.zeros(self.n, *shape, dtype=dtype, **z)
            t = _outputs[name]
            if tuple(t.shape) != (self.n, *shape) or t.dtype != dtype or not t.is_contiguous() or t.device != self.device:
                raise ValueError("output tensor %r does not fit this batch" % name)
            return t
        self.obs_num = out("obs_num", abi.FTL_OBS_NUM, dtype=torch.float32)
        self.lasers = out("lasers", L, dtype=torch.float32)
        self.target = out("target", 2, dtype=torch.float64)
        self.reward = out("reward", dtype=torch.float64)
        self.done = out("done", dtype=torch.uint8)
        self.status = out("status", 3, dtype=torch.uint8)
        o = abi.Outputs()
        o.obs_num, o.lasers, o.target = self.obs_num.data_ptr(), self.lasers.data_ptr(), self.target.data_ptr()
        o.reward, o.done, o.status = self.reward.data_ptr(), self.done.data_ptr(), self.status.data_ptr()
        # fused ContinuousObserveModifier_sensorPrev output (wrappers.py:169-221): [n, H, sum of row widths], float32, over the
        # sensor classes the wrapper concatenates (LaserSpec.in_policy_obs), in dict order
        sel = [l for l in self.cfg.lasers if l.in_policy_obs]
        hs = {l.history for l in sel}
        self.policy_obs = None
        if policy_obs and sel:
            if len(hs) != 1:
                raise ValueError("policy_obs needs the same max_prev_obs on every sensor it concatenates (wrappers.py:207, 217 assert it)")
            self.policy_obs = out("policy_obs", hs.pop(), sum(l.width for l in sel), dtype=torch.float32)
            o.policy_obs = self.policy_obs.data_ptr()
        self._metrics = torch.zeros(abi.FTL_N_METRICS, dtype=torch.float64, **z)
        self._errors = torch.zeros(2, dtype=torch.int32, **z)
        self._out = o
        self.pool = None
        self._fields = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.ftl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ scenarios / reset / step
    def load_scenarios(self, pool: ScenarioPool):
        self.pool = pool
        _lib.check(self.lib.ftl_load_scenarios(self.h, C.byref(pool.c_struct)), self.lib)

    def set_reset_window(self, base, count, stride=0):
        """Pool entries ``[base, base + count)`` the in-kernel auto-reset draws from and the step of its walk (``ftl_set_reset_window``;
        the whole pool with stride n_envs after ``load_scenarios``): how a ``ScenarioRing`` hands freshly generated worlds to a running
        batch.  ``stride`` should be coprime to ``count`` (0 keeps n_envs)."""
        _lib.check(self.lib.ftl_set_reset_window(self.h, int(base), int(count), int(stride)), self.lib)

    def tune(self, coscheduled_envs=None, regroup_every=None, two_streams=None):
        """Scheduling hints (``ftl_tune``; results never depend on them): how many envs are stepped on the device at the same time when
        this batch is one of several on several streams, how often the cost order of the envs is rebuilt, the handle's own two-stream mode."""
        for key, v in ((abi.FTL_TUNE_COSCHEDULED_ENVS, coscheduled_envs), (abi.FTL_TUNE_REGROUP_EVERY, regroup_every), (abi.FTL_TUNE_TWO_STREAMS, two_streams)):
            if v is not None:
                _lib.check(self.lib.ftl_tune(self.h, key, int(v)), self.lib)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, scen_idx=None, mask=None, check_errors=False, live_errors=False):
        if self.pool is None:
            raise _lib.FtlError("load_scenarios() first")
        if scen_idx is None:
            scen_idx = torch.arange(self.n, dtype=torch.int32, device=self.device) % self.pool.n
        scen_idx = torch.as_tensor(scen_idx, dtype=torch.int32, device=self.device).contiguous()
        if scen_idx.numel() != self.n:
            raise ValueError("scen_idx must have one entry per env")
        if bool((scen_idx < 0).any()) or bool((scen_idx >= self.pool.n).any()):
            raise ValueError("scen_idx out of range")
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, dtype=torch.uint8, device=self.device).contiguous()
            mptr = mask.data_ptr()
        self._keep = (scen_idx, mask)
        _lib.check(self.lib.ftl_reset(self.h, scen_idx.data_ptr(), mptr, C.byref(self._out), self._stream()), self.lib)
        if check_errors:
            self.raise_on_errors(live_errors)
        return self.obs_num, self.lasers

    def step(self, action, auto_reset=False, check_errors=False, live_errors=False):
        """action: f64[N,2] device tensor = (speed px/frame, signed rotation deg/frame) (ENV:927-933).  With
        ``discrete_action_space=True`` an integer tensor [N] (or [N,1]) of Discrete(5) indices, with ``constant_follower_speed=True`` a
        float tensor [N] (or [N,1]) of rotations: both are decoded on the device as ENV:909-925 does (``ftl_step_encoded``).
        ``check_errors=True`` synchronises and raises what the reference would have raised in any env (``raise_on_errors``);
        the default leaves the per-env sticky error words for ``error_report()`` so that the step stays asynchronous."""
        action, enc = self._encode_action(action, self.n)
        flags = abi.FTL_STEP_AUTO_RESET if auto_reset else 0
        _lib.check(self.lib.ftl_step_encoded(self.h, action.data_ptr(), enc, C.byref(self._out), flags, self._stream()), self.lib)
        if check_errors:
            self.raise_on_errors(live_errors)
        return self.obs_num, self.lasers, self.reward, self.done, self.status

    def _encode_action(self, action, n):
        """(tensor to hand to ftl_step_encoded, FTL_ACTION_* encoding) for an action of ``n`` envs of this config (ENV:909-925)."""
        enc = abi.FTL_ACTION_BOX2
        if action.device != self.device:
            raise ValueError("action must live on %s" % self.device)
        if self.cfg.discrete_action_space or self.cfg.constant_follower_speed:
            if tuple(action.shape) not in ((n,), (n, 1)):
                raise ValueError("action must be [n_envs] or [n_envs, 1] for this action space (ENV:358-372)")
            if self.cfg.discrete_action_space and self.cfg.constant_follower_speed:
                # ENV:922 then ENV:925: np.concatenate([[0.25], (max_speed, rotation)]) -- the follower's max_speed ends up as the rotation
                action = torch.tensor([0.25, self.cfg.c.follower.max_speed], dtype=torch.float64, device=self.device).repeat(n, 1)
            elif self.cfg.discrete_action_space:
                if action.dtype.is_floating_point or action.dtype == torch.bool:
                    raise ValueError("Discrete(5) actions must be an integer tensor")
                action, enc = action.reshape(n).to(torch.int32).contiguous(), abi.FTL_ACTION_DISCRETE
            else:
                if not action.dtype.is_floating_point:
                    raise ValueError("Box(1) actions must be a float tensor")
                action, enc = action.reshape(n).to(torch.float64).contiguous(), abi.FTL_ACTION_TURN
            self._keep_action = action
        elif action.dtype != torch.float64 or not action.is_contiguous() or tuple(action.shape) != (n, 2):
            raise ValueError("action must be a contiguous float64 [n_envs, 2] tensor on %s" % self.device)
        return action, enc

    # ------------------------------------------------------------------ episode metrics / error report
    def episode_metrics(self, clear=False):
        """f64[8] device tensor ``[episodes, sum return, sum frames, n_success, n_crash, n_low_reward, n_too_far, n_timeout]``
        over the episodes that ended since the state was created (or since the last ``clear=True`` call): what the
        reference reports at ``done`` (ENV:941-944), accumulated on the device before auto-reset wipes the counters.
        This is the vector a multi-GPU job all-reduces (``shard.reduce_metrics``).  Also refreshes ``error_report()``."""
        flags = abi.FTL_METRICS_CLEAR if clear else 0
        _lib.check(self.lib.ftl_episode_metrics(self.h, self._metrics.data_ptr(), self._errors.data_ptr(), flags, self._stream()), self.lib)
        return self._metrics

    def kernel_timing(self, enable=True):
        """Measurement hook: HIP events around every kernel of the following steps (``kernel_times``)."""
        _lib.check(self.lib.ftl_kernel_timing(self.h, 1 if enable else 0), self.lib)

    def kernel_times(self):
        """Average per-kernel duration in microseconds over the steps timed since the last call:
        ``dict(frames_us, rays_us, aux_us, regroup_us, steps)`` (frames includes the v1 tracker's kernel when the config has one; aux =
        ftl_aux_kernel of configs with row-f3 sensors; regroup = both regroup kernels, averaged over ALL steps)."""
        ms, n = (C.c_double * 4)(), C.c_int32()
        _lib.check(self.lib.ftl_kernel_times(self.h, C.byref(ms), C.byref(n)), self.lib)
        k = max(n.value, 1)
        return dict(frames_us=ms[0] / k * 1e3, rays_us=ms[1] / k * 1e3, aux_us=ms[2] / k * 1e3, regroup_us=ms[3] / k * 1e3, steps=n.value)

    def error_report(self):
        """(number of envs whose sticky error word is set, OR of the FTL_ERR_* bits) -- the conditions under which a
        reference run would have raised (or a capacity of the batched state overflowed).  The words survive reset and
        auto-reset; ``episode_metrics(clear=True)`` clears them together with the metrics records."""
        self.episode_metrics()
        n, bits = self._errors.tolist()
        return int(n), int(bits)

    def raise_on_errors(self, live=False):
        """The exception the reference would have raised (or FtlError for a capacity overflow) if any env reported one.
        ``live=True`` looks at the error words of the episodes in progress (``FTL_EI_ERROR``, cleared by reset like the sensors the
        reference's reset() rebuilds) instead of the sticky words that survive resets: what a caller that handles the exception and
        resets -- the single-env facade -- needs."""
        if live:
            w = self.state_field("env_int")[:, abi.EI_ERROR]
            bad = w != 0
            n = int(bad.sum().item())
            bits = 0
            if n:
                for v in w[bad].unique().tolist():
                    bits |= int(v)
        else:
            n, bits = self.error_report()
        if bits:
            raise error_for_bits(bits, n)

    # ------------------------------------------------------------------ views
    def laser_view(self, name):
        for l in self.cfg.lasers:
            if l.name == name:
                return self.lasers[:, l.out_offset:l.out_offset + l.history * l.width].view(self.n, l.history, l.width)
        raise KeyError(name)

    def aux_view(self, name):
        """Output block of a LaserSensor / LeaderTrackDetector_vector / _radar sensor: float32 ``[n_envs, *shape]`` with the shape
        the reference's ``scan`` returns (SEN:131-134, 381, 476)."""
        for a in self.cfg.aux:
            if a.name == name:
                return self.lasers[:, a.out_offset:a.out_offset + a.out_len].view(self.n, *a.shape)
        raise KeyError(name)

    def follower_info(self, name):
        """FollowerInfo.scan for every env (SEN:834-842): float32 [n, speed_direction_param] = (follower speed / max_speed,
        direction / 360, then ones) -- two divisions on the state, done with torch on the device."""
        k = dict(self.cfg.follower_info)[name]
        rd = self.state_field("rb_dbl").view(self.n, self.cfg.n_robots, abi.RD_COUNT)[:, 1]
        out = torch.ones(self.n, k, dtype=torch.float32, device=self.device)
        out[:, 0] = (rd[:, abi.RD_SPEED] / self.cfg.c.follower.max_speed).to(torch.float32)
        out[:, 1] = (rd[:, abi.RD_DIRECTION] / 360).to(torch.float32)
        return out

    def state_field(self, name):
        """Typed [n_envs, per_env] view of a named field of the state blob (parity tests / tracker obs)."""
        if name not in self._fields:
            off, per, dt, st = C.c_size_t(), C.c_size_t(), C.c_int32(), C.c_size_t()
            _lib.check(self.lib.ftl_state_field(self.h, name.encode(), C.byref(off), C.byref(per), C.byref(dt), C.byref(st)), self.lib)
            tdt = _DT[dt.value]
            esz = torch.empty((), dtype=tdt).element_size()
            a = self._state_off + off.value
            if per.value == 0:
                self._fields[name] = torch.empty(self.n, 0, dtype=tdt, device=self.device)
            else:      # rows of the per-env record are st bytes apart (a strided view), the long fields are dense
                span = (self.n - 1) * st.value + per.value * esz
                self._fields[name] = torch.as_strided(self.state[a:a + span + (-span) % esz].view(tdt), (self.n, per.value), (st.value // esz, 1))
        return self._fields[name]

    def tracker_obs(self, env):
        """(leader_positions_hist, corridor) of one env, as the reference returns them under the tracker key
        (sensors.py:324-325): hist f64[C,2]; corridor f64[C,2(right/left),2]."""
        ei = self.state_field("env_int")[env].cpu().numpy()
        lo, hi = int(ei[abi.EI_CORR_LO]), int(ei[abi.EI_CORR_HI])
        cap = self.cfg.c.corr_cap
        idx = torch.arange(lo, hi, device=self.device) % cap
        corr = self.state_field("corr")[env].view(cap, 2, 2)[idx].cpu().numpy()
        if self.cfg.c.has_tracker == 1:       # v1 tracker (SEN:148-229): float32 history list of its own, corridor never trimmed
            hist = self.state_field("hist1")[env].view(-1, 2)[:int(ei[abi.EI_HIST1_LEN])].cpu().numpy()
        else:
            hist = self.state_field("hist")[env].view(cap, 2)[idx].cpu().numpy()
        return hist, corr


class PipelinedVecGame:
    """A batch of N envs stepped as ``parts`` independent sub-batches, each on its own HIP stream of this process.

    Envs never interact, so part k's step t+1 depends on nothing but part k's step t.  Run that way -- no join between the parts --
    one part's ray kernel (VALU-bound, 80 registers a lane) runs beside another part's frame kernel (latency-bound, half the VALU idle),
    and the last, thinly occupied wavefronts of either are covered by the other stream's work: 250 M env-steps/s against 215 M for the
    same 65,536 envs of config B as one batch on one stream.  Every env sees exactly the arithmetic of ``VecGame``:
    the parts are ``VecGame`` batches over consecutive env ranges, their per-env random streams keyed by the GLOBAL env index and
    their auto-reset walking the scenario pool with the whole batch's stride, so results are bit-identical to one ``VecGame(N)``
    (tests/test_gpu_api.py).

    ``step(action)`` enqueues every part on its stream and returns the combined output tensors WITHOUT waiting: part k's rows are valid
    on ``stream(k)``; ``join()`` makes the current stream wait for all parts.  A caller that joins after every step re-aligns the parts
    and gets ``VecGame``'s throughput back; a training loop keeps them apart by consuming part k's rows (``rows(k)``) on ``stream(k)``
    and feeding ``step_part(k, action_k)`` -- the double-buffered sampling loop of asynchronous RL frameworks."""

    def __init__(self, n_envs, parts=2, device="cuda:0", config: GameConfig = None, policy_obs=False, **game_kwargs):
        import dataclasses
        from .shard import shard_range
        cfg = config if config is not None else make_config(**game_kwargs)
        self.cfg, self.n, self.device = cfg, int(n_envs), torch.device(device)
        if parts < 1 or parts > self.n:
            raise ValueError("parts must be in 1..n_envs")
        z = dict(device=self.device)
        L = max(cfg.lasers_len, 1)
        outs = dict(obs_num=torch.zeros(self.n, abi.FTL_OBS_NUM, dtype=torch.float32, **z), lasers=torch.zeros(self.n, L, dtype=torch.float32, **z),
                    target=torch.zeros(self.n, 2, dtype=torch.float64, **z), reward=torch.zeros(self.n, dtype=torch.float64, **z),
                    done=torch.zeros(self.n, dtype=torch.uint8, **z), status=torch.zeros(self.n, 3, dtype=torch.uint8, **z))
        sel = [l for l in cfg.lasers if l.in_policy_obs]
        if policy_obs and sel:
            hs = {l.history for l in sel}
            if len(hs) != 1:
                raise ValueError("policy_obs needs the same max_prev_obs on every sensor it concatenates (wrappers.py:207, 217 assert it)")
            outs["policy_obs"] = torch.zeros(self.n, hs.pop(), sum(l.width for l in sel), dtype=torch.float32, **z)
        self.shards = [shard_range(self.n, k, parts) for k in range(parts)]
        self.games, self.streams = [], []
        for sh in self.shards:
            ck = dataclasses.replace(cfg, c=abi.Config.from_buffer_copy(cfg.c))
            ck.c.env_id_base = cfg.c.env_id_base + sh.lo          # per-env random streams are keyed by the global env index
            g = VecGame(sh.n, device=self.device, config=ck, policy_obs=policy_obs, _outputs={k: v[sh.lo:sh.hi] for k, v in outs.items()})
            if parts > 1:
                # the parts take the role of the handle's own two-stream mode (random_frames_per_step), without its join; the envs are sorted
                # by cost when the WHOLE batch oversubscribes the device, on a staler order than a lone handle's (259 against 255 M env-steps/s)
                g.tune(two_streams=0, coscheduled_envs=self.n, regroup_every=8)
            self.games.append(g)
            self.streams.append(torch.cuda.Stream(device=self.device))
        for k, v in outs.items():
            setattr(self, k, v)
        if "policy_obs" not in outs:
            self.policy_obs = None
        self.pool = None
        self._serial = False
        self._metrics = torch.zeros(abi.FTL_N_METRICS, dtype=torch.float64, **z)
        self._stream_ptrs = [C.c_void_p(s.cuda_stream) for s in self.streams]
        self._ev = torch.cuda.Event()

    parts = property(lambda self: len(self.games))

    def close(self):
        for g in self.games:
            g.close()

    def stream(self, k):
        return self.streams[k]

    def rows(self, k):
        """(lo, hi) of part k's envs in the combined tensors."""
        return self.shards[k].lo, self.shards[k].hi

    def _on(self, k):
        """Context: part k's stream, after everything the current stream has been given so far (the action tensor's producer)."""
        cur = torch.cuda.current_stream(self.device)
        if self._serial:
            return torch.cuda.stream(cur)
        self.streams[k].wait_stream(cur)
        return torch.cuda.stream(self.streams[k])

    def join(self):
        """The current stream waits for every part (outputs of all rows valid on it afterwards)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def load_scenarios(self, pool: ScenarioPool):
        self.pool = pool
        for g in self.games:
            g.load_scenarios(pool)
            g.set_reset_window(0, pool.n, self.n)      # the auto-reset walks the pool with the WHOLE batch's stride, as VecGame(n) does

    def set_reset_window(self, base, count, stride=0):
        """``VecGame.set_reset_window`` for every part; ``stride`` 0 keeps the whole batch's n_envs."""
        for g in self.games:
            g.set_reset_window(base, count, stride if stride > 0 else self.n)

    def state_field(self, name):
        """[n_envs, per_env] COPY of a named state field over all parts (``VecGame.state_field`` gives views, part by part)."""
        self.join()
        return torch.cat([g.state_field(name) for g in self.games], 0)

    def reset(self, scen_idx=None, mask=None):
        if self.pool is None:
            raise _lib.FtlError("load_scenarios() first")
        if scen_idx is None:
            scen_idx = torch.arange(self.n, dtype=torch.int32, device=self.device) % self.pool.n
        scen_idx = torch.as_tensor(scen_idx, dtype=torch.int32, device=self.device).contiguous()
        if scen_idx.numel() != self.n:
            raise ValueError("scen_idx must have one entry per env")
        if mask is not None:
            mask = torch.as_tensor(mask, dtype=torch.uint8, device=self.device).contiguous()
        for k, (g, sh) in enumerate(zip(self.games, self.shards)):
            with self._on(k):
                g.reset(scen_idx[sh.lo:sh.hi], None if mask is None else mask[sh.lo:sh.hi])
        self.join()
        return self.obs_num, self.lasers

    def step_part(self, k, action, auto_reset=False):
        """One step of part k on its stream; ``action`` = the rows of part k (any layout ``VecGame.step`` takes)."""
        with self._on(k):
            self.games[k].step(action, auto_reset=auto_reset)
            if not self._serial:
                action.record_stream(self.streams[k])

    def step(self, action, auto_reset=False):
        """One step of every part (``action``: the whole batch's tensor, rows in env order).  Does not join -- see the class text.
        The part streams wait for what the current stream has been given so far (the producer of ``action``); nothing waits for them."""
        g0 = self.games[0]
        action, enc = g0._encode_action(action, self.n)            # checked / decoded once for the whole batch, then handed over by row range
        self._keep_action = action
        flags = abi.FTL_STEP_AUTO_RESET if auto_reset else 0
        cur = torch.cuda.current_stream(self.device)
        base, row = action.data_ptr(), action.element_size() * (2 if enc == abi.FTL_ACTION_BOX2 else 1)
        if not self._serial:
            self._ev.record(cur)
        for g, sh, stream, sptr in zip(self.games, self.shards, self.streams, self._stream_ptrs):
            if self._serial:
                sptr = C.c_void_p(cur.cuda_stream)
            else:
                stream.wait_event(self._ev)
                action.record_stream(stream)       # (the caller may drop the tensor right away: its memory must outlive the part's read)
            _lib.check(g.lib.ftl_step_encoded(g.h, base + sh.lo * row, enc, C.byref(g._out), flags, sptr), g.lib)
        return self.obs_num, self.lasers, self.reward, self.done, self.status

    def episode_metrics(self, clear=False):
        self.join()
        self._metrics.zero_()
        for g in self.games:
            self._metrics += g.episode_metrics(clear)
        return self._metrics

    def error_report(self):
        self.join()
        n, bits = 0, 0
        for g in self.games:
            a, b = g.error_report()
            n, bits = n + a, bits | b
        return n, bits

    def raise_on_errors(self, live=False):
        """``VecGame.raise_on_errors`` over all parts (joins first)."""
        self.join()
        for g in self.games:
            g.raise_on_errors(live)

    def laser_view(self, name):
        for l in self.cfg.lasers:
            if l.name == name:
                return self.lasers[:, l.out_offset:l.out_offset + l.history * l.width].view(self.n, l.history, l.width)
        raise KeyError(name)

    def aux_view(self, name):
        for a in self.cfg.aux:
            if a.name == name:
                return self.lasers[:, a.out_offset:a.out_offset + a.out_len].view(self.n, *a.shape)
        raise KeyError(name)

    def kernel_timing(self, enable=True):
        """Measurement hook.  While enabled the parts run one after the other on the CURRENT stream, so that every kernel's HIP events
        time that kernel alone (``kernel_times``: averages per LAUNCH, i.e. per part)."""
        self.join()
        torch.cuda.current_stream(self.device).synchronize()
        self._serial = bool(enable)
        for g in self.games:
            g.kernel_timing(enable)

    def kernel_times(self):
        ts = [g.kernel_times() for g in self.games]
        out = {k: sum(t[k] for t in ts) / len(ts) for k in ("frames_us", "rays_us", "aux_us", "regroup_us")}
        out["steps"] = ts[0]["steps"]
        out["launches_per_step"] = len(ts)
        return out

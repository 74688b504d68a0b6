"""Follower-relative ("Gazebo") tracker + ray sensors, batched: the host-side mirror of the reference's
``GazeboLeaderPositionsTracker_v2`` / ``GazeboCorridor_Prev_lasers_v2`` (src/arctic_gym/gazebo_utils/gazebo_tracker.py:13-297) as
``arctic_env.py:62-90, 190-211`` constructs and calls them, over N independent robots, behind the C-ABI of ``include/ftl_gazebo.h``.

PyTorch owns the device buffers; the arithmetic happens in ``libftl_hip.so`` (``ftl_gz_kernel``).  There is no CPU path."""
import ctypes as C

import torch

from . import _lib

FTL_GZ_MAX_LASERS = 2
FTL_GZ_HIST_CAP = 64


class GzLaserCfg(C.Structure):
    _fields_ = [("count", C.c_int32), ("history", C.c_int32), ("react_corridor", C.c_int32), ("react_green", C.c_int32),
                ("react_obstacles", C.c_int32), ("pad_sectors", C.c_int32), ("length", C.c_double)]


class GzConfig(C.Structure):
    _fields_ = [("n_lasers", C.c_int32), ("max_pts", C.c_int32), ("lasers", GzLaserCfg * FTL_GZ_MAX_LASERS)]


# the two sensors arctic_env.py:69-90 creates
ARCTIC_ENV_LASERS = (dict(lasers_count=12, laser_length=10, max_prev_obs=10, react_to_green_zone=True, react_to_safe_corridor=True,
                          react_to_obstacles=True, pad_sectors=False),
                     dict(lasers_count=36, laser_length=15, max_prev_obs=10, react_to_green_zone=False, react_to_safe_corridor=False,
                          react_to_obstacles=True, pad_sectors=False))


def make_gz_config(lasers=ARCTIC_ENV_LASERS, max_pts=64):
    """``lasers``: constructor kwargs of each GazeboCorridor_Prev_lasers_v2 (defaults of SEN:742-769, 873-881)."""
    if not (0 <= len(lasers) <= FTL_GZ_MAX_LASERS):
        raise NotImplementedError("at most %d ray sensors" % FTL_GZ_MAX_LASERS)
    c = GzConfig()
    c.n_lasers, c.max_pts = len(lasers), int(max_pts)
    for k, kw in enumerate(lasers):
        n = int(kw.get("lasers_count", 12))
        if n not in (12, 24, 20, 36):
            raise ValueError("Invalid number of laser beams, should be 12,24,20 or 36")      # SEN:761-762
        h = int(kw.get("max_prev_obs", 0))
        assert h > 0                                                                          # SEN:876
        lc = c.lasers[k]
        lc.count, lc.history, lc.length = n, h, float(kw.get("laser_length", 100))
        lc.react_corridor = int(bool(kw.get("react_to_safe_corridor", True)))
        lc.react_green = int(bool(kw.get("react_to_green_zone", False)))
        lc.react_obstacles = int(bool(kw.get("react_to_obstacles", False)))
        lc.pad_sectors = int(bool(kw.get("pad_sectors", True)))
    return c


class GazeboTrackerBatch:
    """N follower-relative trackers with their ray sensors on one GPU: ``step`` = one ``tracker_v2.scan`` + every ``laser.scan`` of
    arctic_env.py:190-211 for every robot, one kernel launch."""

    def __init__(self, n_envs, device="cuda:0", lasers=ARCTIC_ENV_LASERS, max_pts=64):
        self.cfg = make_gz_config(lasers, max_pts)
        self.n, self.device = int(n_envs), torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.FtlError("GazeboTrackerBatch needs a ROCm device: there is no CPU path")
        self.lib = _lib.load()
        h = C.c_void_p()
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.ftl_gz_create(C.byref(self.cfg), self.n, dev, C.byref(h)), self.lib)
        self.h = h
        nbytes = self.lib.ftl_gz_state_bytes(self.h)
        self.state = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        self._off = (-self.state.data_ptr()) % 256
        _lib.check(self.lib.ftl_gz_bind_state(self.h, self.state.data_ptr() + self._off, nbytes), self.lib)
        self.lasers_len = self.lib.ftl_gz_lasers_len(self.h)
        self.lasers = torch.zeros(self.n, max(self.lasers_len, 1), dtype=torch.float32, device=self.device)
        self.reset()

    def close(self):
        if getattr(self, "h", None):
            self.lib.ftl_gz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask=None):
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, dtype=torch.uint8, device=self.device).contiguous()
            mptr = mask.data_ptr()
        self._keep = mask
        _lib.check(self.lib.ftl_gz_reset(self.h, mptr, self._stream()), self.lib)

    def step(self, leader_pos, yaw, delta, pts1, pts2, n_pts):
        """leader_pos f64[n,2], yaw f64[n] (rad), delta f64[n,2], pts1 / pts2 f64[n,max_pts,2], n_pts i32[n] -- device tensors."""
        def chk(t, dt, shape):
            if t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise ValueError("expected a contiguous %s %s tensor on %s" % (dt, shape, self.device))
        f64, mp = torch.float64, self.cfg.max_pts
        chk(leader_pos, f64, (self.n, 2)); chk(yaw, f64, (self.n,)); chk(delta, f64, (self.n, 2))
        chk(pts1, f64, (self.n, mp, 2)); chk(pts2, f64, (self.n, mp, 2)); chk(n_pts, torch.int32, (self.n,))
        _lib.check(self.lib.ftl_gz_step(self.h, leader_pos.data_ptr(), yaw.data_ptr(), delta.data_ptr(), pts1.data_ptr(), pts2.data_ptr(),
                                        n_pts.data_ptr(), self.lasers.data_ptr(), self._stream()), self.lib)
        return self.lasers

    def laser_view(self, k):
        off = 0
        for j in range(k):
            l = self.cfg.lasers[j]
            off += l.history * l.count * (4 if l.pad_sectors else 1)
        l = self.cfg.lasers[k]
        w = l.count * (4 if l.pad_sectors else 1)
        return self.lasers[:, off:off + l.history * w].view(self.n, l.history, w)

    def _field(self, name, dtype):
        off, per, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
        _lib.check(self.lib.ftl_gz_state_field(self.h, name.encode(), C.byref(off), C.byref(per), C.byref(dt)), self.lib)
        esz = torch.empty((), dtype=dtype).element_size()
        a = self._off + off.value
        return self.state[a:a + per.value * esz * self.n].view(dtype).view(self.n, per.value)

    def tracker_state(self, env):
        """(saving_counter, leader_positions_hist f64[C,2], corridor f64[C,2,2], error bits) of one robot (GZ:172 returns the two deques)."""
        gi = self._field("gz_int", torch.int32)[env].cpu().numpy()
        hist = self._field("gz_hist", torch.float64)[env].view(-1, 2)[:int(gi[1])].cpu().numpy()
        corr = self._field("gz_corr", torch.float64)[env].view(-1, 2, 2)[:int(gi[2])].cpu().numpy()
        return int(gi[0]), hist, corr, int(gi[3])

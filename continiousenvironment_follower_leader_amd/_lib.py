"""Build + load the HIP shared library (``libftl_hip.so``) that implements ``include/ftl.h``.

The product has no CPU fallback: if the library is missing or cannot be loaded, importing the
batched env raises."""
import ctypes as C
import os
import subprocess

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SO_PATH = os.path.join(_PKG, "libftl_hip.so")
SOURCES = [os.path.join(_PKG, "csrc", "ftl_abi.hip"), os.path.join(_PKG, "csrc", "ftl_device.hpp"),
           os.path.join(_PKG, "csrc", "ftl_frames_group.hpp"), os.path.join(_PKG, "csrc", "ftl_aux.hpp"), os.path.join(_PKG, "csrc", "ftl_gazebo.hpp"), os.path.join(_ROOT, "include", "ftl_gazebo.h"), os.path.join(_PKG, "csrc", "ftl_scenario.cpp"),
           os.path.join(_ROOT, "include", "ftl.h")]
# translation units: the device code + C-ABI, and the host-only scenario generator (reset-time, no GPU code)
UNITS = [os.path.join(_PKG, "csrc", "ftl_abi.hip"), os.path.join(_PKG, "csrc", "ftl_scenario.cpp")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the arithmetic must follow the reference operation by operation (no implicit FMA)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]

_LIB = None


def build(force=False, verbose=False):
    """Compile the HIP library in-tree for gfx950 (works without a GPU: hipcc cross-compiles)."""
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in SOURCES)
    if force or stale:
        cmd = [HIPCC] + HIPCC_FLAGS + ["-pthread", "-o", SO_PATH] + UNITS
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=os.path.join(_PKG, "csrc"))
    return SO_PATH


class FtlError(RuntimeError):
    pass


def load():
    """dlopen the library and declare every entry point of include/ftl.h."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so_path = os.environ.get("FTL_LIB", SO_PATH)     # A/B builds of the same sources (tuning only)
    if not os.path.exists(so_path):
        raise FtlError("libftl_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                       "there is no CPU fallback")
    # PyTorch-ROCm bundles its own HIP runtime; it must be the one already loaded when our library is opened so
    # that both resolve to the SAME libamdhip64 (two runtimes in one process do not see the device).
    import torch  # noqa: F401
    lib = C.CDLL(so_path)
    vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
    lib.ftl_last_error.restype = C.c_char_p
    lib.ftl_create.argtypes = [C.POINTER(abi.Config), i32, i32, C.POINTER(vp)]
    lib.ftl_create.restype = C.c_int
    lib.ftl_destroy.argtypes = [vp]
    lib.ftl_destroy.restype = None
    lib.ftl_lasers_len.argtypes = [vp]
    lib.ftl_lasers_len.restype = i32
    lib.ftl_get_config.argtypes = [vp, C.POINTER(abi.Config)]
    lib.ftl_state_bytes.argtypes = [vp]
    lib.ftl_state_bytes.restype = C.c_size_t
    lib.ftl_bind_state.argtypes = [vp, vp, C.c_size_t]
    lib.ftl_state_field.argtypes = [vp, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(i32), C.POINTER(C.c_size_t)]
    lib.ftl_load_scenarios.argtypes = [vp, C.POINTER(abi.Scenarios)]
    lib.ftl_set_reset_window.argtypes = [vp, i32, i32, i32]
    lib.ftl_tune.argtypes = [vp, i32, i32]
    lib.ftl_reset.argtypes = [vp, vp, vp, C.POINTER(abi.Outputs), vp]
    lib.ftl_step.argtypes = [vp, vp, C.POINTER(abi.Outputs), u32, vp]
    lib.ftl_step_encoded.argtypes = [vp, vp, i32, C.POINTER(abi.Outputs), u32, vp]
    lib.ftl_kernel_timing.argtypes = [vp, i32]
    lib.ftl_kernel_times.argtypes = [vp, C.POINTER(C.c_double * 4), C.POINTER(i32)]
    lib.ftl_episode_metrics.argtypes = [vp, vp, vp, u32, vp]
    lib.ftl_episode_metrics.restype = C.c_int
    lib.ftl_generate_scenarios.argtypes = [C.POINTER(abi.Config), C.POINTER(abi.ScenParams), vp, i32, i32,
                                           C.POINTER(abi.Scenarios), vp]
    lib.ftl_generate_scenarios.restype = C.c_int
    lib.ftl_sizeof_scen_params.restype = C.c_size_t
    # include/ftl_gazebo.h
    lib.ftl_gz_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
    lib.ftl_gz_destroy.argtypes = [vp]
    lib.ftl_gz_destroy.restype = None
    lib.ftl_gz_state_bytes.argtypes = [vp]
    lib.ftl_gz_state_bytes.restype = C.c_size_t
    lib.ftl_gz_bind_state.argtypes = [vp, vp, C.c_size_t]
    lib.ftl_gz_lasers_len.argtypes = [vp]
    lib.ftl_gz_lasers_len.restype = i32
    lib.ftl_gz_reset.argtypes = [vp, vp, vp]
    lib.ftl_gz_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ftl_gz_state_field.argtypes = [vp, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(i32)]
    for n in ("ftl_sizeof_config", "ftl_sizeof_scenarios", "ftl_sizeof_outputs"):
        getattr(lib, n).restype = C.c_size_t
    _LIB = lib
    return lib


EXPORTS = ("ftl_create", "ftl_destroy", "ftl_lasers_len", "ftl_get_config", "ftl_state_bytes", "ftl_bind_state",
           "ftl_state_field", "ftl_load_scenarios", "ftl_set_reset_window", "ftl_tune", "ftl_reset", "ftl_step", "ftl_step_encoded", "ftl_last_error", "ftl_generate_scenarios",
           "ftl_episode_metrics", "ftl_kernel_timing", "ftl_kernel_times",
           "ftl_gz_create", "ftl_gz_destroy", "ftl_gz_state_bytes", "ftl_gz_bind_state", "ftl_gz_lasers_len", "ftl_gz_reset", "ftl_gz_step",
           "ftl_gz_state_field")


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        msg = lib.ftl_last_error().decode("utf-8", "replace")
        if rc == abi.FTL_E_INVALID:
            raise ValueError(msg)
        if rc == abi.FTL_E_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise FtlError("ftl error %d: %s" % (rc, msg))

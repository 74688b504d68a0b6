"""C-ABI checks that need no GPU: the shared library loads, exports every symbol include/ftl.h declares, the ctypes
mirror has the C struct sizes, and the validation / error paths that never touch the device behave."""
import ctypes as C
import os
import re

import pytest

from continiousenvironment_follower_leader_amd import _lib, abi, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ftl.h")).read() + open(os.path.join(ROOT, "include", "ftl_gazebo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    inline = set(re.findall(r"static\s+inline\s+[\w\s]+?\b(ftl_\w+)\s*\(", src))      # header-only helpers, not exports
    return sorted(set(re.findall(r"\b(ftl_[a-z_]+)\s*\(", src)) - inline)


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert set(_lib.EXPORTS) <= set(syms)
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(lib, s), "include/ftl.h declares %s but libftl_hip.so does not export it" % s


def test_struct_sizes_match(lib):
    assert lib.ftl_sizeof_config() == C.sizeof(abi.Config)
    assert lib.ftl_sizeof_scenarios() == C.sizeof(abi.Scenarios)
    assert lib.ftl_sizeof_outputs() == C.sizeof(abi.Outputs)


def test_create_validates_without_touching_the_device(lib):
    cfg = make_config(bear_number=1)
    h = C.c_void_p()
    assert lib.ftl_create(C.byref(cfg.c), 4, -1, C.byref(h)) == abi.FTL_E_INVALID      # no CPU path
    assert b"no CPU path" in lib.ftl_last_error()
    assert lib.ftl_create(C.byref(cfg.c), 0, 0, C.byref(h)) == abi.FTL_E_INVALID
    bad = make_config(bear_number=1)
    bad.c.abi_version = 99
    assert lib.ftl_create(C.byref(bad.c), 4, 0, C.byref(h)) == abi.FTL_E_INVALID
    bad = make_config(bear_number=1)
    bad.c.n_bears = 9
    assert lib.ftl_create(C.byref(bad.c), 4, 0, C.byref(h)) == abi.FTL_E_INVALID
    # a valid create is pure host work: layout + config freeze
    assert lib.ftl_create(C.byref(cfg.c), 4, 0, C.byref(h)) == 0
    try:
        assert lib.ftl_state_bytes(h) > 0 and lib.ftl_lasers_len(h) == 0
        off, per, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
        for name, want in (("rb_pos", 1), ("rb_dbl", 2), ("env_int", 0), ("traj", 1), ("corr", 2), ("snap_win", 0)):
            assert lib.ftl_state_field(h, name.encode(), C.byref(off), C.byref(per), C.byref(dt)) == 0
            assert dt.value == want and off.value % 256 == 0 and per.value > 0
        assert lib.ftl_state_field(h, b"nope", C.byref(off), C.byref(per), C.byref(dt)) == abi.FTL_E_INVALID
        # call order is enforced before anything is launched
        out = abi.Outputs()
        assert lib.ftl_step(h, C.c_void_p(8), C.byref(out), 0, None) == abi.FTL_E_STATE
        assert lib.ftl_reset(h, C.c_void_p(8), None, C.byref(out), None) == abi.FTL_E_STATE
    finally:
        lib.ftl_destroy(h)


def test_laser_offsets_are_frozen_by_create(lib):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import config_for, load_episode
    z, meta = load_episode("B_s1_chase")
    cfg = config_for(meta)
    h = C.c_void_p()
    assert lib.ftl_create(C.byref(cfg.c), 2, 0, C.byref(h)) == 0
    got = abi.Config()
    assert lib.ftl_get_config(h, C.byref(got)) == 0
    assert lib.ftl_lasers_len(h) == 5 * 12 + 5 * 24
    assert [got.lasers[k].out_offset for k in range(2)] == [0, 60]
    lib.ftl_destroy(h)

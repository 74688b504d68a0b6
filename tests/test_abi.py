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
        off, per, dt, st = C.c_size_t(), C.c_size_t(), C.c_int32(), C.c_size_t()
        rec_stride, spans = None, []
        for name, want, in_record in (("rb_pos", 1, True), ("rb_dbl", 2, True), ("rb_int", 0, True), ("env_int", 0, True), ("env_dbl", 2, True), ("fol_cs", 2, True),
                                      ("snap_win", 0, True), ("snap_rects", 0, True), ("traj", 1, False), ("corr", 2, False), ("ep_stats", 2, False)):
            assert lib.ftl_state_field(h, name.encode(), C.byref(off), C.byref(per), C.byref(dt), C.byref(st)) == 0
            esz = (4, 4, 8)[dt.value]
            assert dt.value == want and per.value > 0 and off.value % 16 == 0
            if in_record:       # the small fields share one per-env record: a common stride (whole cache lines), disjoint byte ranges inside it
                rec_stride = rec_stride or st.value
                assert st.value == rec_stride and rec_stride % 128 == 0 and off.value + per.value * esz <= rec_stride
                spans.append((off.value, off.value + per.value * esz))
            else:               # the long fields are dense arrays behind the records
                assert st.value == per.value * esz and off.value % 256 == 0 and off.value >= 4 * rec_stride
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
        assert lib.ftl_state_field(h, b"nope", C.byref(off), C.byref(per), C.byref(dt), C.byref(st)) == abi.FTL_E_INVALID
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

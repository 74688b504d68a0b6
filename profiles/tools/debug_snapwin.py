import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_api import _pool_cfg, _vec, _actions
from continiousenvironment_follower_leader_amd import abi
n = 96
cfg = _pool_cfg(max_steps=60, warm_start=10)
a, b = _vec(n, cfg), _vec(n, cfg)
idx = torch.arange(n, dtype=torch.int32)
a.reset(idx); b.reset(idx)
for f in ("snap_win", "snap_rects", "env_int"):
    print("after reset", f, torch.equal(a.state_field(f), b.state_field(f)))
act = _actions(cfg, n, 0)
a.step(act, auto_reset=True); b.step(act, auto_reset=False)
sa, sb = a.state_field("snap_win").cpu(), b.state_field("snap_win").cpu()
d = (sa != sb).nonzero()
print("n diff", len(d), d[:10].tolist())
for e, j in d[:6].tolist():
    print(e, j, sa[e].tolist(), sb[e].tolist(), a.state_field("env_int")[e, [abi.EI_SNAP_COUNT, abi.EI_CORR_LO, abi.EI_CORR_HI, abi.EI_TRK_COUNTER, abi.EI_DONE]].tolist(), b.state_field("env_int")[e, [abi.EI_SNAP_COUNT, abi.EI_CORR_LO, abi.EI_CORR_HI, abi.EI_TRK_COUNTER, abi.EI_DONE]].tolist())

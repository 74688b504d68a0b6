import sys, os, json, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from golden_util import GOLDEN, config_for
from continiousenvironment_follower_leader_amd.vec_game import ScenarioPool, VecGame
from continiousenvironment_follower_leader_amd import abi
import bench
z = np.load(GOLDEN + "/pool_B.npz"); meta = json.loads(str(z["meta"]))
cfg = config_for(dict(kwargs=meta["kwargs"], post=None), scen_route_len=int(z["route_len"].max()))
n = 8192
env = VecGame(n, device="cuda:0", config=cfg); env.load_scenarios(ScenarioPool.from_npz(cfg, GOLDEN + "/pool_B.npz", "cuda:0"))
idx = (torch.arange(n, dtype=torch.int64)) % env.pool.n
env.reset(idx.to(torch.int32)); acts = bench.make_actions(cfg, n, 16, 0, "cuda:0"); torch.cuda.synchronize()
ei = env.state_field("env_int")
prev = None
for t in range(110):
    env.step(acts[t % 16], auto_reset=False)
    e = ei[:, abi.EI_ERROR].cpu().numpy()
    cnt = (e >> 16) & 0xffff; why = (e >> 8) & 0xff
    if prev is not None and (t % 10 == 9 or 86 <= t <= 95):
        d = cnt - prev
        print(t, "envs with exact this step:", int((d > 0).sum()), "total exact calls:", int(d.sum()), "why hist:", np.bincount(why[d > 0], minlength=4).tolist(),
              "steps-since-reset mean", ei[:, abi.EI_STEP_COUNT].float().mean().item(), "green mean", ei[:, abi.EI_GREEN_COUNT].float().mean().item())
    prev = cnt.copy()
d = cnt - prev if False else None
e = ei.cpu().numpy()
cnt2 = (e[:, abi.EI_ERROR] >> 16) & 0xffff
env.step(acts[0], auto_reset=False)
e2 = ei.cpu().numpy(); cnt3 = (e2[:, abi.EI_ERROR] >> 16) & 0xffff
bad = np.nonzero(cnt3 - cnt2 > 0)[0]
print("bad envs", len(bad), bad[:8])
rd = env.state_field("rb_dbl").cpu().numpy().reshape(n, 3, 5)
ed = env.state_field("env_dbl").cpu().numpy()
for b in bad[:6]:
    print(b, "done", e2[b, abi.EI_DONE], "lfin", e2[b, abi.EI_LEADER_FINISHED], "crash", e2[b, abi.EI_CRASH], "traj", e2[b, abi.EI_TRAJ_LEN], "green", e2[b, abi.EI_GREEN_COUNT], "W", ed[b, abi.ED_GREEN_W], "leader speed", rd[b, 0, 1], "rot", rd[b,0,2], "target id", e2[b, abi.EI_TARGET_ID])
    tr = env.state_field("traj")[b].view(-1, 2).cpu().numpy()[: e2[b, abi.EI_TRAJ_LEN]]
    seg = np.linalg.norm(np.diff(tr.astype(np.float64), axis=0), axis=1)
    print("   newest segs", np.round(seg[-4:], 4), "old-end segs", np.round(seg[-e2[b, abi.EI_GREEN_COUNT]-3:-e2[b, abi.EI_GREEN_COUNT]+2], 4))

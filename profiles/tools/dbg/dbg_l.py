import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from test_gpu_configs import _cfg_pool, _vec, _actions
from oracle_batch import OracleBatch, pool_scenarios
from continiousenvironment_follower_leader_amd import abi
n=192
cfg,pool=_cfg_pool("L_s2_chase",96)
env=_vec(n,cfg,pool); scen=pool_scenarios(pool); idx=np.arange(n)%pool.n
env.reset(torch.from_numpy(idx.astype(np.int32)))
ora=OracleBatch(cfg,n); ora.reset(scen,idx)
a=_actions(cfg,n,0,"random",seed=13)
env.step(torch.tensor(a,dtype=torch.float64,device="cuda:0")); ora.step(a)
las=env.lasers.cpu().numpy()
bad=np.abs(las-ora.lasers)>1e-3
print("bad envs", np.unique(np.argwhere(bad)[:,0])[:40], bad.sum())
for aux in cfg.aux:
    b=bad[:,aux.out_offset:aux.out_offset+aux.out_len]
    print(aux.name, b.sum(), np.unique(np.argwhere(b)[:,0])[:10])
e=int(np.argwhere(bad)[0,0])
ei=env.state_field("env_int")[e].cpu().numpy()
print("env",e,"HW0",ei[abi.EI_HW0_LO],ei[abi.EI_HW0_HI],"CORR",ei[abi.EI_CORR_LO],ei[abi.EI_CORR_HI],"seed_end",ei[abi.EI_SEED_END],"trk",ei[abi.EI_TRK_COUNTER], "done", ei[abi.EI_DONE])
A=cfg.aux[0]
print(las[e,A.out_offset:A.out_offset+A.out_len]); print(ora.lasers[e,A.out_offset:A.out_offset+A.out_len])
A=cfg.aux[-1]
print(las[e,A.out_offset:A.out_offset+A.out_len]); print(ora.lasers[e,A.out_offset:A.out_offset+A.out_len])

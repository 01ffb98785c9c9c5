import numpy as np, sys
sys.path.insert(0,'.')
from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
from mjmpc_amd.models.reacher7dof import reacher7dof_raw
for dt in ("f64","f32"):
    eng = ArmRolloutEngine(reacher7dof_raw(), dtype=dt)
    rs=np.random.RandomState(3)
    q0=np.array([0.3,-1.0,0.2,1.9,0.1,0.6,0.0]); 
    eng.set_env_state(dict(qp=q0, qv=0.5*rs.standard_normal(7), target_pos=np.array([0.1,0.1,0.1])))
    P,H=256,16
    npdt=np.float32 if dt=="f32" else np.float64
    noise=(2.0*rs.standard_normal((P,H,7))).astype(npdt); mean=np.zeros((H,7))
    def run(nz):
        out=eng.rollout_device(nz.shape[0],H,mean,nz)
        return out[0].cpu().numpy().copy()
    c=run(noise); perm=rs.permutation(P); cp=run(noise[perm])
    print(dt,"perm equal",np.array_equal(cp,c[perm]), "max diff", np.abs(cp-c[perm]).max())
    for k in (0,5,100):
        c1=run(noise[k:k+1]); print("  alone",k,np.array_equal(c1[0],c[k]), np.abs(c1[0]-c[k]).max())

import sys, importlib, time, os, subprocess
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
def rand_scalars(n, seed):
    rng=np.random.default_rng(seed)
    s=rng.integers(0,2**63,size=(n,4),dtype=np.uint64)*2+rng.integers(0,2,size=(n,4),dtype=np.uint64)
    s[:,3]&=np.uint64((1<<39)-1)
    return s
ln=int(sys.argv[1]); n=1<<ln
k=rand_scalars(n,1); s=rand_scalars(n,2)
xy,inf=dvp.curve.point_scalar_mul_gen_batch(k)
d_s=torch.from_numpy(s.view(np.int64)).cuda(); d_b=torch.from_numpy(xy.view(np.int64)).cuda()
d_out=torch.zeros(8,dtype=torch.int64,device='cuda'); d_inf=torch.zeros(2,dtype=torch.int32,device='cuda')
st=torch.cuda.current_stream().cuda_stream
ref=None
for cfg in sys.argv[2:]:
    c,K=cfg.split(',')
    dvp.lib.dvp_tune_set(b'DVP_MSM_C',int(c)); dvp.lib.dvp_tune_set(b'DVP_MSM_K',int(K))
    for it in range(2):
        dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(),d_b.data_ptr(),0,n,d_out.data_ptr(),d_inf.data_ptr(),st)
    torch.cuda.synchronize()
    if ref is None: ref=d_out.clone()
    assert (ref==d_out).all()
    reps=3; t0=time.time()
    for it in range(reps):
        dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(),d_b.data_ptr(),0,n,d_out.data_ptr(),d_inf.data_ptr(),st)
    torch.cuda.synchronize(); dt=(time.time()-t0)/reps
    print("n=2^%d c=%s K=%s  %.3f ms  %.2f Mpoints/s"%(ln,c,K,dt*1e3,n/dt/1e6),flush=True)

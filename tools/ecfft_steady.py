"""steady-state enter / exit at 2^log_n for a kernel trace: the bootstrap (exit tables) runs first; then a marker (a 1-element
batch inversion), ONE enter, a marker, ONE exit, a marker"""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
t = dvp.ec_fft.FFTree(n)
rng = np.random.default_rng(1)
c = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64); c[:, 3] &= np.uint64((1 << 38) - 1)
d_in = torch.from_numpy(c.view(np.int64)).cuda(); d_ev = torch.empty_like(d_in); d_back = torch.empty_like(d_in)
one = torch.ones(4, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def marker():
    dvp.check(dvp.lib.dvp_fr_batch_inverse_dev(one.data_ptr(), 1, st))
for _ in range(2):
    t.enter_dev(d_in.data_ptr(), d_ev.data_ptr(), st)
    t.exit_dev(d_ev.data_ptr(), d_back.data_ptr(), st)
torch.cuda.synchronize()
marker(); t.enter_dev(d_in.data_ptr(), d_ev.data_ptr(), st)
marker(); t.exit_dev(d_ev.data_ptr(), d_back.data_ptr(), st)
marker(); torch.cuda.synchronize()
assert (d_back == d_in).all()
print("ok")

"""enter / exit / extend timings on device-resident vectors (BASELINE config #3: 2^20 coefficients)."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
t = dvp.ec_fft.FFTree(n)
rng = np.random.default_rng(1)
c = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64); c[:, 3] &= np.uint64((1 << 38) - 1)
d_in = torch.from_numpy(c.view(np.int64)).cuda(); d_ev = torch.empty_like(d_in); d_back = torch.empty_like(d_in)
st = torch.cuda.current_stream().cuda_stream
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e3
t_enter = timeit(lambda: t.enter_dev(d_in.data_ptr(), d_ev.data_ptr(), st))
t0 = time.time(); t.exit_dev(d_ev.data_ptr(), d_back.data_ptr(), st); torch.cuda.synchronize(); t_first = (time.time() - t0) * 1e3
t_exit = timeit(lambda: t.exit_dev(d_ev.data_ptr(), d_back.data_ptr(), st))
assert (d_back == d_in).all()
T2 = dvp.ec_fft.FFTree(2 * n)
x = d_in.reshape(1, n, 4).repeat(4, 1, 1).contiguous(); y = torch.empty_like(x)
t_ext = timeit(lambda: T2.extend_dev(x.data_ptr(), 4, y.data_ptr(), st))
print(f"n=2^{log_n}: enter {t_enter:.2f} ms, exit {t_exit:.2f} ms (first call incl. table bootstrap {t_first:.0f} ms), extend x4 (m=2^{log_n}) {t_ext:.2f} ms; round trip exact")

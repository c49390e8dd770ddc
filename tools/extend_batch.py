"""extend of `batch` vectors of 2^lg values, time per vector against the batch size (is one chip-full of the fused kernel as fast
per element as eight?): python tools/extend_batch.py [lg]"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n = 1 << lg
T = dvp.ec_fft.FFTree(2 * n)
rng = np.random.default_rng(1)
st = torch.cuda.current_stream().cuda_stream
for batch in (1, 2, 3, 4, 8, 16):
    c = rng.integers(0, 2**62, size=(batch, n, 4), dtype=np.uint64); c[..., 3] &= np.uint64((1 << 38) - 1)
    x = torch.from_numpy(c.view(np.int64)).cuda(); y = torch.empty_like(x)
    for _ in range(2): T.extend_dev(x.data_ptr(), batch, y.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): T.extend_dev(x.data_ptr(), batch, y.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    print(f"extend 2^{lg} x{batch}: {dt:.3f} ms = {dt / batch * 1e3:.1f} us per vector", flush=True)

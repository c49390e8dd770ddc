"""Does HBM traffic lower the rate of the GF(2^233) multiplier?  Runs dvp_ubench_gf_mul alone, then again while a second
stream streams tensor copies (pure HBM reads + writes) -- the experiment behind DESIGN.md section 9."""
import ctypes as C, importlib, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
dvp = importlib.import_module("dv-pari_amd")

def mul_rate(reps=3000):
    r = C.c_double(0)
    dvp.check(dvp.lib.dvp_ubench_gf_mul(reps, C.byref(r)), "ubench")
    return r.value / 1e9

print("multiplier alone: %.2f G products/s" % mul_rate())
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
side = torch.cuda.Stream()
for pause in (0.0, 0.0004, 0.0008, 0.0016):  # host sleep between 1 GB copies: throttles the copy stream
    stop = False
    copied = [0]
    def pump():
        with torch.cuda.stream(side):
            while not stop:
                for _ in range(8):
                    b.copy_(a)
                    if pause:
                        side.synchronize(); time.sleep(pause)
                side.synchronize()
                copied[0] += 8
    t = threading.Thread(target=pump); t.start()
    time.sleep(0.5)
    c0, t0 = copied[0], time.time()
    rates = [mul_rate() for _ in range(3)]
    dt = time.time() - t0
    gbs = (copied[0] - c0) * 2 * (1 << 30) / dt / 1e9
    stop = True; t.join()
    print("multiplier with copies on a second stream: %s G products/s; copies moved %.0f GB/s meanwhile" % (["%.2f" % r for r in rates], gbs))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(32): b.copy_(a)
torch.cuda.synchronize()
print("copies alone: %.0f GB/s" % (32 * 2 * (1 << 30) / (time.time() - t0) / 1e9))

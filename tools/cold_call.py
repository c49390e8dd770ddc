#!/usr/bin/env python3
"""The COLD call of the drop-in entry, from a fresh process: Proof::prove(cache_dir, public, private) (src/proving.rs:426-688) reads the
R1CS dump and the five SRS files, decodes 6 m points and -- here -- builds the fixed-base tables on its first call for a directory.
    python tools/cold_call.py <cache_dir> <witness.npy> [n_public]
<witness.npy>: the assignment [1, public.., private..] as an (n_wires, 4) uint64 array (written by bench.py).  Prints one JSON line:
seconds of the first dvp_prove_cache_dir call (file reads from the page cache the parent just filled, decodes, table build, proof), of
the second (warm) call, and the proof bytes (hex) for the parent to compare."""
import importlib, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
t_import = time.perf_counter()
import torch  # noqa: F401  (first import of a fresh process: not part of the call)
dvp = importlib.import_module("dv-pari_amd")
t_import = time.perf_counter() - t_import
cache, wfile = sys.argv[1], sys.argv[2]
n_pub = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w = np.load(wfile)
pub, prv = np.ascontiguousarray(w[1:1 + n_pub]), np.ascontiguousarray(w[1 + n_pub:])
torch.cuda.init()
torch.zeros(1, device="cuda")  # the HIP context exists before the clock starts (a host that proves has one)
torch.cuda.synchronize()
t0 = time.perf_counter()
p1 = dvp.proving.Proof.prove(cache, pub, prv)
t1 = time.perf_counter()
p2 = dvp.proving.Proof.prove(cache, pub, prv)
t2 = time.perf_counter()
assert p1 == p2
print(json.dumps({"cold_call_s": t1 - t0, "second_call_s": t2 - t1, "import_s": t_import, "proof_hex": p1.to_bytes().hex()}), flush=True)

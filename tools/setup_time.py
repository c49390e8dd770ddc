"""Seconds of SRS::verifier_runs_setup through ONE library call (dvp_setup_cache_dir, csrc/setup.hip) on an SP1-like sparse synthetic
circuit of 2^log_m constraints (the reference quotes "10 mins" for the SRS at 2^23 and "2 hrs+" for its vanishing polynomial,
src/artifacts.rs:92-98):   python tools/setup_time.py [log_m ...]     (writes into a temporary directory, removed afterwards)"""
import importlib, os, shutil, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
nat = importlib.import_module("dv-pari_amd._native")
td = (0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
t, d, e = (dvp.fr.limbs(x) for x in td)
for log_m in [int(x) for x in sys.argv[1:]] or [20]:
    t0 = time.time()
    inst, pub, prv = dvp.gnark_r1cs.synthetic_sparse_fast(log_m, max_terms=8)
    tmp = tempfile.mkdtemp(prefix="dvp_setup_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        path = os.path.join(tmp, dvp.artifacts.R1CS_CONSTRAINTS_FILE)
        inst.write_dump_file(path)
        nnz = [int(x.row_ptr[-1]) for x in (inst.l, inst.r, inst.o)]
        print(f"2^{log_m}: rows {inst.n_rows} wires {inst.n_wires} terms {sum(nnz)} dump {os.path.getsize(path) / 1e6:.0f} MB (generated in {time.time() - t0:.1f} s)", flush=True)
        for pre in (0, 1):
            for rep in range(2):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                dvp.check(dvp.lib.dvp_setup_cache_dir(nat.ptr(t), nat.ptr(d), nat.ptr(e), os.fsencode(tmp), len(pub), pre), "dvp_setup_cache_dir")
                torch.cuda.synchronize(); dt = time.perf_counter() - t1
                print(f"  dvp_setup_cache_dir(write_precomputes={pre}) call {rep + 1}: {dt:.2f} s", flush=True)
        # prover_prepares_precomputes (src/proving.rs:225-325) on the directory the setup left: first with bar_wts / z_vals2inv / tree2n
        # missing (generated + written), then validating what it finds (the reference quotes "10-20 mins" for the trees at 2^23)
        for n in (dvp.artifacts.BAR_WTS, dvp.artifacts.Z_VALS2_INV):
            os.remove(os.path.join(tmp, n))
        for label, validate in (("generate + write tree2n, bar_wts, z_vals2inv", False), ("validate everything found", True)):
            t1 = time.perf_counter()
            rep_bits = dvp.proving.prover_prepares_precomputes(tmp, validate_precompute=validate)
            print(f"  dvp_prover_prepares_precomputes ({label}): {time.perf_counter() - t1:.2f} s (report 0x{rep_bits:x})", flush=True)
        sizes = {n: os.path.getsize(os.path.join(tmp, n)) for n in sorted(os.listdir(tmp))}
        print("  files (MB):", {k: round(v / 1e6, 1) for k, v in sizes.items()}, flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

#!/usr/bin/env python3
"""Turns gpurun_out/refresh/ (tools/refresh_profiles.sh) into the committed summaries under profiles/.
usage: python tools/digest_profiles.py rNN"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    assert hits, pattern
    return hits[0]


def last_json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, f"{tag}_bench_prove2p20.json"))  # the ONE line (<= 6 KB)
shutil.copy(os.path.join(SRC, "bench_detail.json"), os.path.join(DST, f"{tag}_bench_detail.json"))  # its sidecar: everything the run measured
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, f"{tag}_bench_prove2p20_under_rocprof.json"))
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_prove2p20_kernel_stats.csv"))
try:
    shutil.copy(one("stats_default/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_prove2p20_default_cmd_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "bench_default_under_rocprof.json"), os.path.join(DST, f"{tag}_bench_prove2p20_default_cmd_under_rocprof.json"))
except Exception as e:  # older refresh runs have no such pass
    print("no default-command stats pass:", e)
shutil.copy(one("msm/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_msm_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "msm_bench.log"), os.path.join(DST, f"{tag}_msm_bench.log"))
try:
    shutil.copy(one("msm16/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_msm_2p16_kernel_stats.csv"))
except AssertionError:
    pass


try:  # the wave-level trace of the pair rounds (tools/wave_trace.py), when the refresh ran it
    shutil.copy(os.path.join(SRC, "wave_trace.json"), os.path.join(DST, f"{tag}_wave_trace_pair_rounds.json"))
except Exception as e:
    print("no wave trace in this refresh:", e)


def round0_dispatches(dirname):
    """per dispatch {counter: value, ms}: the k_affine_round<true> launches, i.e. the FIRST pair round of each MSM (the
    dominant kernel; the later rounds, k_affine_round<false>, run the same code on compacted inputs)"""
    rows = list(csv.DictReader(open(one(f"{dirname}/**/*counter_collection.csv"))))
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    # bench.py's untimed legs (microbenchmarks, stand-alone MSMs, the second table flavour) come after the proofs: only the
    # launches before the first microbenchmark dispatch belong to the timed configuration
    first_extra = min([i for i, d in disp.items() if "k_ubench" in d["name"]], default=None)
    out = [d for i, d in disp.items() if "k_affine_round<true" in d["name"] and (first_extra is None or i < first_extra)]
    assert out, dirname
    return out


def avg(ds, key):
    return sum(d[key] for d in ds) / len(ds)


fetch = round0_dispatches("pmc_fetch")
write = round0_dispatches("pmc_write")
b = json.load(open(os.path.join(SRC, "bench_detail.json")))  # (the line itself carries the headline only since round 6)
alg = b["roofline"]["algorithmic_bytes_per_launch"]
f_kb, w_kb = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE")
traffic = {
    "kernel": "dvp::k_affine_round<true>, first pair round of each MSM",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, counters + --kernel-trace only) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
    "launches": len(fetch),
    "avg_FETCH_SIZE_KB": f_kb,
    "avg_WRITE_SIZE_KB": w_kb,
    "avg_duration_ms_under_pmc": avg(fetch, "ms"),
    "traffic_bytes_per_launch_raw": (f_kb + w_kb) * 1024.0,
    "traffic_bytes_per_launch_fetch_x2": (2 * f_kb + w_kb) * 1024.0,
    "algorithmic_bytes_per_launch": alg,
    "conventions": "raw = FETCH_SIZE + WRITE_SIZE as reported; fetch_x2 = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of "
                   "MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B; measured there on wide coalesced streams -- this kernel's "
                   "reads are 64-B gathers of the bases plus coalesced descriptor / prefix streams, so the true figure lies between the two). "
                   "Expected per pair addition: two 64-B bases gathered in pass 1 and again in pass 2 (HBM or Infinity-Cache hits, which the "
                   "counter includes), 8 B descriptor x 2, 32 B prefix product written + read, 64 B output = ~400 B, against 96 B x 2/W "
                   "algorithmic bytes: the gathers are the price of sharing one inversion among ~36 additions.",
}
# request-level pass (TCC_EA0_RDREQ / _32B / HIT / MISS), when the counters exist on this ROCm build: turns the raw .. x2 spread
# into one figure -- read bytes = 32 B x (32-byte requests) + 64 B x (the others), unless the guide's 128-byte tally applies
try:
    tcc = round0_dispatches("pmc_tcc")
    rd, rd32, rd64, rd128 = (avg(tcc, k) for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
    read_bytes = 32.0 * rd32 + 64.0 * rd64 + 128.0 * rd128
    traffic["tcc_requests"] = {
        "source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace, and a second pass with "
                  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum (separate passes of the same command, counters + kernel trace only)",
        "TCC_EA0_RDREQ": rd, "TCC_EA0_RDREQ_32B": rd32, "TCC_EA0_RDREQ_64B": rd64, "TCC_EA0_RDREQ_128B": rd128,
        "read_bytes_by_request_size": read_bytes,
        "read_requests_per_addition": rd / b["roofline"]["work_model"]["additions_per_launch"],
        "avg_duration_ms_under_pmc": avg(tcc, "ms"),
    }
    try:
        t2 = round0_dispatches("pmc_tcc2")
        hit, miss = avg(t2, "TCC_HIT_sum"), avg(t2, "TCC_MISS_sum")
        traffic["tcc_requests"].update({"TCC_HIT": hit, "TCC_MISS": miss, "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
                                        "TCC_EA0_RDREQ_DRAM": avg(t2, "TCC_EA0_RDREQ_DRAM_sum"), "TCC_REQ": avg(t2, "TCC_REQ_sum")})
    except Exception as e2:
        traffic["tcc_requests"]["second_pass_unavailable"] = str(e2)[:200]
    traffic["traffic_bytes_per_launch_best"] = read_bytes + w_kb * 1024.0
    traffic["reading"] = ("read bytes by request size (32 B x %.3g + 64 B x %.3g + 128 B x %.3g) = %.3g B against FETCH_SIZE x 1024 = %.3g B: ratio %.2f -- "
                          "1.0 means FETCH_SIZE already counts this kernel's requests at their size, 2.0 that the guide's x2 correction applies"
                          % (rd32, rd64, rd128, read_bytes, f_kb * 1024.0, read_bytes / (f_kb * 1024.0)))
except Exception as e:  # counters missing on this build: keep the two-convention spread
    traffic["tcc_requests"] = {"unavailable": str(e)[:200]}
json.dump(traffic, open(os.path.join(DST, f"{tag}_pmc_traffic_k_affine_round0.json"), "w"), indent=1)

sq = round0_dispatches("pmc_sq")
n_cu, n_simd = 256, 1024
cyc = avg(sq, "GRBM_GUI_ACTIVE") / 8.0  # the counter sums the 8 XCDs
valu, lds = avg(sq, "SQ_INSTS_VALU"), avg(sq, "SQ_INSTS_LDS")
issue = {
    "kernel": traffic["kernel"],
    "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace "
              "-- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (averages over the first-round launches)",
    "launches": len(sq),
    "avg_duration_ms_under_pmc": avg(sq, "ms"),
    "counters_per_launch": {k: avg(sq, k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE")},
    "shader_cycles_per_launch": cyc,
    "effective_clock_ghz": cyc / (avg(sq, "ms") * 1e-3) / 1e9,
    "valu_insts_per_simd_cycle": valu / n_simd / cyc,
    "valu_issue_ceiling_insts_per_simd_cycle": "0.5 for full-rate ops (xor/and/bitop3: 2 cycles per wave64 on a SIMD-32), 0.25 for the half-rate ones "
                                                "(shifts, v_alignbit: 309 of the multiplier's 862 VALU instructions since round 5, 461 of 1 018 before) -> "
                                                "862 / (553 x 2 + 309 x 4) = 0.37 for this mix (rounds 2-4: 0.31)",
    "valu_issue_frac_of_mix_ceiling": (valu / n_simd / cyc) / 0.368,
    "lds_busy_frac": lds * 5.34 / n_cu / cyc,
    "lds_busy_note": "SQ_INSTS_LDS x 5.34 LDS cycles per wave-instruction (per product: 120 ds_read_b128 x 4 + 21 ds_write_b128 x 13 cycles over 141 "
                     "instructions, MI355X_MICROARCH.md LDS table) / (256 CUs x shader cycles)",
    "resident_wave_frac": avg(sq, "SQ_WAVE_CYCLES") * 4.0 / (cyc * n_cu * 12),
    "resident_wave_note": "SQ_WAVE_CYCLES (quad-cycles) x 4 / (shader cycles x 256 CUs x 12 waves per CU)",
    "valu_insts_per_pair_addition": valu * 64.0 / b["roofline"]["work_model"]["additions_per_launch"],
}
# the multiplier microbenchmark (k_ubench_mul, run by bench.py outside its timed loop) under the same counters: the rate
# bench.py's work model divides by is measured at a HIGHER clock than the pair rounds hold (no HBM traffic, less power)
rows = list(csv.DictReader(open(one("pmc_sq/**/*counter_collection.csv"))))
ub = collections.defaultdict(dict)
for r in rows:
    if "k_ubench_mul" in r["Kernel_Name"]:
        ub[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        ub[r["Dispatch_Id"]]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
if ub:
    u = list(ub.values())
    ucyc = avg(u, "GRBM_GUI_ACTIVE") / 8.0
    issue["multiplier_microbenchmark_same_counters"] = {
        "effective_clock_ghz": ucyc / (avg(u, "ms") * 1e-3) / 1e9,
        "valu_insts_per_simd_cycle": avg(u, "SQ_INSTS_VALU") / n_simd / ucyc,
        "resident_wave_frac": avg(u, "SQ_WAVE_CYCLES") * 4.0 / (ucyc * n_cu * 12),
        "reading": "per CYCLE the pair round issues valu_insts_per_simd_cycle / this figure of the microbenchmark's VALU rate, and most of that "
                   "difference is the resident-wave fraction; the rest of the gap between the kernel's product rate and the microbenchmark's is "
                   "the CLOCK: the chip holds a lower clock in the pair rounds (VALU + LDS + 2-4 TB/s of HBM gathers) than in the microbenchmark",
    }
json.dump(issue, open(os.path.join(DST, f"{tag}_pmc_sq_k_affine_round0.json"), "w"), indent=1)
print("bench:", b["value"], "constraints/s", b["ms_per_step"], "ms; round0", b["roofline"]["avg_launch_ms"], "ms/launch; work_model frac",
      b["roofline"]["work_model"]["frac"])
print("traffic per launch: raw %.3f GB, fetch x2 %.3f GB, algorithmic %.3f GB" % (traffic["traffic_bytes_per_launch_raw"] / 1e9,
                                                                              traffic["traffic_bytes_per_launch_fetch_x2"] / 1e9, alg / 1e9))
print("issue:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in issue.items() if k.endswith("frac") or k.endswith("cycle") or k.endswith("ghz") or k.endswith("ceiling")})
for r in list(csv.DictReader(open(os.path.join(DST, f"{tag}_bench_prove2p20_kernel_stats.csv"))))[:8]:
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg")

# ---- round 5: every pair round (later rounds included), the request-size experiment, the other configs, and the stamp ----
import subprocess
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "digest_rounds.py"), "refresh", tag, "by_round"], check=False)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "digest_rounds.py"), "refresh_b", tag, "gather64", "configs"], check=False)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_stamp
try:
    taken_at = open(os.path.join(SRC, "source_sha16.txt")).read().split()[0]
except Exception:
    taken_at = None
print("stamp:", profile_stamp.write(tag, taken_at))

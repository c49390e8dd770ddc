#!/usr/bin/env python3
"""Round-5 additions to the committed profile summaries:

  profiles/<tag>_pmc_pair_rounds_by_round.json   every pair round of one 2^20 proof -- k_affine_round<true> AND <false>, the later rounds
                                                 that had no counters before -- with request-level, write-side and issue-side counters
  profiles/<tag>_gather64_load_forms.json        tools/ubench/gather64.hip: which load form makes a 64-byte point cost a 64-byte request
  profiles/<tag>_config3_ecfft_2p20_*            rocprofv3 --kernel-trace --stats of BASELINE configs #3, #5 and of the setup
  profiles/<tag>_config5_sparse_2p22_*
  profiles/<tag>_setup_2p20_*

usage: python tools/digest_rounds.py <dir under gpurun_out/> <tag> [by_round|gather64|configs ...]
       (tools/digest_profiles.py calls it: by_round on gpurun_out/refresh, gather64 + configs on gpurun_out/refresh_b)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "refresh")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r06"
DST = os.path.join(ROOT, "profiles")
N_CU, N_SIMD = 256, 1024


def dispatches(name):
    """per dispatch {counter: value, ms, name} of the trimmed per-dispatch table <name>_pair_rounds.csv (tools/r5_first.sh,
    tools/refresh_profiles.sh keep the rows of the pair-round kernels and the microbenchmarks of a --pmc pass)"""
    path = os.path.join(SRC, name + "_pair_rounds.csv")
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    # bench.py's untimed legs come after the proofs (microbenchmarks first, then stand-alone MSMs, proofs without tables ...): only the
    # launches before the first microbenchmark dispatch belong to proofs of the timed configuration
    first_extra = min([i for i, v in disp.items() if "k_ubench" in v["name"]], default=None)
    return [v for i, v in sorted(disp.items()) if "k_affine_round" in v["name"] and (first_extra is None or i < first_extra)]


def last_proof(seq):
    """the pair rounds of the LAST proof of the run: [commit MSM rounds], [K MSM rounds] (each starts with its <true> launch)"""
    firsts = [i for i, v in enumerate(seq) if "<true" in v["name"]]
    a, b = firsts[-2], firsts[-1]
    return seq[a:b], seq[b:]


def by_round():
    tcc, tcc2, wr, sq = (dispatches(n) for n in ("pmc_tcc", "pmc_tcc2", "pmc_wr", "pmc_sq"))
    bench = None
    for cand in ("pmc_sq_detail.json", "bench_detail.json", "pmc_sq.json", "pmc_tcc.json"):
        try:
            txt = open(os.path.join(SRC, cand)).read()
            bench = json.loads(txt) if cand.endswith("_detail.json") else json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
            if "msm_windows" not in bench.get("config", {}):
                continue
            break
        except Exception:
            continue
    cfg = bench["config"]
    m = 1 << cfg["log2_constraints"]
    pairs = [cfg["n_wires"] + m, 4 * m]
    windows = [cfg["msm_windows"]["commit_msm"]["windows"], cfg["msm_windows"]["k_msm"]["windows"]]
    out = {"source": "four separate rocprofv3 --pmc passes (counters + --kernel-trace only) of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "
                     "--in-flight 1`: TCC_EA0_RDREQ_sum/_32B/_64B/_128B; TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum; TCC_EA0_WRREQ_sum "
                     "TCC_EA0_WRREQ_64B_sum WRITE_SIZE; SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE.  "
                     "Rows = the pair rounds of the LAST proof of each pass, matched by position (the launch sequence is deterministic).",
           "additions_note": "additions of round r = entries / 2^(r+1) with entries = pairs x windows (one entry per signed window; zero digits and odd "
                             "leftovers make the true figure a few per cent smaller)",
           "msms": []}
    tot = {"first_ms": 0.0, "later_ms": 0.0, "first_adds": 0.0, "later_adds": 0.0}
    for which, (s_tcc, s_tcc2, s_wr, s_sq) in enumerate(zip(last_proof(tcc), last_proof(tcc2), last_proof(wr), last_proof(sq))):
        rows = []
        entries = pairs[which] * windows[which]
        for r, (a, b, c, e) in enumerate(zip(s_tcc, s_tcc2, s_wr, s_sq)):
            adds = entries / 2.0 ** (r + 1)
            ms = (a["ms"] + b["ms"] + c["ms"] + e["ms"]) / 4
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            rd = a["TCC_EA0_RDREQ_sum"]
            rd_bytes = 32 * a["TCC_EA0_RDREQ_32B_sum"] + 64 * a["TCC_EA0_RDREQ_64B_sum"] + 128 * a["TCC_EA0_RDREQ_128B_sum"]
            wrq, wr64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            wr_bytes = 64 * wr64 + 32 * (wrq - wr64)
            rows.append({
                "round": r, "kernel": "k_affine_round<%s>" % ("true" if "<true" in a["name"] else "false"),
                "additions": adds, "ms_under_pmc": ms, "additions_per_s": adds / (ms * 1e-3),
                "read_requests": rd, "read_requests_128B": a["TCC_EA0_RDREQ_128B_sum"], "read_requests_64B": a["TCC_EA0_RDREQ_64B_sum"],
                "read_requests_per_addition": rd / adds, "read_bytes": rd_bytes,
                "write_requests": wrq, "write_requests_64B": wr64, "write_bytes": wr_bytes, "WRITE_SIZE_KB": c.get("WRITE_SIZE"),
                "traffic_bytes": rd_bytes + wr_bytes, "traffic_bytes_per_addition": (rd_bytes + wr_bytes) / adds,
                "traffic_tb_per_s": (rd_bytes + wr_bytes) / (ms * 1e-3) / 1e12,
                "algorithmic_bytes": 96.0 * pairs[which] if r == 0 else 128.0 * adds,
                "l2_hit_rate": b["TCC_HIT_sum"] / (b["TCC_HIT_sum"] + b["TCC_MISS_sum"]),
                "valu_insts": e["SQ_INSTS_VALU"], "lds_insts": e["SQ_INSTS_LDS"], "valu_insts_per_addition": e["SQ_INSTS_VALU"] * 64.0 / adds,
                "effective_clock_ghz": cyc / (e["ms"] * 1e-3) / 1e9,
                "valu_insts_per_simd_cycle": e["SQ_INSTS_VALU"] / N_SIMD / cyc,
                "resident_wave_frac": e["SQ_WAVE_CYCLES"] * 4.0 / (cyc * N_CU * 12),
                "product_equivalents_per_s": adds * (5.13 + 15.0 / max(1.0, min(48.0, adds / 196608.0))) / (ms * 1e-3),
            })
            key = "first" if r == 0 else "later"
            tot[key + "_ms"] += ms
            tot[key + "_adds"] += adds
        out["msms"].append({"which": "commit MSM" if which == 0 else "K MSM", "pairs": pairs[which], "windows": windows[which], "rounds": rows})
    out["per_proof"] = {"first_rounds_ms": tot["first_ms"], "later_rounds_ms": tot["later_ms"],
                        "first_rounds_additions_per_s": tot["first_adds"] / (tot["first_ms"] * 1e-3),
                        "later_rounds_additions_per_s": tot["later_adds"] / (tot["later_ms"] * 1e-3)}
    out["reading"] = ("the later rounds issue AS MANY VALU instructions per SIMD-cycle as the multiplier microbenchmark (0.235) while a round fills the chip "
                      "several times over; what they lose is the clock the chip holds (1.66-2.0 GHz against 2.2) and, below ~2 M additions, the whole "
                      "chip: a round of B = 8 slots per thread on 1.5 workgroups per CU ends on half-empty CUs (resident-wave fraction 0.6)")
    json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_pair_rounds_by_round.json"), "w"), indent=1)
    for msm in out["msms"]:
        for r in msm["rounds"]:
            print("%-10s r%d %-22s %6.3f ms %5.2f G adds/s  %4.2f rd req/add %4.0f B/add  L2 hit %.2f  clk %.2f GHz  valu/simd/cyc %.3f  resident %.2f"
                  % (msm["which"], r["round"], r["kernel"], r["ms_under_pmc"], r["additions_per_s"] / 1e9, r["read_requests_per_addition"],
                     r["traffic_bytes_per_addition"], r["l2_hit_rate"], r["effective_clock_ghz"], r["valu_insts_per_simd_cycle"], r["resident_wave_frac"]))
    print("per proof:", out["per_proof"])


def gather64():
    rates = {}
    for line in open(os.path.join(SRC, "gather64_rates.log")):
        if "G points/s" in line:
            name, rest = line[:64].strip(), line[64:].split()
            rates[name] = {"g_points_per_s": float(rest[0]), "ms_per_launch": float(rest[3])}
    d1, d2, d3 = (json.load(open(os.path.join(SRC, f"g64_{k}_digest.json"))) for k in ("tcc", "tcc2", "fetch"))
    kern = {"plain (4 x dwordx4 per lane, default policy)": "k_g_plain", "nt (__builtin_nontemporal_load)": "k_g_nt", "asm default": "k_g_asm_default",
            "asm sc0": "k_g_asm_sc0", "asm sc1": "k_g_asm_sc1", "asm sc0 sc1": "k_g_asm_sc0sc1", "asm nt": "k_g_asm_nt", "asm sc1 nt": "k_g_asm_sc1nt",
            "asm sc0 sc1 nt": "k_g_asm_sc0sc1nt", "x only (first 32 bytes of the point)": "k_g_x32",
            "quad of lanes per point (1 x dwordx4 per lane)": "k_g_quad", "scalar loads (s_load_dwordx16, one point per wave-instruction)": "k_g_sload",
            "plain, table in hipDeviceMallocUncached memory": "k_g_plain_uncached", "plain, table in hipDeviceMallocFinegrained memory": "k_g_plain_finegrained"}
    points = 3072 * 256 * 32
    rows = []
    for name, k in kern.items():
        if k not in d1:
            continue
        pts = points / 16 if k == "k_g_sload" else points
        a, b, c = d1[k], d2.get(k, {}), d3.get(k, {})
        rows.append({"variant": name, "kernel": k, "points_per_launch": pts, **rates.get(name, {}),
                     "TCC_EA0_RDREQ": a.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": a.get("TCC_EA0_RDREQ_32B_sum"),
                     "TCC_EA0_RDREQ_64B": a.get("TCC_EA0_RDREQ_64B_sum"), "TCC_EA0_RDREQ_128B": a.get("TCC_EA0_RDREQ_128B_sum"),
                     "fabric_read_requests_per_point": a.get("TCC_EA0_RDREQ_sum", 0) / pts,
                     "fabric_read_bytes_per_point": (64 * a.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * a.get("TCC_EA0_RDREQ_128B_sum", 0) + 32 * a.get("TCC_EA0_RDREQ_32B_sum", 0)) / pts,
                     "TCC_REQ": b.get("TCC_REQ_sum"), "TCC_HIT": b.get("TCC_HIT_sum"), "TCC_MISS": b.get("TCC_MISS_sum"),
                     "l2_requests_per_point": (b.get("TCC_REQ_sum") or 0) / pts, "TCP_TCC_READ_REQ": c.get("TCP_TCC_READ_REQ_sum"), "FETCH_SIZE_KB": c.get("FETCH_SIZE")})
    out = {"source": "tools/ubench/gather64.hip on one MI355X: un-profiled rates, then three separate rocprofv3 --pmc passes (counters + --kernel-trace only): "
                     "TCC_EA0_RDREQ_sum/_32B/_64B/_128B; TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum; FETCH_SIZE TCP_TCC_READ_REQ_sum.  Every variant "
                     "gathers random 64-byte points out of the same 3 GB table (the scalar-load variant 1/16 as many).",
           "variants": rows,
           "reading": "Every VECTOR-memory form -- default, nt, sc0, sc1, sc0 sc1, their combinations, x only (32 of the 64 bytes), a quad of lanes per point, "
                      "uncached and fine-grained device memory -- costs ONE 128-byte fabric request per 64-byte point (64-byte requests: < 0.01 %).  sc1 / nt "
                      "only stop the vector L1 from keeping the line: the four 16-byte loads of a point then reach the L2 separately (4 L2 requests per point, "
                      "three of them hits) and the gather runs at HALF the rate.  The only path that produces 64-byte fabric requests is the SCALAR one "
                      "(s_load_dwordx16 through the scalar cache, whose lines are 64 bytes): all of its requests are 64-byte ones -- so the L2 can fill half a "
                      "line, but only for a client that asks for one, and the vector L1 always asks for its whole 128-byte line.  The scalar path is not a "
                      "way to feed per-lane gathers (one point per wave-instruction, 16 v_writelane per point to reach a lane: ~2 k more VALU instructions "
                      "per wave and slot pair against ~1 k for the product it would feed).  Recorded as a negative: the 128-byte line per 64-byte point stays."}
    json.dump(out, open(os.path.join(DST, f"{TAG}_gather64_load_forms.json"), "w"), indent=1)
    for r in rows:
        print("%-64s %6.2f G pts/s  %.2f fabric req/pt  %5.1f B/pt  64B req %.3g  128B req %.3g  L2 req/pt %.2f"
              % (r["variant"], r.get("g_points_per_s", 0), r["fabric_read_requests_per_point"], r["fabric_read_bytes_per_point"], r["TCC_EA0_RDREQ_64B"] or 0,
                 r["TCC_EA0_RDREQ_128B"] or 0, r["l2_requests_per_point"]))


def configs():
    for src, dst in (("ecfft20", "config3_ecfft_2p20"), ("sparse22", "config5_sparse_2p22"), ("setup20", "setup_2p20")):
        st = os.path.join(SRC, src + "_kernel_stats.csv")
        if os.path.exists(st):
            shutil.copy(st, os.path.join(DST, f"{TAG}_{dst}_kernel_stats.csv"))
        lg = os.path.join(SRC, src + ".log")
        if os.path.exists(lg):
            keep = [l for l in open(lg, errors="replace") if not (l[:1] in "WEI" and l[1:5].isdigit())]  # drop the profiler's own glog lines
            open(os.path.join(DST, f"{TAG}_{dst}.log"), "w").writelines(keep)


def config3_steady():
    for src, dst in (("ecfft20_steady.txt", "config3_ecfft_2p20_steady_state.txt"), ("ecfft20_steady_enter_kernel_stats.csv", "config3_ecfft_2p20_steady_enter_kernel_stats.csv"),
                     ("ecfft20_steady_exit_kernel_stats.csv", "config3_ecfft_2p20_steady_exit_kernel_stats.csv")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, f"{TAG}_{dst}"))


if __name__ == "__main__":
    want = sys.argv[3:] or ["by_round", "gather64", "configs"]
    if "configs" in want:
        config3_steady()
    for fn in [f for f in (by_round, gather64, configs) if f.__name__ in want]:
        try:
            fn()
        except FileNotFoundError as e:
            print(f"{fn.__name__}: skipped ({e})")

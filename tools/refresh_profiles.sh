#!/bin/bash
# Runs ON THE MI355X BOX (gpurun -- 'bash tools/refresh_profiles.sh'): the un-profiled bench line, the same command under
# rocprofv3 --kernel-trace --stats, and three separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ issue counters -- counters
# only, with --kernel-trace, as the pool requires).  Everything lands in gpurun_out/refresh/; tools/digest_profiles.py
# then writes the summaries into profiles/.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
python3 tools/profile_stamp.py > $OUT/source_sha16.txt   # which kernel sources these profiles are taken at (bench.py compares)
DVP_BENCH_WRITE_PROFILE=1 timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cp $ROOT/gpurun_out/bench_detail.json $OUT/bench_detail.json   # the sidecar of that line (everything the run measured)
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
# the default command (the timed one-at-a-time loop, then -- outside it -- the two-in-flight leg and the other extras skipped here) under the same tracer
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_default -o d --output-format csv -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_default_under_rocprof.json 2> $OUT/stats_default.err
echo "stats (default command) done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write done"
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o q --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --extras ubench > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || echo "pmc sq pass failed (see pmc_sq.err)"
cp $ROOT/gpurun_out/bench_detail.json $OUT/pmc_sq_detail.json || true
echo "pmc sq done"
# request-level view of the same launches (resolves FETCH_SIZE's request-size ambiguity): read requests of the L2's memory side, how many
# of them are 32-byte ones, and the L2 hit / miss split.  Counter names differ between ROCm builds: the list is saved first.
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $OUT/pmc_tcc -o t --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/pmc_tcc.json 2> $OUT/pmc_tcc.err || echo "pmc tcc pass failed (see pmc_tcc.err)"
timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum --kernel-trace -d $OUT/pmc_tcc2 -o t --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/pmc_tcc2.json 2> $OUT/pmc_tcc2.err || echo "pmc tcc2 pass failed (see pmc_tcc2.err)"
# write side of the same launches (the later rounds' bytes: tools/digest_rounds.py)
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum WRITE_SIZE --kernel-trace -d $OUT/pmc_wr -o t --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/pmc_wr.json 2> $OUT/pmc_wr.err || echo "pmc wr pass failed (see pmc_wr.err)"
echo "pmc tcc done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/msm -o m --output-format csv -- python3 $ROOT/tools/msm_bench.py 16 20 22 > $OUT/msm_bench.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/msm16 -o m --output-format csv -- python3 $ROOT/tools/msm_bench.py 16 > $OUT/msm16_bench.log 2>&1
echo "msm stats done"
cd $ROOT && timeout -s USR1 -k 30 300 python3 tools/wave_trace.py 20 $OUT/wave_trace.json > $OUT/wave_trace.log 2>&1 || echo "wave trace failed"
rm -f $OUT/wave_trace.npy
echo "wave trace done"
# per-dispatch rows of the pair-round kernels (first AND later rounds) of every counter pass, for tools/digest_rounds.py
python3 - <<PY
import csv, glob, os
out = "$OUT"
for d in ("pmc_tcc", "pmc_tcc2", "pmc_wr", "pmc_sq"):
    hits = glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)
    if not hits: continue
    rows = [r for r in csv.DictReader(open(hits[0])) if "k_affine_round" in r["Kernel_Name"] or "k_ubench" in r["Kernel_Name"]]
    with open(os.path.join(out, d + "_pair_rounds.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp", "Grid_Size"], extrasaction="ignore")
        w.writeheader()
        for r in rows:
            r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
            w.writerow(r)
PY
ls -R $OUT | head -40

#!/bin/bash
# Runs ON THE MI355X BOX, second half of the profile refresh (tools/refresh_profiles.sh is the first; two gpurun calls because one
# would exceed a call's time limit); everything lands in gpurun_out/refresh_b/ and tools/digest_profiles.py reads it:
#   1. tools/ubench/gather64.hip -- the load forms that might make a 64-byte point cost a 64-byte fabric request: rates un-profiled,
#      then the request-size and hit / miss counters per variant (two separate --pmc passes, counters + kernel trace only)
#   2. request-level and issue-side counters over a short bench run, ALL kernels (the digest reads k_affine_round<false>, the later
#      pair rounds, which had no counters before)
#   3. rocprofv3 --kernel-trace --stats for BASELINE configs #3 (ecfft_bench 20), #5 (sparse_bench 22 8) and the setup (setup_time 20)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/refresh_b
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
mkdir -p $ROOT/scratch
G=$ROOT/scratch/gather64
[ -x $G ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $G tools/ubench/gather64.hip
timeout -k 10 120 $G 3 > $OUT/gather64_rates.log 2>&1
echo "gather64 rates done"; cat $OUT/gather64_rates.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $OUT/g64_tcc -o g --output-format csv -- $G 3 > $OUT/g64_tcc.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace -d $OUT/g64_tcc2 -o g --output-format csv -- $G 3 > $OUT/g64_tcc2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum --kernel-trace -d $OUT/g64_fetch -o g --output-format csv -- $G 3 > $OUT/g64_fetch.log 2>&1 || echo "g64 fetch pass failed"
echo "gather64 counters done"
for d in g64_tcc g64_tcc2 g64_fetch; do
  PMC_TOP=40 python3 $ROOT/tools/pmc_digest.py $OUT/$d > $OUT/${d}_digest.txt && cp $OUT/$d/digest.json $OUT/${d}_digest.json || true
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/ecfft20 -o e --output-format csv -- python3 $ROOT/tools/ecfft_bench.py 20 > $OUT/ecfft20.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/sparse22 -o s --output-format csv -- python3 $ROOT/tools/sparse_bench.py 22 8 > $OUT/sparse22.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/setup20 -o s --output-format csv -- python3 $ROOT/tools/setup_time.py 20 > $OUT/setup20.log 2>&1
# config #3 again as ONE steady-state enter and ONE steady-state exit (the table bootstrap runs before the markers): per-launch durations
bash $ROOT/tools/ecfft_trace.sh > $OUT/ecfft20_steady.txt 2>&1 || echo "steady-state ecfft trace failed"
cp $ROOT/gpurun_out/ecfft_trace/steady_enter_kernel_stats.csv $OUT/ecfft20_steady_enter_kernel_stats.csv || true
cp $ROOT/gpurun_out/ecfft_trace/steady_exit_kernel_stats.csv $OUT/ecfft20_steady_exit_kernel_stats.csv || true
cd /tmp
echo "config stats done"
python3 - <<PY
import glob, os
out = "$OUT"
for d in ("ecfft20", "sparse22", "setup20"):
    for f in glob.glob(os.path.join(out, d, "**", "*kernel_stats.csv"), recursive=True):
        os.replace(f, os.path.join(out, d + "_kernel_stats.csv"))
PY
for d in g64_tcc g64_tcc2 g64_fetch ecfft20 sparse22 setup20; do rm -rf $OUT/$d; done
ls -la $OUT
cat $OUT/g64_tcc_digest.txt
for f in $OUT/ecfft20.log $OUT/sparse22.log; do tail -n 2 $f; done; tail -n 8 $OUT/setup20.log

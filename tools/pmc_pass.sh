#!/bin/bash
# Runs ON THE MI355X BOX: one rocprofv3 PMC pass (counters only + kernel trace) over a short bench.py run.
# usage: bash tools/pmc_pass.sh <tag> <counter> [<counter> ...]      -> gpurun_out/pmc_<tag>/
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o q --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --extras ubench > $OUT/bench.json 2> $OUT/err.log
python3 $ROOT/tools/pmc_digest.py $OUT

#!/bin/bash
# Runs ON THE MI355X BOX: rocprofv3 --kernel-trace --stats over a short bench.py run -> gpurun_out/stats_<tag>/
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/stats_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras ubench "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print(r["Name"].split("(")[0][-50:].ljust(50), r["Calls"].rjust(5), "%9.3f ms total %9.1f us avg"%(int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY

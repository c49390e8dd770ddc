#!/bin/bash
# Runs ON THE MI355X BOX: kernel trace of a short bench run under each value of one knob, then the timeline of the last proof
#   bash tools/r5_gaps.sh DVP_PROVE_HOST_TRANSCRIPT "1 0"      -> gpurun_out/gaps_<knob>_<value>.txt
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
KNOB=$1; VALS=$2
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  OUT=$ROOT/gpurun_out/gaps_${KNOB}_$v
  rm -rf $OUT && mkdir -p $OUT
  export $KNOB=$v
  timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT -o g --output-format csv -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/bench.json 2> $OUT/err.log
  f=$(find $OUT -name '*kernel_trace.csv' | head -1)
  python3 $ROOT/tools/timeline_gaps.py $f 1 30 > $ROOT/gpurun_out/gaps_${KNOB}_$v.txt
  python3 $ROOT/tools/timeline_gaps.py $f 2 12 >> $ROOT/gpurun_out/gaps_${KNOB}_$v.txt
  rm -rf $OUT
  echo "== $KNOB=$v"; head -3 $ROOT/gpurun_out/gaps_${KNOB}_$v.txt; grep -A14 'largest gaps' $ROOT/gpurun_out/gaps_${KNOB}_$v.txt | head -16
done

#!/bin/bash
# Runs ON THE MI355X BOX: bench.py's own timed loop under two (or more) values of one environment knob, alternating, N rounds:
#   bash tools/ab_knob.sh DVP_PROVE_HOST_TRANSCRIPT "1 0" 3        -> gpurun_out/ab_<knob>.log (one line per run)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
KNOB=$1; VALS=$2; N=${3:-3}; STEPS=${4:-30}
OUT=$ROOT/gpurun_out/ab_$KNOB.log
: > $OUT
cd $ROOT
for r in $(seq 1 $N); do
  for v in $VALS; do
    line=$(env $KNOB=$v timeout -k 10 300 python3 bench.py --steps $STEPS --warmup 3 --no-extras --in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$KNOB=$v round $r: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); s=d["stages_ms_per_step"]; print("%.3f ms/step | msm %.3f sort %.3f r0 %.3f later %.3f tail %.3f extend %.3f" % (d["ms_per_step"], s["msm_total"], s["msm_recode_sort"], s["msm_affine_round0"], s["msm_affine_later_rounds"], s["msm_merge_frobenius_tail"], s["extend"]))')" | tee -a $OUT
  done
done

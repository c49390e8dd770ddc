#!/bin/bash
# Runs ON THE MI355X BOX: kernel trace of steady-state enter / exit calls at 2^20 (the table bootstrap call is made first and its
# dispatches are cut off by tools/ecfft_trace_digest.py through the marker kernel count).   -> gpurun_out/ecfft_trace/
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ecfft_trace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o e --output-format csv -- python3 $ROOT/tools/ecfft_steady.py 20 > $OUT/run.log 2>&1
python3 $ROOT/tools/ecfft_trace_digest.py $OUT > $OUT/digest.txt
cat $OUT/digest.txt

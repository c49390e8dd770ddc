#!/bin/bash
# VGPRs / SGPRs / scratch / occupancy of every kernel of one source (no GPU needed):  tools/kernel_resources.sh msm [name filter]
src=dv-pari_amd/csrc/$1.hip; [ -f "$src" ] || src=dv-pari_amd/csrc/$1.cpp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -x hip "$src" -o /dev/null -Wno-pass-failed -Wno-int-to-pointer-cast \
  -Rpass-analysis=kernel-resource-usage $3 2>&1 | awk -v f="${2:-.}" '
  /remark: Function Name:/ {name=$(NF-1)} /remark: +VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /TotalSGPRs:/ {sg=$(NF-1)}
  /LDS Size/ { if (name ~ f) printf "%s vgpr %s sgpr %s scratch %s occ %s\n", name, v, sg, s, o }' | while read -r n rest; do
    echo "$(echo "$n" | c++filt | sed 's/(.*//' | cut -c1-70) | $rest"; done

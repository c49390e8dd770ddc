#!/bin/bash
# A/B builds: tools/build_variant.sh NAME SRC "-DFLAG=1 ..."  ->  ab/libdvpari_NAME.so = the in-tree objects with SRC recompiled under the flags
# (load it with DVP_LIB=ab/libdvpari_NAME.so; ab/ is git-ignored but travels with gpurun)
set -e
name=$1; src=$2; flags=$3
C=dv-pari_amd/csrc
mkdir -p ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -x hip $C/$src.hip -o ab/${src}_$name.o -Wno-pass-failed -Wno-int-to-pointer-cast $flags
objs=""
for o in capi cache tree_io ecfft msm codec fr_ops prove setup; do
  [ -f $C/$o.o ] || continue
  if [ $o == $src ]; then objs="$objs ab/${src}_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libdvpari_$name.so $objs
echo built ab/libdvpari_$name.so

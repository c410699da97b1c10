#!/bin/bash
# tools/mkvar.sh <name> "<extra flags>" [tu ...]: variant of libbrmi.so in scratch/variants/<name>/ -- the named translation units (default:
# brmi_light) recompiled with the extra flags, the others taken from build/hip
set -e
name=$1; extra=$2; shift 2; tus=${@:-brmi_light}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Iinclude -Wall"
mkdir -p build/hip_$name scratch/variants/$name
objs=""
for f in basicrenderer_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [[ " $tus " == *" $b "* ]]; then rm -f build/hip_$name/$b.o; /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o build/hip_$name/$b.o 2>build/hip_$name/$b.log & objs="$objs build/hip_$name/$b.o"; else objs="$objs build/hip/$b.o"; fi
done
wait
for o in $objs; do [ -f $o ] || { echo "FAILED: $o"; grep error ${o%.o}.log | head -5; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o scratch/variants/$name/libbrmi.so
echo "built scratch/variants/$name/libbrmi.so"

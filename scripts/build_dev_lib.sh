#!/bin/bash
# usage: scripts/build_dev_lib.sh <out.so> <file.hip>[,<file2.hip>...] [extra hipcc flags for those files...]
# A second library for A/B and instrumentation runs (TN_LIB_PATH): the objects of the regular build (python -m tinynerf_amd.build first)
# with the named sources recompiled under extra flags.  scratch/ travels to the GPU box with the tree but stays out of git.
set -e
out=$1; srcs=$2; shift 2
objs=$(ls tinynerf_amd/csrc/_obj/*.o)
new=""
for src in ${srcs//,/ }; do
  obj=/tmp/_dev_$(basename ${src%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics -Iinclude "$@" -c tinynerf_amd/csrc/$src -o $obj
  objs=$(echo "$objs" | grep -v "/$(basename ${src%.hip}).o")
  new="$new $obj"
done
mkdir -p $(dirname $out)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs $new
echo built $out

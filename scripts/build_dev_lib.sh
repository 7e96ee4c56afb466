#!/bin/bash
# usage: scripts/build_dev_lib.sh <out.so> <file.hip> [extra hipcc flags for that file...]
# A second library for A/B and instrumentation runs (TN_LIB_PATH): the objects of the regular build (python -m tinynerf_amd.build first)
# with ONE source recompiled under extra flags.  scratch/ travels to the GPU box with the tree but stays out of git.
set -e
out=$1; src=$2; shift 2
obj=/tmp/_dev_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics "$@" -c tinynerf_amd/csrc/$src -o $obj
objs=$(ls tinynerf_amd/csrc/_obj/*.o | grep -v "/$(basename ${src%.hip}).o")
mkdir -p $(dirname $out)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs $obj
echo built $out

#!/bin/bash
# Build a VARIANT of libcmlhip.so for an A/B on the GPU box (tools/ab_run.sh): one translation unit recompiled with extra flags, the other objects as built.
#   bash tools/build_variant.sh <name> <unit.hip> [extra hipcc flags ...]   ->  ab_tmp/libcmlhip_<name>.so   (ab_tmp/ is untracked; it travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p ab_tmp
obj=ab_tmp/${unit%.hip}_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value "$@" -c libcml_amd/csrc/$unit -o $obj
objs=""
for o in libcml_amd/csrc/*.o; do
  if [ "$(basename $o)" = "${unit%.hip}.o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ab_tmp/libcmlhip_$name.so $objs
echo built ab_tmp/libcmlhip_$name.so

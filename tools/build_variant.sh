#!/bin/bash
# usage: tools/build_variant.sh NAME -DMACRO=VALUE … — build/libmeepo_hip_NAME.so with extra compiler flags (A/B runs: MEE_LIB_PATH=build/libmeepo_hip_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/obj_$name
# MEE_VARIANT_TU=meepo_apply: the extra flags go to that translation unit only
for f in meepo_table meepo_find meepo_export meepo_apply meepo_dedup meepo_router meepo_group meepo_sharded; do
  extra=("$@"); if [ -n "$MEE_VARIANT_TU" ] && [ "$MEE_VARIANT_TU" != "$f" ]; then extra=(); fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-pass-failed "${extra[@]}" \
    -c meepoembedding_amd/csrc/$f.hip -o build/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj_$name/*.o -ldl -o build/libmeepo_hip_$name.so
echo build/libmeepo_hip_$name.so

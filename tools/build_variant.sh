#!/bin/bash
# Development builds of libibvh.so for A/B measurements: tools/build_variant.sh TAG [extra hipcc flags...]
# -> variants/libibvh_TAG.so (Float32 / Int32 instantiations only: IBVH_ONLY_BENCH_TYPES; never shipped).
# Load one with IBVH_LIB=variants/libibvh_TAG.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
OBJ=variants/obj_$TAG
mkdir -p $OBJ
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DIBVH_ONLY_BENCH_TYPES -w $*"
pids=()
for f in implicitbvh.jl_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  # only the files that changed flags need a rebuild; objects of other variants are not shared on purpose (simple > clever)
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $OBJ/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libibvh_$TAG.so $OBJ/*.o -ldl
rm -rf $OBJ
echo built variants/libibvh_$TAG.so
